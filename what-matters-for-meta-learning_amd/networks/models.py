"""Shared building blocks (reference: networks/models.py:27-60 EncoderFC, 195-203 AttnLinear).

These classes are parameter containers with the reference's construction order (so a
seeded model draws identical initial weights) and state_dict keys; their own forward()
runs the mlhot HIP linear kernels, while the fused model path reads the parameters directly.
"""
import torch
from torch import nn

from mlhot.ops import LinearFunction, MaxPool2Function
from networks.ResNet import BasicBlock, ResNet, run_conv


class EncoderFC(nn.Module):
    """(Linear+ReLU) x len(n_hidden_units_r) then Linear(-> dim_r); keys `layers.{0,2,..}`."""

    def __init__(self, input_dim, n_hidden_units_r, dim_r):
        super().__init__()
        self.input_dim, self.n_hidden_units_r, self.dim_r = input_dim, list(n_hidden_units_r), dim_r
        widths = [input_dim] + self.n_hidden_units_r
        mods = []
        for a, b in zip(widths[:-1], widths[1:]):
            mods += [nn.Linear(a, b), nn.ReLU(inplace=True)]
        mods.append(nn.Linear(widths[-1], dim_r))
        self.layers = nn.Sequential(*mods)

    def linears(self):
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    def forward(self, x):
        lins = self.linears()
        for lin in lins[:-1]:
            x = LinearFunction.apply(x, lin.weight, lin.bias, "relu")
        return LinearFunction.apply(x, lins[-1].weight, lins[-1].bias, "none")


class AttnLinear(nn.Module):
    """nn.Linear whose weight is re-drawn N(0, in^-1)  (models.py:195-203)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.linear = nn.Linear(in_channels, out_channels, bias=True)
        torch.nn.init.normal_(self.linear.weight, std=in_channels ** -0.5)

    def forward(self, x):
        return LinearFunction.apply(x, self.linear.weight, self.linear.bias, "none")


def _aggregate_feature_map(x, aggregate):
    """img_agg of ImageEncoder / NPDecoder (models.py:105-113): [n,64,h,w] -> [n,F]."""
    if aggregate in ("max", "baco"):                    # AdaptiveMaxPool2d((2,2))
        if x.shape[-1] == 4:
            x = MaxPool2Function.apply(x)
        elif x.shape[-1] != 2:
            raise NotImplementedError("adaptive max pool to 2x2 is implemented for 2x2 and 4x4 maps (64/128 px inputs)")
    elif aggregate == "mean":
        x = x.mean(dim=(2, 3))                          # 64 features; no shipped config uses it
    elif aggregate != "reshape":
        raise TypeError(f"img_agg {aggregate!r} is not supported")
    return x.reshape(x.size(0), -1)


def _mlp3(x, seq, last_relu):
    """Sequential(Linear, ReLU, Linear, ReLU, Linear[, ReLU]) through the HIP linear kernels."""
    lins = [m for m in seq if isinstance(m, nn.Linear)]
    for i, lin in enumerate(lins):
        act = "relu" if (i < len(lins) - 1 or last_relu) else "none"
        x = LinearFunction.apply(x, lin.weight, lin.bias, act)
    return x


class ImageEncoder(nn.Module):
    """5x5 s2 stem + ReLU, four BN-free BasicBlocks, img_agg  (models.py:63-117) -> [T, N, F]."""

    def __init__(self, aggregate, task_num, img_channels):
        super().__init__()
        self.img_channels, self.task_num, self.aggregate = img_channels, task_num, aggregate
        self.conv1 = nn.Conv2d(img_channels, 64, kernel_size=5, stride=2, padding=2, bias=True)
        self.resnet = ResNet(BasicBlock, [1, 1, 1, 1], pretrained=False, progress=True)
        self.tap_log = None      # set to a list to record, per call, the post-ReLU activations (tests)

    def _taps(self):
        if self.tap_log is None:
            return None
        self.tap_log.append([])
        return self.tap_log[-1]

    def forward(self, img):
        x = self.resnet.trunk(run_conv(self.conv1, img, relu=True), self._taps())
        x = _aggregate_feature_map(x, self.aggregate)
        return x.view(self.task_num, -1, x.size(1))


class NPDecoder(nn.Module):
    """Second ResNet over the target images, cat with the sampled latent, fc_mu  (models.py:120-192)."""

    def __init__(self, aggregate, output_dim, task_num, img_channels, img_size, pr_unc=False):
        super().__init__()
        self.img_channels, self.task_num, self.img_size, self.output_dim = img_channels, task_num, img_size, output_dim
        self.aggregate = aggregate
        self.conv1 = nn.Conv2d(img_channels, 64, kernel_size=5, stride=2, padding=2, bias=True)
        self.resnet = ResNet(BasicBlock, [1, 1, 1, 1], pretrained=False, progress=True)
        self.fc_mu = nn.Sequential(nn.Linear(256 + 256, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                   nn.Linear(256, output_dim))
        if pr_unc:
            raise NotImplementedError("pr_unc / fc_var is never enabled by the reference models (models.py:185-190)")
        self.tap_log = None

    def forward(self, test_images, sample_features, log_variance=None):
        n_per_task = sample_features.size(1)
        imgs = test_images.reshape(self.task_num * n_per_task, self.img_channels, self.img_size[0], self.img_size[1])
        taps = None
        if self.tap_log is not None:
            self.tap_log.append([])
            taps = self.tap_log[-1]
        x = self.resnet.trunk(run_conv(self.conv1, imgs, relu=True), taps)
        x = _aggregate_feature_map(x, self.aggregate).reshape(self.task_num, n_per_task, -1)
        mu = _mlp3(torch.cat([x, sample_features], dim=-1), self.fc_mu, last_relu=False)
        return mu, None
