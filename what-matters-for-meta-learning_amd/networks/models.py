"""Shared building blocks (reference: networks/models.py:27-60 EncoderFC, 195-203 AttnLinear).

These classes are parameter containers with the reference's construction order (so a
seeded model draws identical initial weights) and state_dict keys; their own forward()
runs the mlhot HIP linear kernels, while the fused model path reads the parameters directly.
"""
import os

import torch
from torch import nn

from mlhot import lib
from mlhot.ops import Linear2Function, LinearFunction, MaxPool2Function, ResNetTrunkFunction, linear2_ok, mlp_chain
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


CHAINS_IN_MODELS = os.environ.get("MLHOT_MLP_CHAIN") == "1"


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


def _mlp3(x, seq, last_relu, side=None, side_first=False, pre=None):
    """Sequential(Linear, ReLU, Linear, ReLU, Linear[, ReLU]) over [side | x] / [x | side] (the reference's torch.cat in front of the
    first layer; `side` = None: just x), optionally behind a plain Linear `pre` applied to x first (the attention models' `mu`,
    ANP.py:52,128): ONE launch per direction through the chain kernels (mlhot.ops.mlp_chain) when the shapes fit them, the HIP
    linear kernels layer by layer otherwise.

    Measured on MI355X (c5: 120 rows, 256-wide layers): the chain kernels LOSE to one launch per
    layer at these shapes - a workgroup that owns 16 rows has to pull every layer's whole weight matrix (256 KB - 512 KB) through ONE
    CU (~70 GB/s) and to issue all of the layer's MFMAs on that CU: 48 us for a four-layer chain against 4 x 7.7 us - so the models
    use them only when MLHOT_MLP_CHAIN=1 (kept for A/B runs and for shapes with small weights); the default is one launch per layer."""
    lins = [m for m in seq if isinstance(m, nn.Linear)]
    acts = ["relu" if (i < len(lins) - 1 or last_relu) else "none" for i in range(len(lins))]
    layers = [(pre.weight, pre.bias, "none", None, False)] if pre is not None else []
    layers += [(lin.weight, lin.bias, act, side if i == 0 else None, side_first) for i, (lin, act) in enumerate(zip(lins, acts))]
    if CHAINS_IN_MODELS:
        y = mlp_chain(x, layers)
        if y is not None:
            return y
    if pre is not None:
        x = LinearFunction.apply(x, pre.weight, pre.bias, "none")
    first = 0
    if side is not None:
        xa, xb = (side, x) if side_first else (x, side)
        if xb.dim() == xa.dim() and xa.shape[:-1] == xb.shape[:-1] and linear2_ok(xa, xb, lins[0].weight):
            # the concatenation folded into the first layer: its two inputs are the two k ranges of one launch
            x, first = Linear2Function.apply(xa, xb, lins[0].weight, lins[0].bias, acts[0]), 1
        else:
            x = torch.cat([xa, xb], dim=-1)
    for lin, act in zip(lins[first:], acts[first:]):
        x = LinearFunction.apply(x, lin.weight, lin.bias, act)
    return x


def trunk_weights(conv1, resnet):
    """The 26 parameter tensors of a stem + ResNet(BasicBlock, [1,1,1,1]) trunk in the order mlhot_trunk_fwd takes them:
    (weight, bias) of the stem, then of (conv1, conv2, skip) of the four blocks."""
    out = [conv1.weight, conv1.bias]
    for layer in (resnet.layer1, resnet.layer2, resnet.layer3, resnet.layer4):
        blk = layer[0]
        out += [blk.conv1.weight, blk.conv1.bias, blk.conv2.weight, blk.conv2.bias, blk.downsample[0].weight, blk.downsample[0].bias]
    return out


def run_trunks(jobs):
    """Every ResNet-trunk pass of a model step in one call per direction (mlhot.ops.ResNetTrunkFunction).
    jobs: [(images [n, C, H, W], weights = list of 26 tensors, skip kernel 1 | 3, tap_log or None)]; passes whose `weights` hold
    the same tensor objects share one weight set (their gradients come out summed).  Returns the output maps [n, 64, H/32, W/32].
    Image sizes without weight-stationary kernels (anything but 3x64x64 / 1x128x128) return None: the caller composes the
    run-time-shaped convolution operators instead."""
    C, H, W = jobs[0][0].shape[1:]
    if H != W or not lib().trunk_supported(C, H) or any(tuple(j[0].shape[1:]) != (C, H, W) for j in jobs):
        return None
    imgs, wsets, passes = [], [], []
    for img, weights, skip_k, _ in jobs:
        ii = next((k for k, t in enumerate(imgs) if t is img), None)
        if ii is None:
            imgs.append(img)
            ii = len(imgs) - 1
        wi = next((k for k, (ws, _) in enumerate(wsets) if len(ws) == len(weights) and all(a is b for a, b in zip(ws, weights))), None)
        if wi is None:
            wsets.append((weights, skip_k))
            wi = len(wsets) - 1
        passes.append((ii, wi))
    want_taps = any(j[3] is not None for j in jobs)
    taps = [] if want_taps else None
    spec = (passes, [k for _, k in wsets], len(imgs), taps)
    outs = ResNetTrunkFunction.apply(spec, *imgs, *[t for ws, _ in wsets for t in ws])
    if want_taps:
        for (_, _, _, log), acts in zip(jobs, taps):
            if log is not None:
                log.append(list(acts))
    return list(outs)


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

    def trunk_job(self, img):
        """This encoder's pass over `img` as a run_trunks job.  The weight list is rebuilt on every call (26 attribute reads): a
        cached list would keep feeding the kernels the OLD tensors after a parameter is re-bound (load_state_dict(assign=True),
        parametrisations, `m.conv1.weight = nn.Parameter(...)`); two passes of one step share a weight set because run_trunks
        compares the tensors themselves."""
        return (img, trunk_weights(self.conv1, self.resnet), 1, self.tap_log)

    def features(self, fmap):
        """img_agg + reshape of a trunk output map [n, 64, h, w] -> [T, N, F]  (models.py:105-115)."""
        x = _aggregate_feature_map(fmap, self.aggregate)
        return x.view(self.task_num, -1, x.size(1))

    def forward(self, img):
        maps = run_trunks([self.trunk_job(img)])
        if maps is None:                                   # no weight-stationary kernels for this image size: per-operator route
            maps = [self.resnet.trunk(run_conv(self.conv1, img, relu=True), self._taps())]
        return self.features(maps[0])


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

    def trunk_job(self, imgs):
        return (imgs, trunk_weights(self.conv1, self.resnet), 1, self.tap_log)      # rebuilt per call, see ImageEncoder.trunk_job

    def forward(self, test_images, sample_features, log_variance=None, fmap=None, pre=None):
        """`fmap`: this decoder's trunk output over the target images when the caller already ran it together with the
        encoder passes (run_trunks); computed here otherwise.  `pre`: a Linear still to be applied to `sample_features` (the
        attention models' `mu`), so that it runs inside fc_mu's launch."""
        n_per_task = sample_features.size(1)
        if fmap is None:
            imgs = test_images.reshape(self.task_num * n_per_task, self.img_channels, self.img_size[0], self.img_size[1])
            maps = run_trunks([self.trunk_job(imgs)])
            if maps is None:
                taps = None
                if self.tap_log is not None:
                    self.tap_log.append([])
                    taps = self.tap_log[-1]
                maps = [self.resnet.trunk(run_conv(self.conv1, imgs, relu=True), taps)]
            fmap = maps[0]
        x = _aggregate_feature_map(fmap, self.aggregate).reshape(self.task_num, n_per_task, -1)
        mu = _mlp3(sample_features, self.fc_mu, last_relu=False, side=x, side_first=True, pre=pre)      # cat([x, sample_features]), models.py:182
        return mu, None
