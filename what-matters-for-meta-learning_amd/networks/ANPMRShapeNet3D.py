"""Plugin `networks.ANPMRShapeNet3D` (reference: networks/ANPMRShapeNet3D.py; BASELINE config c5):
the ResNet-encoder ANP whose context/target image encoder is Bayes-by-backprop (weights re-sampled on
every call, KL returned as `kl`), with a deterministic NPDecoder ResNet over the targets."""
from collections import OrderedDict

import torch
from torch import nn

from mlhot.ops import AddReluFunction, LinearFunction
from networks._resnet_np import ResNetNP
from networks.bbb.BBBConv import BBBConv2d
from networks.bbb.misc import FlattenLayer, ModuleWrapper, sample_all, sample_twice
from networks.fast_attention import FastAttention
from networks.models import AttnLinear, NPDecoder, _aggregate_feature_map, _mlp3, run_trunks


class BasicBlock(nn.Module):
    """BBB twin of the residual block: 3x3 s2 + ReLU, 3x3 s1, 3x3 s2 skip (ANPMRShapeNet3D.py:38-62)."""

    def __init__(self, inplanes, planes, stride=1, downsample=None, **kwargs):
        super().__init__()
        self.conv1 = BBBConv2d(inplanes, planes, stride=stride, **kwargs)
        self.conv1.fuse_relu = True
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = BBBConv2d(planes, planes, **kwargs)
        if stride != 1:
            downsample = nn.Sequential(BBBConv2d(inplanes, inplanes, stride=stride, **kwargs))
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, taps=None):
        mid = self.conv1(x)                                   # eps order: conv1, conv2, then the skip
        out = self.conv2(mid)
        identity = self.downsample(x) if self.downsample is not None else x
        y = AddReluFunction.apply(out, identity)
        if taps is not None:
            taps += [mid, y]
        return y


def conv_block(in_channels, out_channels, **kwargs):
    conv = BBBConv2d(in_channels, out_channels, **kwargs)
    conv.fuse_relu = True
    return nn.Sequential(OrderedDict([("conv", conv), ("relu", nn.Identity())]))   # ReLU is fused into the conv kernel


class BBBEncoder(ModuleWrapper):
    def __init__(self, img_channels, device):
        super().__init__()
        kw = dict(kernel_size=3, padding=1, bias=True)
        self.net = nn.Sequential(OrderedDict([
            ("layer1", conv_block(img_channels, 64, kernel_size=5, stride=2, padding=2, bias=True)),
            ("layer2", BasicBlock(64, 64, stride=2, **kw)), ("layer3", BasicBlock(64, 64, stride=2, **kw)),
            ("layer4", BasicBlock(64, 64, stride=2, **kw)), ("layer5", BasicBlock(64, 64, stride=2, **kw)),
            ("flatten", FlattenLayer(256))]))
        self.tap_log = None      # set to a list to record, per call, the post-ReLU activations (tests; as models.ImageEncoder)

    def _bbb_layers(self):
        """The BBB convolutions in the order their forwards run (= the reference's eps draw order): stem, then per block
        conv1, conv2, skip."""
        out = [self.net.layer1.conv]
        for name in ("layer2", "layer3", "layer4", "layer5"):
            blk = getattr(self.net, name)
            out += [blk.conv1, blk.conv2] + ([blk.downsample[0]] if blk.downsample is not None else [])
        return out

    def forward(self, x):
        kl = sample_all(self._bbb_layers())          # 26 tensors, one launch pair; each layer picks its sample up below
        taps = None
        if self.tap_log is not None:
            taps = []
            self.tap_log.append(taps)
        x = self.net.layer1(x)
        if taps is not None:
            taps.append(x)
        for name in ("layer2", "layer3", "layer4", "layer5"):
            x = getattr(self.net, name)(x, taps)
        return self.net.flatten(x), kl


class ANPMRShapeNet3D(ResNetNP):
    ATTENTION = True

    def __init__(self, config):
        nn.Module.__init__(self)
        self.device = config.device
        self.img_size = config.img_size
        self.img_channels = self.img_size[2] - 1 if config.task == "shapenet_3d" else self.img_size[2]
        self.task_num = config.tasks_per_batch
        self.label_dim = config.input_dim
        self.agg_mode, self.img_agg, self.y_dim = config.agg_mode, config.img_agg, config.output_dim
        torch.manual_seed(config.seed)
        self.img_encoder = BBBEncoder(img_channels=self.img_channels, device=self.device)
        self.task_encoder = nn.Sequential(nn.Linear(256 + self.label_dim, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                          nn.Linear(256, 256), nn.ReLU())
        self.mu = nn.Linear(256, 256)
        self.decoder = NPDecoder(aggregate=self.img_agg, output_dim=self.y_dim, task_num=self.task_num,
                                 img_channels=self.img_channels, img_size=self.img_size)
        h = 256
        self._W_k = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
        self._W_v = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
        self._W_q = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
        self._W = AttnLinear(self.N_HEADS * h, h)
        self.attn = FastAttention(dim_heads=256, causal=False)
        self.n_heads = self.N_HEADS

    def pixel_agg(self, x):
        # the BBB encoder already flattened to [n, 256]; only "reshape" is shape-consistent with it
        if self.img_agg not in ("reshape", "max", "baco", "mean"):
            raise TypeError("Non-valid img_agg!")
        x = x.reshape(x.size(0), -1)
        return x.view(self.task_num, -1, x.size(1))

    def forward(self, batch_train_images, label_train, batch_test_images, test=False):
        self._refresh_arena()
        self.test_num = batch_test_images.shape[1]
        self.ctx_num = batch_train_images.shape[1]
        C, H, W = self.img_channels, self.img_size[0], self.img_size[1]
        if self.ctx_num:
            ctx_imgs, tgt_imgs = batch_train_images.reshape(-1, C, H, W), batch_test_images.reshape(-1, C, H, W)
            fmap_dec = None
            maps = None
            from mlhot import lib
            if H == W and lib().trunk_supported(C, H):
                # both weight samples of the step in one launch pair, then all three ResNet passes (context and target images
                # through the two samples of the Bayes-by-backprop encoder, target images through the decoder) together
                w_ctx, w_tgt, kl = sample_twice(self.img_encoder._bbb_layers())
                log = self.img_encoder.tap_log
                maps = run_trunks([(ctx_imgs, w_ctx, 3, log), (tgt_imgs, w_tgt, 3, log), self.decoder.trunk_job(tgt_imgs)])
                if maps is not None and self.__dict__.get("_split_backward"):
                    self.__dict__["_cut_pairs"] = []
                    *maps, kl = self._cut(list(maps) + [kl])          # the KL hangs off the same sampling node as the trunks' weights
            if maps is not None:
                x_ctx, x_tgt, fmap_dec = maps[0].reshape(-1, 256), maps[1].reshape(-1, 256), maps[2]
            else:
                x_ctx, _ = self.img_encoder(ctx_imgs)
                x_tgt, kl = self.img_encoder(tgt_imgs)     # a second, independent weight sample
            x_ctx, x_tgt = self.pixel_agg(x_ctx), self.pixel_agg(x_tgt)
            feats = _mlp3(x_ctx, self.task_encoder, last_relu=True, side=label_train)      # cat([x_ctx, labels]), ANPMRShapeNet3D.py:204-205
            # mu (ANPMRShapeNet3D.py:209) runs inside the decoder head's launch: mu -> cat([features, sample]) -> fc_mu as one chain
            out, var = self.decoder(batch_test_images, self._multihead_attention(x_ctx, feats, x_tgt), fmap=fmap_dec, pre=self.mu)
            return out, var, kl
        sample = torch.zeros(self.task_num, self.test_num, 256, device=batch_test_images.device)
        out, var = self.decoder(batch_test_images, sample)
        return out, var, 0
