"""Meta-regularised (MR) twins of the vanilla CNP/ANP plugins (reference: networks/ANPMR.py,
ANPMRShapeNet1D.py, CNPMR.py, CNPMRShapeNet1D.py; SURVEY.md §8a rows E1 "MR twin" and B1).

The image encoder is the vanilla conv-conv-pool-conv-linear stack with Bayes-by-backprop weights:
all 8 tensors are re-sampled on EVERY encoder call (so the context and the target pass use different
draws), the KL of the call is returned as `kl`.  The sampled weights go straight into the E1 kernels
(mlhot_enc_vanilla_fwd/_bwd: the fused weight-stationary MFMA convolutions), the sample / KL are the
mlhot_bbb_sample kernels, the task-side layers run the mlhot linear / aggregator / FAVOR+ kernels.

The reference files also construct a ResNet NPDecoder, a `task_encoder` and a `mu` layer that forward()
never touches; they are kept (same construction order -> same seeded initial weights, same state_dict
keys; their gradients stay None).
"""
from collections import OrderedDict

import torch
from torch import nn

from mlhot.ops import AggFunction, EncVanillaFunction, FavorFunction, LinearFunction
from networks.bbb.BBBConv import BBBConv2d
from networks.bbb.BBBLinear import BBBLinear
from networks.bbb.misc import FlattenLayer, ModuleWrapper, sample_all
from networks.fast_attention import FastAttention
from networks.models import AttnLinear, EncoderFC, NPDecoder


def conv_block(in_channels, out_channels, **kwargs):
    return nn.Sequential(OrderedDict([("conv", BBBConv2d(in_channels, out_channels, **kwargs)), ("relu", nn.ReLU())]))


class BBBEncoder(ModuleWrapper):
    """ANPMR.py:40-53.  forward(img [n,1,128,128]) -> (features [n,dim_w], kl of this call's draws)."""

    def __init__(self, img_channels, dim_w, device):
        super().__init__()
        kw = dict(kernel_size=3, stride=2, padding=1, bias=True)
        self.net = nn.Sequential(OrderedDict([
            ("layer1", conv_block(img_channels, 32, **kw)), ("layer2", conv_block(32, 48, **kw)),
            ("pool", nn.MaxPool2d((2, 2))), ("layer3", conv_block(48, 64, **kw)),
            ("flatten", FlattenLayer(4096)), ("linear", BBBLinear(4096, dim_w, bias=True))]))

    def forward(self, img):
        layers = (self.net.layer1.conv, self.net.layer2.conv, self.net.layer3.conv, self.net.linear)
        kl = sample_all(layers)                # draw order of ModuleWrapper.forward: layer by layer, weight then bias; one launch pair
        params = []
        for layer in layers:
            (w, b), layer.presampled = layer.presampled, None
            params += [w, b]
        return EncVanillaFunction.apply(img, *params), kl


class VanillaMR(nn.Module):
    ATTENTION = False     # ANPMR* classes
    OUT_TANH = False      # *ShapeNet1D classes end decoder0 with nn.Tanh
    REDRAW_DECODER0 = False   # CNPMRShapeNet1D builds CNPMR first and then REPLACES decoder0 (new draws, last)
    N_HEADS = 8

    def __init__(self, config):
        super().__init__()
        self.device = config.device
        self.img_size = config.img_size
        self.img_channels = self.img_size[2]
        self.task_num = config.tasks_per_batch
        self.label_dim = config.input_dim
        self.agg_mode = config.agg_mode
        self.img_agg = config.img_agg
        self.y_dim = config.output_dim
        self.dim_w = config.dim_w
        self.n_hidden_units_r = config.n_hidden_units_r
        self.dim_r = config.dim_r
        self.dim_z = config.dim_z
        if list(self.img_size) != [128, 128, 1]:
            raise NotImplementedError("the vanilla encoder kernels are built for 128x128x1 images")
        torch.manual_seed(config.seed)

        self.encoder_w0 = BBBEncoder(self.img_channels, self.dim_w, device=self.device)
        self.transform_y = nn.Linear(self.label_dim, self.dim_w // 4)
        self.encoder_r = EncoderFC(input_dim=self.dim_w + self.dim_w // 4, n_hidden_units_r=self.n_hidden_units_r, dim_r=self.dim_r)
        self.r_to_z = nn.Linear(self.dim_r, self.dim_z)

        def decoder0(tanh):
            mods = [nn.Linear(self.dim_w + self.dim_z, 100), nn.ReLU(inplace=True), nn.Linear(100, 100), nn.ReLU(inplace=True),
                    nn.Linear(100, self.y_dim)]
            return nn.Sequential(*(mods + ([nn.Tanh()] if tanh else [])))

        self.decoder0 = decoder0(self.OUT_TANH and not self.REDRAW_DECODER0)
        self.task_encoder = nn.Sequential(nn.Linear(256 + self.label_dim, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                                          nn.Linear(256, 256), nn.ReLU())
        if self.agg_mode == "baco":
            self.rs_to_mu = nn.Linear(256, 256)
            self.rs_to_var = nn.Linear(256, 256)
        self.mu = nn.Linear(256, 256)
        self.decoder = NPDecoder(aggregate=self.img_agg, output_dim=self.y_dim, task_num=self.task_num,
                                 img_channels=self.img_channels, img_size=self.img_size)
        if self.ATTENTION:
            h = self.dim_w
            self._W_k = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W_v = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W_q = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W = AttnLinear(self.N_HEADS * h, h)
            self.attn = FastAttention(dim_heads=self.dim_r, causal=False)
            self.n_heads = self.N_HEADS
        if self.REDRAW_DECODER0:
            self.decoder0 = decoder0(self.OUT_TANH)

    def _heads(self, x, mods):
        w = torch.cat([m.linear.weight for m in mods], dim=0)
        b = torch.cat([m.linear.bias for m in mods], dim=0)
        T, N, _ = x.shape
        return LinearFunction.apply(x, w, b, "none").view(T, N, self.N_HEADS, -1)

    def _multihead_attention(self, k, v, q):
        merged = FavorFunction.apply(self._heads(q, self._W_q), self._heads(k, self._W_k), self._heads(v, self._W_v),
                                     self.attn.projection_matrix)
        return self._W(merged)

    def _encode(self, imgs, n):
        x, kl = self.encoder_w0(imgs.reshape(-1, self.img_channels, self.img_size[0], self.img_size[1]))
        return x.reshape(self.task_num, n, self.dim_w), kl

    def _context(self, batch_train_images, label_train, x_qry):
        x_ctx, _ = self._encode(batch_train_images, self.ctx_num)
        ly = LinearFunction.apply(label_train, self.transform_y.weight, self.transform_y.bias, "none")
        rs = self.encoder_r(torch.cat([x_ctx, ly], dim=2))
        if self.ATTENTION:
            if self.agg_mode != "attention":
                raise TypeError("agg_mode is not applicable for CNP, choose from ['attention']")
            r = self._multihead_attention(x_ctx, rs, x_qry)
            return LinearFunction.apply(r, self.r_to_z.weight, self.r_to_z.bias, "none")
        if self.agg_mode in ("mean", "max"):
            r, _ = AggFunction.apply(self.agg_mode, rs, None)
        elif self.agg_mode == "baco":
            mu_l = LinearFunction.apply(rs, self.rs_to_mu.weight, self.rs_to_mu.bias, "none")
            lv = LinearFunction.apply(rs, self.rs_to_var.weight, self.rs_to_var.bias, "none")
            r, _ = AggFunction.apply("baco", mu_l, lv)
        else:
            raise TypeError("agg_mode is not applicable for CNP, choose from ['mean', 'max', 'baco']")
        z = LinearFunction.apply(r, self.r_to_z.weight, self.r_to_z.bias, "none")
        return z[:, None, :].expand(-1, self.test_num, -1)

    def forward(self, batch_train_images, label_train, batch_test_images, test=False):
        """-> (mu [T,Nq,y], None, kl of the TARGET pass's draws)  (ANPMR.py:173-216, CNPMR.py:128-172)."""
        self.test_num = batch_test_images.shape[1]
        self.ctx_num = batch_train_images.shape[1]
        if self.ATTENTION:                                    # ANPMR encodes the targets first ...
            x_qry, kl = self._encode(batch_test_images, self.test_num)
        if self.ctx_num:
            z = self._context(batch_train_images, label_train, x_qry if self.ATTENTION else None)
        else:
            z = torch.zeros(self.task_num, self.test_num, self.dim_z, device=batch_test_images.device)
        if not self.ATTENTION:                                # ... CNPMR after the context: the eps draw order differs
            x_qry, kl = self._encode(batch_test_images, self.test_num)
        x = torch.cat([x_qry, z], dim=-1)
        lins = [m for m in self.decoder0 if isinstance(m, nn.Linear)]
        for i, lin in enumerate(lins):                        # decoder0: Linear+ReLU, Linear+ReLU, Linear[+Tanh], fused per layer
            act = "relu" if i < len(lins) - 1 else ("tanh" if self.OUT_TANH else "none")
            x = LinearFunction.apply(x, lin.weight, lin.bias, act)
        return x, None, kl
