"""Common implementation of the four vanilla CNP/ANP plugins.

The reference keeps four near-identical files (CNPVanillaPascal1D.py, CNPShapeNet1D.py,
ANPVanillaPascal1D.py, ANPShapeNet1D.py); here they are one class parameterised by
(attention?, tanh?).  The constructor creates the same torch layers in the same order
under `torch.manual_seed(config.seed)`, so initial weights and state_dict keys/shapes are
those of the reference; forward() is ONE call into libmlhot.so (mlhot_np_vanilla_fwd) and
its autograd backward is mlhot_np_vanilla_bwd.
"""
import torch
from torch import nn

from mlhot import lib
from mlhot.ops import VanillaNPFunction
from networks.fast_attention import FastAttention
from networks.models import AttnLinear, EncoderFC


class VanillaNP(nn.Module):
    ATTENTION = False     # ANP* classes
    OUT_TANH = False      # *ShapeNet1D classes end decoder0 with nn.Tanh
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
            raise NotImplementedError("the vanilla encoder kernels are built for 128x128x1 images "
                                      "(every Pascal1D / ShapeNet1D config)")
        torch.manual_seed(config.seed)  # fixed initialisation, as the reference

        self.encoder_w0 = nn.Sequential(
            nn.Conv2d(self.img_channels, 32, kernel_size=3, stride=2, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(32, 48, kernel_size=3, stride=2, padding=1), nn.ReLU(inplace=True),
            nn.MaxPool2d((2, 2)),
            nn.Conv2d(48, 64, kernel_size=3, stride=2, padding=1), nn.ReLU(inplace=True),
            nn.Flatten(), nn.Linear(4096, self.dim_w))
        self.transform_y = nn.Linear(self.label_dim, self.dim_w // 4)
        self.encoder_r = EncoderFC(input_dim=self.dim_w + self.dim_w // 4,
                                   n_hidden_units_r=self.n_hidden_units_r, dim_r=self.dim_r)
        self.r_to_z = nn.Linear(self.dim_r, self.dim_z)
        dec = [nn.Linear(self.dim_w + self.dim_z, 100), nn.ReLU(inplace=True), nn.Linear(100, 100),
               nn.ReLU(inplace=True), nn.Linear(100, self.y_dim)]
        if self.OUT_TANH:
            dec.append(nn.Tanh())
        self.decoder0 = nn.Sequential(*dec)

        if self.ATTENTION:
            h = self.dim_w
            self._W_k = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W_v = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W_q = nn.ModuleList([AttnLinear(h, h) for _ in range(self.N_HEADS)])
            self._W = AttnLinear(self.N_HEADS * h, h)
            self.attn = FastAttention(dim_heads=self.dim_r, causal=False)
            self.n_heads = self.N_HEADS
        elif self.agg_mode == "baco":
            self.rs_to_mu = nn.Linear(256, 256)
            self.rs_to_var = nn.Linear(256, 256)
        self._keys = None

    def _check_agg(self):
        ok = ("attention",) if self.ATTENTION else ("mean", "max", "baco")
        if self.agg_mode not in ok:
            raise TypeError(f"agg_mode is not applicable for {'ANP' if self.ATTENTION else 'CNP'}, "
                            f"choose from {list(ok)}")

    def _dims(self, ctx_num, test_num):
        proj = self.attn.projection_matrix if self.ATTENTION else None
        return lib().np_dims(self.task_num, ctx_num, test_num, self.label_dim, self.y_dim, self.dim_w, self.dim_r, self.dim_z,
                             list(self.n_hidden_units_r), 100, self.agg_mode or "mean", self.OUT_TANH,
                             proj.shape[0] if proj is not None else 0)

    def flat_layout(self, ctx_num, test_num):
        """(total floats, {parameter name: offset}) of the flat gradient buffer backward() fills for this batch shape
        (mlhot_np_grads_flat_layout); mlhot.optim.FlatAdam lays the parameters out the same way."""
        return lib().np_grads_layout(self._dims(ctx_num, test_num))

    def forward(self, batch_train_images, label_train, batch_test_images, test=False):
        """ctx images [T,Nc,1,128,128], ctx labels [T,Nc,L], target images [T,Nq,1,128,128]
        -> (mu [T,Nq,y], None, 0)   (same contract as the reference forward)."""
        self.test_num = batch_test_images.shape[1]
        self.ctx_num = batch_train_images.shape[1]
        if self.ctx_num:
            self._check_agg()
        if batch_test_images.shape[0] != self.task_num:
            raise ValueError(f"batch has {batch_test_images.shape[0]} tasks, model was built for {self.task_num}")
        if self._keys is None:
            self._keys = tuple(k for k, _ in self.named_parameters())
        params = [p for _, p in self.named_parameters()]
        proj = self.attn.projection_matrix if self.ATTENTION else None
        dims = self._dims(self.ctx_num, self.test_num)
        mu = VanillaNPFunction.apply(dims, self._keys, proj, batch_train_images, label_train, batch_test_images, *params)
        return mu, None, 0
