"""Plugin `networks.FCLCNPShapeNet1D` (reference: networks/FCLCNPShapeNet1D.py): CNPShapeNet1D with functional contrastive
learning.  Same parameters and construction order as `networks.CNPShapeNet1D` (networks/_vanilla.py); the forward additionally
pushes the TARGET set through transform_y / encoder_r / a max over the shots / r_to_z and returns the NT-Xent term between the
context-set and the target-set task embeddings (trainer/losses.py:83-88) as 4th value.

The fused whole-model call (mlhot_np_vanilla_fwd) keeps the per-set embeddings inside its kernels, so this variant composes the
HIP operators instead: the vanilla encoder kernels on [context | target] images in one pass, mlhot linears, the shot-axis
aggregators.
"""
import torch
from torch import nn

from mlhot.ops import AggFunction, EncVanillaFunction, LinearFunction
from networks._vanilla import VanillaNP


class FCLCNPShapeNet1D(VanillaNP):
    ATTENTION = False
    OUT_TANH = True

    def _encode(self, images):
        convs = [m for m in self.encoder_w0 if isinstance(m, (nn.Conv2d, nn.Linear))]
        params = [t for m in convs for t in (m.weight, m.bias)]
        return EncVanillaFunction.apply(images, *params)

    def _task_embedding(self, x_img, labels, agg_mode):
        lab = LinearFunction.apply(labels, self.transform_y.weight, self.transform_y.bias, "none")
        rs = self.encoder_r(torch.cat([x_img, lab], dim=2))
        if agg_mode in ("mean", "max"):
            r, _ = AggFunction.apply(agg_mode, rs, None)
        elif agg_mode == "baco":
            mu = LinearFunction.apply(rs, self.rs_to_mu.weight, self.rs_to_mu.bias, "none")
            lv = LinearFunction.apply(rs, self.rs_to_var.weight, self.rs_to_var.bias, "none")
            r, _ = AggFunction.apply("baco", mu, lv)
        else:
            raise TypeError("agg_mode is not applicable for CNP, choose from ['mean', 'max', 'baco']")
        return LinearFunction.apply(r, self.r_to_z.weight, self.r_to_z.bias, "none")

    def forward(self, batch_train_images, label_train, batch_test_images, label_test, test=False):
        """-> (mu [T, Nq, y], None, 0, contrastive term)   (FCLCNPShapeNet1D.py:101-159)."""
        from trainer.losses import LossFunc
        self.test_num = batch_test_images.shape[1]
        self.ctx_num = batch_train_images.shape[1]
        if batch_test_images.shape[0] != self.task_num:
            raise ValueError(f"batch has {batch_test_images.shape[0]} tasks, model was built for {self.task_num}")
        T, Nc, Nq = self.task_num, self.ctx_num, self.test_num
        C, H, W = self.img_channels, self.img_size[0], self.img_size[1]
        images = torch.cat([batch_train_images.reshape(-1, C, H, W), batch_test_images.reshape(-1, C, H, W)], dim=0)
        feats = self._encode(images)
        x_ctx, x_qry = feats[:T * Nc].view(T, Nc, self.dim_w), feats[T * Nc:].view(T, Nq, self.dim_w)
        z_0 = None
        if Nc:
            z_0 = self._task_embedding(x_ctx, label_train, self.agg_mode)
            z = z_0[:, None, :].expand(-1, Nq, -1)
        else:
            z = torch.zeros(T, Nq, self.dim_z, device=batch_test_images.device)
        contra = 0
        if not test:
            if z_0 is None:
                raise ValueError("the contrastive term needs a non-empty context set (the reference fails here as well: z_0 is unbound)")
            z_q = self._task_embedding(x_qry, label_test, "max")          # the target set is always max-aggregated (line 147)
            contra = LossFunc.contrastive_loss(z_0, z_q)
        lins = [m for m in self.decoder0 if isinstance(m, nn.Linear)]
        h = torch.cat([x_qry, z], dim=-1)
        for lin, act in zip(lins, ("relu", "relu", "tanh")):
            h = LinearFunction.apply(h, lin.weight, lin.bias, act)
        return h, None, 0, contra
