"""LossFunc with the reference's interface (trainer/losses.py:22-99); the regression losses are the fused mlhot loss
kernels (forward reduction + backward seed in one launch each).

The functional-contrastive term (losses.py:82-99) is NT-Xent from the un-vendored dependency `pytorch_metric_learning`
(listed without a version in the reference's requirements.txt:12; `NTXentLoss(temperature=t)` with its defaults: cosine
similarity, every same-label pair a positive, every different-label pair a negative of its anchor, mean over the positive
pairs).  Its labels are always arange blocks, so the term runs as mlhot_nt_xent_fwd / _bwd (csrc/nt_xent.h: similarity on the
matrix core, per-anchor log-sum-exp, hand-derived backward; 2 + 1 launches, nothing built on the host, capturable); embeddings
outside the kernels' limits (N > 2048 rows or d > 256 / d % 16 != 0) raise - there is no torch fallback on the product path.
"""
import torch

from mlhot.binding import MlhotError
from mlhot.ops import LossFunction, NTXentFunction, add_scaled, loss_plus


def nt_xent(z, div, mod, t=0.07):
    """NTXentLoss(temperature=t)(z, labels) of pytorch_metric_learning for labels(i) = (i // div) % mod: for every positive pair
    (a, p)  -log( exp(s_ap / t) / (exp(s_ap / t) + sum over negatives n of a of exp(s_an / t)) ), cosine similarities s, the
    maximum subtracted before the exponentials, `finfo.tiny` added inside the log, averaged over the positive pairs."""
    N, d = z.shape
    if N > 2048 or d > 256 or d % 16:
        raise MlhotError(f"nt_xent: [{N}, {d}] embeddings are outside the kernel's limits (N <= 2048, d <= 256, d % 16 == 0)")
    return NTXentFunction.apply(z, int(div), int(mod), float(t))


class LossFunc:
    def __init__(self, loss_type, task):
        """loss_type: "mse"; task: shapenet_3d | shapenet_1d | pascal_1d | distractor"""
        self.loss_type = loss_type
        self.task = task

    def calc_loss(self, pr_mu, pr_var, gt_y, test=False):
        if self.loss_type != "mse":
            return None   # the reference returns None for any other loss_type (losses.py:33-48)
        if self.task == "shapenet_3d":
            return self.quaternion_loss(gt_y, pr_mu)
        if self.task == "shapenet_1d":
            return self.degree_loss(gt_y, pr_mu) if test else self.azimuth_loss(gt_y, pr_mu)
        if self.task == "pascal_1d":
            return self.mean_square_loss(gt_y, pr_mu)
        if self.task == "distractor":
            return LossFunction.apply("distractor", pr_mu, gt_y)
        return None

    def calc_objective(self, pr_mu, pr_var, gt_y, kl, beta):
        """`calc_loss(pr_mu, pr_var, gt_y) + kl * beta` (the reference's trainer/model_trainer.py:77-78) - the same value and gradients, bit
        for bit; with a KL term on the device the sum rides in the loss's own launches (mlhot.ops.loss_plus: in a replayed step two
        dependent launches instead of four).  Not part of the reference's LossFunc; trainer.ModelTrainer uses it when the loss object
        has it and writes `add_scaled(calc_loss(...), kl, beta)` otherwise."""
        kind = {"shapenet_3d": "quaternion", "shapenet_1d": "azimuth", "pascal_1d": "mse", "distractor": "distractor"}.get(self.task)
        if self.loss_type != "mse" or kind is None:
            return None
        if torch.is_tensor(kl) and kl.is_cuda and kl.numel() == 1 and beta:
            return loss_plus(kind, pr_mu, gt_y, kl, beta)
        return add_scaled(self.calc_loss(pr_mu, pr_var, gt_y), kl, beta)

    def quaternion_loss(self, q_gt, q_pr):
        return LossFunction.apply("quaternion", q_pr, q_gt)

    def azimuth_loss(self, q_gt, q_pr):
        return LossFunction.apply("azimuth", q_pr, q_gt)

    def degree_loss(self, q_gt, q_pr):
        return LossFunction.apply("degree", q_pr, q_gt)

    def mean_square_loss(self, q_gt, q_pr):
        return LossFunction.apply("mse", q_pr, q_gt)

    @staticmethod
    def contrastive_loss(z_1, z_2, t=0.07):
        """Context-set vs target-set task embeddings [T, dim_z] each: embedding i of either set carries label i (losses.py:83-88),
        i.e. labels = [0..T-1, 0..T-1] = row % T."""
        if z_1.shape[0] != z_2.shape[0]:
            raise ValueError("contrastive_loss: both embedding sets hold one row per task")
        return nt_xent(torch.cat((z_1, z_2), dim=0), 1, z_1.shape[0], t)

    @staticmethod
    def contrastive_loss_ANP(z, t=0.07):
        """Per-target attention outputs [T, Nq, d]: the Nq embeddings of task i carry label i (losses.py:91-99) = row // Nq."""
        return nt_xent(z.reshape(-1, z.shape[-1]), z.shape[1], z.shape[0], t)
