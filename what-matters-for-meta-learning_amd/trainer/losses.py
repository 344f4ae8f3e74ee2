"""LossFunc with the reference's interface (trainer/losses.py:22-99); the regression losses are the fused mlhot loss
kernels (forward reduction + backward seed in one launch each).

The functional-contrastive term (losses.py:82-99) is NT-Xent from the un-vendored dependency `pytorch_metric_learning`
(listed without a version in the reference's requirements.txt:12; `NTXentLoss(temperature=t)` with its defaults: cosine
similarity, every same-label pair a positive, every different-label pair a negative of its anchor, mean over the positive
pairs).  `nt_xent` below restates that published algorithm on device tensors with torch operators - a few [2T x 2T] /
[T*Nq x T*Nq] matrices, plumbing-sized work next to the encoders; autograd supplies its backward.
"""
import numpy as np
import torch
import torch.nn.functional as F

from mlhot.ops import LossFunction


def nt_xent(z, labels, t=0.07):
    """NTXentLoss(temperature=t)(z, labels) of pytorch_metric_learning: for every positive pair (a, p)
    -log( exp(s_ap / t) / (exp(s_ap / t) + sum over negatives n of a of exp(s_an / t)) ), cosine similarities s, the row
    maximum subtracted before the exponentials, `finfo.tiny` added inside the log, averaged over the positive pairs.
    `labels`: host integers (the reference builds them with torch.arange on the CPU), so the pair lists cost no device sync."""
    lab = np.asarray(labels.cpu() if torch.is_tensor(labels) else labels).reshape(-1)
    same = lab[:, None] == lab[None, :]
    a1, p = np.nonzero(same & ~np.eye(len(lab), dtype=bool))
    if len(a1) == 0 or same.all():
        return z.sum() * 0.0
    dev = z.device
    a1_t, p_t = torch.from_numpy(a1).to(dev), torch.from_numpy(p).to(dev)
    neg_mask = torch.from_numpy(~same[a1]).to(dev)                       # [P, N]: the negatives of each positive pair's anchor
    zn = F.normalize(z, p=2, dim=1)
    sim = zn @ zn.t()
    pos = sim[a1_t, p_t].unsqueeze(1) / t
    neg = (sim / t)[a1_t].masked_fill(~neg_mask, torch.finfo(z.dtype).min)
    max_val = torch.max(pos, neg.max(dim=1, keepdim=True)[0]).detach()
    num = torch.exp(pos - max_val).squeeze(1)
    den = torch.exp(neg - max_val).sum(dim=1) + num
    return (-torch.log(num / den + torch.finfo(z.dtype).tiny)).mean()


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
        """Context-set vs target-set task embeddings [T, dim_z] each: embedding i of either set carries label i (losses.py:83-88)."""
        z = torch.cat((z_1, z_2), dim=0)
        labels = np.concatenate((np.arange(z_1.shape[0]), np.arange(z_2.shape[0])))
        return nt_xent(z, labels, t)

    @staticmethod
    def contrastive_loss_ANP(z, t=0.07):
        """Per-target attention outputs [T, Nq, d]: the Nq embeddings of task i carry label i (losses.py:91-99)."""
        labels = np.repeat(np.arange(z.shape[0]), z.shape[1])
        return nt_xent(z.reshape(-1, z.shape[-1]), labels, t)
