"""LossFunc with the reference's interface (trainer/losses.py:22-80); the arithmetic is the
fused mlhot loss kernels (forward reduction + backward seed in one launch each)."""
from mlhot.ops import LossFunction


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
