"""Shape-faithful synthetic meta-batches (the datasets are git-LFS pointers in the reference).

Layouts follow dataset/shapenet_1d.py:189-196 / pascal_1d.py get_batch: float32 tensors
ctx_x [T,Nc,C,H,W], qry_x [T,Nq,C,H,W] in [0,1) (the reference divides uint8 by 255),
labels [T,N,L]: shapenet_1d L=3 = [cos a, sin a, a] with a ~ U[0,2pi); pascal_1d L=1 ~ U[0,1)."""
import math

import torch


def get_batch(task, tasks_per_batch, n_ctx, n_qry, seed=1234, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    xs = torch.rand(tasks_per_batch, n_ctx, 1, 128, 128, generator=g)
    xq = torch.rand(tasks_per_batch, n_qry, 1, 128, 128, generator=g)
    if task == "shapenet_1d":
        a_s = torch.rand(tasks_per_batch, n_ctx, 1, generator=g) * 2 * math.pi
        a_q = torch.rand(tasks_per_batch, n_qry, 1, generator=g) * 2 * math.pi
        ys = torch.cat([torch.cos(a_s), torch.sin(a_s), a_s], dim=-1)
        yq = torch.cat([torch.cos(a_q), torch.sin(a_q), a_q], dim=-1)
    elif task == "pascal_1d":
        ys = torch.rand(tasks_per_batch, n_ctx, 1, generator=g)
        yq = torch.rand(tasks_per_batch, n_qry, 1, generator=g)
    else:
        raise ValueError(task)
    return tuple(t.to(device) for t in (xs, xq, ys, yq))


def get_batch_u8(task, tasks_per_batch, n_ctx, n_qry, seed=1234):
    """The same batch shapes as the loaders hold them BEFORE their host-side conversion (dataset/shapenet_1d.py:174-188):
    uint8 channel-last images [T,N,128,128,1] as numpy arrays + the fp32 labels.  Feed to mlhot.ingest.BatchIngest."""
    g = torch.Generator().manual_seed(seed)
    xs = torch.randint(0, 256, (tasks_per_batch, n_ctx, 128, 128, 1), generator=g, dtype=torch.uint8)
    xq = torch.randint(0, 256, (tasks_per_batch, n_qry, 128, 128, 1), generator=g, dtype=torch.uint8)
    _, _, ys, yq = get_batch(task, tasks_per_batch, n_ctx, n_qry, seed=seed)
    return xs.numpy(), xq.numpy(), ys, yq


def host_convert(u8):
    """What the reference's loaders do on the host (shapenet_1d.py:189-190 + utils/utils.py:26-30): the pageable-fp32
    route that `get_batch_u8` + mlhot.ingest.BatchIngest replaces (kept for A/B timing in bench.py)."""
    import numpy as np
    x = torch.from_numpy(np.asarray(u8).astype(np.float32) / 255.0).type(torch.FloatTensor)
    return x.permute(0, 1, 4, 2, 3).contiguous()


def get_batch_3d(tasks_per_batch, n_ctx, n_qry, seed=1234, device="cpu", task_aug=True):
    """ShapeNet3D-shaped meta-batch (BASELINE config c5; dataset/shapenet_3d.py:108-122,218-227): images [T, N, 3, 64, 64] in
    [0, 1) (the alpha channel is dropped by the loader), labels = unit quaternions with q[1] >= 0.  `task_aug`: the loader's
    task augmentation (utils/utils.py:33-58) - per task one random azimuth / elevation offset added to every label of the task
    (context and target alike) through Euler angles; host-side label arithmetic, it does not change any device shape.  The
    image augmentation (imgaug) of the loader is out of scope."""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    xs = torch.rand(tasks_per_batch, n_ctx, 3, 64, 64, generator=g)
    xq = torch.rand(tasks_per_batch, n_qry, 3, 64, 64, generator=g)

    def quats(n):
        q = torch.nn.functional.normalize(torch.randn(tasks_per_batch, n, 4, generator=g), dim=-1)
        return torch.where(q[..., 1:2] < 0, -q, q)
    ys, yq = quats(n_ctx), quats(n_qry)
    if task_aug:
        from scipy.spatial.transform import Rotation as R
        rng = np.random.RandomState(seed)
        out = []
        for i in range(tasks_per_batch):
            d_az, d_el = rng.randint(-10, 20), rng.randint(-5, 10)
            pair = []
            for q in (ys[i], yq[i]):
                e = R.from_quat(q.numpy().astype(np.float64)).as_euler("ZYX", degrees=True)
                e[:, 0] += d_el
                e[:, 2] -= d_az
                pair.append(torch.from_numpy(R.from_euler("ZYX", e, degrees=True).as_quat().astype(np.float32)))
            out.append(pair)
        ys, yq = torch.stack([p_[0] for p_ in out]), torch.stack([p_[1] for p_ in out])
    return tuple(t.to(device) for t in (xs, xq, ys, yq))


class SyntheticData:
    """Minimal stand-in for dataset.ShapeNet1D / Pascal1D with the reference's `get_batch` contract
    (dataset/shapenet_1d.py:113-196): train batches draw a random context size in [3, shot], validation /
    test batches use `shot` context images; the target count is always `shot`."""

    def __init__(self, task="shapenet_1d", seed=42):
        import numpy as np
        self.task, self.test_counter = task, 0
        self.rng = np.random.RandomState(seed)
        self.val_rng, self.test_rng = np.random.RandomState(seed), np.random.RandomState(seed)
        self._step = 0

    def gen_bg(self, config, data="all"):
        pass

    def get_batch(self, source, tasks_per_batch, shot):
        rng = {"train": self.rng, "validation": self.val_rng, "test": self.test_rng}[source]
        n_ctx = int(rng.randint(3, shot + 1)) if source == "train" else shot
        self._step += 1
        return get_batch(self.task, tasks_per_batch, n_ctx, shot, seed=int(rng.randint(0, 2 ** 31 - 1)))

    def get_batch_u8(self, source, tasks_per_batch, shot):
        """`get_batch` before the host-side conversion: (ctx uint8 [T,Nc,H,W,C], qry uint8, ctx labels, qry labels)."""
        rng = {"train": self.rng, "validation": self.val_rng, "test": self.test_rng}[source]
        n_ctx = int(rng.randint(3, shot + 1)) if source == "train" else shot
        self._step += 1
        return get_batch_u8(self.task, tasks_per_batch, n_ctx, shot, seed=int(rng.randint(0, 2 ** 31 - 1)))
