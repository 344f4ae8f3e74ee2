"""Soak: trainer.ModelTrainer's promoted loop (flat Adam, hipGraph replay, host prefetch / staged eps on host threads) for thousands of
iterations with validation rounds in between, on synthetic data with LEARNABLE labels (a fixed linear read-out of the image means), for
c3's model (ANPShapeNet1D, random context sizes 3..15 -> 13 graphs) and c5's (ANPMRShapeNet3D, 2 tasks).  Prints the loss at the start /
end, iterations per second, and checks that every loss is finite and that the training loss went down.
usage: python scripts/dev/soak.py [iterations_c3] [iterations_c5]"""
import os, sys, time, types, tempfile
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "what-matters-for-meta-learning_amd"))
from trainer.losses import LossFunc
from trainer.model_trainer import ModelTrainer

DEV = torch.device("cuda:0")


class Data1D:
    """images whose label is a function of the image: angle = 2 pi * mean of the top-left quadrant (learnable, unlike random labels)"""
    def __init__(self):
        self.rng, self.val = np.random.RandomState(1), np.random.RandomState(2)

    def gen_bg(self, *a, **k):
        pass

    def _make(self, rng, T, n):
        g = torch.Generator().manual_seed(int(rng.randint(0, 2 ** 31 - 1)))
        x = torch.rand(T, n, 1, 128, 128, generator=g)
        x[..., :64, :64] *= torch.rand(T, n, 1, 1, 1, generator=g)
        a = x[..., :64, :64].mean(dim=(2, 3, 4), keepdim=False).unsqueeze(-1) * 4 * 3.14159265
        return x, torch.cat([torch.cos(a), torch.sin(a), a], dim=-1)

    def get_batch(self, source, tasks_per_batch, shot):
        """a pool of 48 training / 4 validation batches made once and handed out in turn (synthesising 8 M random floats per batch on
        the host would take 100x the step)"""
        pool = self.__dict__.setdefault("_pool_" + source, [])
        want = 48 if source == "train" else 4
        if len(pool) < want:
            rng = self.rng if source == "train" else self.val
            n_ctx = int(rng.randint(3, shot + 1)) if source == "train" else shot
            cx, cy = self._make(rng, tasks_per_batch, n_ctx)
            qx, qy = self._make(rng, tasks_per_batch, shot)
            pool.append((cx, qx, cy, qy))
            return pool[-1]
        k = self.__dict__.get("_next_" + source, 0)
        self.__dict__["_next_" + source] = k + 1
        return pool[k % want]


class Data3D(Data1D):
    def _make(self, rng, T, n):
        g = torch.Generator().manual_seed(int(rng.randint(0, 2 ** 31 - 1)))
        x = torch.rand(T, n, 3, 64, 64, generator=g)
        s = torch.rand(T, n, 4, 1, 1, generator=g)
        for c in range(3):
            x[:, :, c] *= s[:, :, c]
        q = torch.cat([s[:, :, :3, 0, 0] - 0.5, torch.ones(T, n, 1) * 0.5], dim=-1)
        q = torch.nn.functional.normalize(q, dim=-1)
        return x, torch.where(q[..., 1:2] < 0, -q, q)


def run(method, cfg, data, loss, iters):
    import importlib
    model = getattr(importlib.import_module("networks." + method), method)(cfg).to(DEV)
    tr = ModelTrainer(model=model, loss=loss, optimizer=torch.optim.Adam(model.parameters(), lr=1e-3), config=cfg, data=data)
    seen = []
    orig = tr._report                             # every iteration's loss, whichever call hands it out (the late read: one iteration behind)
    tr._report = lambda it, v: (seen.append(v), orig(it, v))[1]
    torch.manual_seed(7)
    t0 = time.perf_counter()
    tr.train()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(seen) == iters and all(v is not None and np.isfinite(v) for v in seen), "non-finite or missing loss"
    first, last = float(np.mean(seen[:50])), float(np.mean(seen[-50:]))
    graphs = len([v for v in tr._graphs.values() if isinstance(v, tuple)])
    print(f"{method}: {iters} iterations in {dt:.1f} s ({iters / dt:.0f} it/s incl. {iters // cfg.val_freq} validation rounds and the captures), "
          f"{graphs} graphs, optimizer {type(tr.optimizer).__name__}, mean loss of the first 50 iterations {first:.4f}, of the last 50 {last:.4f}", flush=True)
    assert last < first, "the training loss did not go down"


if __name__ == "__main__":
    n3, n5 = (int(sys.argv[1]) if len(sys.argv) > 1 else 3000), (int(sys.argv[2]) if len(sys.argv) > 2 else 600)
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        cfg = types.SimpleNamespace(device=DEV, seed=2578, img_size=[128, 128, 1], tasks_per_batch=16, input_dim=3, output_dim=2, agg_mode="attention",
                                    img_agg="", dim_w=64, n_hidden_units_r=[100, 100], dim_r=64, dim_z=64, task="shapenet_1d", iterations=n3, val_freq=500,
                                    val_iters=2, bg_gen_freq=10 ** 9, gen_bg=False, max_ctx_num=15, beta=0, contrastive=False, log_every=1,
                                    save_path=tmp + "/c3", logger=None)
        run("ANPShapeNet1D", cfg, Data1D(), LossFunc("mse", "shapenet_1d"), n3)
        cfg = types.SimpleNamespace(device=DEV, seed=2578, img_size=[64, 64, 4], tasks_per_batch=2, input_dim=4, output_dim=4, agg_mode="attention",
                                    img_agg="reshape", task="shapenet_3d", temperature=0.07, iterations=n5, val_freq=200, val_iters=2, bg_gen_freq=10 ** 9,
                                    gen_bg=False, max_ctx_num=8, beta=1e-7, contrastive=False, log_every=1, save_path=tmp + "/c5", logger=None)
        run("ANPMRShapeNet3D", cfg, Data3D(), LossFunc("mse", "shapenet_3d"), n5)
    print("soak ok")
