"""Where a promoted ModelTrainer iteration's wall time goes with VARIABLE context sizes (one hipGraph per size), c3's model:
per-iteration wall clock and the host time inside _batch / graph replay / _stage_next / loss.item().  usage: python scripts/dev/trainer_iter_probe.py [fixed]"""
import os, sys, time, types, tempfile
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "what-matters-for-meta-learning_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from soak import Data1D
from trainer.losses import LossFunc
from trainer.model_trainer import ModelTrainer
from networks.ANPShapeNet1D import ANPShapeNet1D

DEV = torch.device("cuda:0")
fixed = "fixed" in sys.argv[1:]


bytes_mode = "bytes" in sys.argv[1:]          # images that ARE bytes / 255 (what the reference's loaders hand out): the ExactU8Feed route


class Data(Data1D):
    def get_batch(self, source, tasks_per_batch, shot):
        if bytes_mode and source == "train":
            from mlhot import synth
            pool = self.__dict__.setdefault("_bytes", [])
            if not pool:
                hb = synth.get_batch_u8("shapenet_1d", tasks_per_batch, shot, shot, seed=5)
                pool.append((synth.host_convert(hb[0]), synth.host_convert(hb[1]), hb[2], hb[3]))
            return pool[0]
        if fixed and source == "train":
            pool = self.__dict__.setdefault("_fixed", [])
            if not pool:
                pool.append((*self._make(self.rng, tasks_per_batch, shot)[:1], *self._make(self.rng, tasks_per_batch, shot)[:1]))
                cx, cy = self._make(self.rng, tasks_per_batch, shot); qx, qy = self._make(self.rng, tasks_per_batch, shot)
                pool[0] = (cx, qx, cy, qy)
            return pool[0]
        return super().get_batch(source, tasks_per_batch, shot)


with tempfile.TemporaryDirectory() as tmp:
    os.chdir(tmp)
    cfg = types.SimpleNamespace(device=DEV, seed=2578, img_size=[128, 128, 1], tasks_per_batch=16, input_dim=3, output_dim=2, agg_mode="attention",
                                img_agg="", dim_w=64, n_hidden_units_r=[100, 100], dim_r=64, dim_z=64, task="shapenet_1d", iterations=400, val_freq=10 ** 9,
                                val_iters=1, bg_gen_freq=10 ** 9, gen_bg=False, max_ctx_num=15, beta=0, contrastive=False, log_every=1, save_path=tmp, logger=None)
    model = ANPShapeNet1D(cfg).to(DEV)
    tr = ModelTrainer(model=model, loss=LossFunc("mse", "shapenet_1d"), optimizer=torch.optim.Adam(model.parameters(), lr=1e-3), config=cfg, data=Data())
    tr.train()                                   # captures
    acc = {}

    def timed(name, fn):
        def w(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return w
    tr._batch = timed("_batch", tr._batch)
    tr._stage_next = timed("_stage_next", tr._stage_next)
    graphs = [v for v in tr._graphs.values() if isinstance(v, tuple)]
    for g in graphs:
        g[0].replay = timed("graph.replay", g[0].replay)
    tr.iterations = 10 ** 9
    n = 1000
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(401, 401 + n):
        tr._prefetch = True
        tr._train_iter(it)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"{'fixed 15+15' if fixed else 'context sizes 3..15'}: {len(graphs)} graphs, {1e3 * wall / n:.3f} ms per iteration; host time inside: "
          + ", ".join(f"{k} {1e3 * v / n:.3f} ms" for k, v in acc.items()))
