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


lagged = "lagged" in sys.argv[1:]            # config.lagged_loss_log: the loss of iteration k is read behind the launch of k + 1
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
                                val_iters=1, bg_gen_freq=10 ** 9, gen_bg=False, max_ctx_num=15, beta=0, contrastive=False, log_every=1, save_path=tmp, logger=None,
                                lagged_loss_log=lagged, close_after_train=False)
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
    tr._flush_loss = timed("_flush_loss (event wait + log)", tr._flush_loss)
    if tr._host_prefetch is not None:
        tr._host_prefetch.take = timed("prefetch.take (worker's future + stream wait)", tr._host_prefetch.take)
    graphs = [v for v in tr._graphs.values() if isinstance(v, tuple)]
    for g in graphs:
        g[0].replay = timed("graph.replay", g[0].replay)
    tr.iterations = 10 ** 9
    n = 1000
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    inner = tr._graph_train_iter

    def stamped(it):                              # device time of one iteration's own work: an event in front of its first launch, one behind its last
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = inner(it)
        b.record()
        evs.append((a, b))
        return r
    if "stamps" in sys.argv[1:]:
        tr._graph_train_iter = stamped
    marks = [time.perf_counter()]
    for it in range(401, 401 + n):
        tr._prefetch = 2      # what train() sets far from a validation round: two batches may be drawn ahead
        tr._train_iter(it)
        marks.append(time.perf_counter())
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = 1e3 * np.diff(np.array(marks))
    print(f"host: per iteration median {np.median(per):.3f} ms, mean {per.mean():.3f}, p90 {np.percentile(per, 90):.3f}, p99 {np.percentile(per, 99):.3f}, "
          f"max {per.max():.3f} (at iteration {int(per.argmax())} of {n}); the {int((per > 3 * np.median(per)).sum())} iterations above 3 x median hold {per[per > 3 * np.median(per)].sum():.1f} ms")
    if evs:
        busy = np.array([a.elapsed_time(b) for a, b in evs[50:]])
        gap = np.array([evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(50, len(evs) - 1)])
        print(f"device: an iteration's own work (first launch -> last) median {np.median(busy):.3f} ms (mean {busy.mean():.3f}, p10 {np.percentile(busy, 10):.3f}, p90 {np.percentile(busy, 90):.3f}, "
              f"p99 {np.percentile(busy, 99):.3f}, max {busy.max():.3f}); between iterations median {np.median(gap):.3f} ms (mean {gap.mean():.3f}, p90 {np.percentile(gap, 90):.3f}, p99 {np.percentile(gap, 99):.3f}, max {gap.max():.3f})")
    tr.close()
    print(f"{'fixed 15+15' if fixed else 'context sizes 3..15'}{', bytes' if bytes_mode else ''}{', lagged log' if lagged else ''}: {len(graphs)} graphs, {1e3 * wall / n:.3f} ms per iteration; host time inside: "
          + ", ".join(f"{k} {1e3 * v / n:.3f} ms" for k, v in acc.items()))
