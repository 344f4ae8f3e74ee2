"""extras.shipped_cfg's loop (bench.measure_shipped_cfg: the reference's shipped training shape through trainer.ModelTrainer) with every
iteration stamped on the device: is an iteration the GPU step or the host's turn?  usage: python scripts/dev/shipped_iter_probe.py [c3|c5]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench                                                  # noqa: E402
from trainer.model_trainer import ModelTrainer               # noqa: E402
from trainer.losses import LossFunc                           # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "c3"
w = dict(bench.WORKLOADS[key], key=key)
dev = torch.device("cuda:0")
evs, host = [], []
inner = ModelTrainer._graph_train_iter


def stamped(self, it):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    r = inner(self, it)
    b.record()
    evs.append((a, b)); host.append(time.perf_counter() - t0)
    return r


ModelTrainer._graph_train_iter = stamped
out = bench.measure_shipped_cfg(w, dev, LossFunc("mse", w["task"]), 100)
torch.cuda.synchronize()
n = out["iterations"]
last = evs[-n:]
busy = np.array([a.elapsed_time(b) for a, b in last])
gap = np.array([last[i][1].elapsed_time(last[i + 1][0]) for i in range(n - 1)])
print(f"{key} shipped shape: {out['ms_per_iter']:.3f} ms per iteration ({out['tasks_per_s']:.0f} tasks/s, {out['graphs']} graphs); device: own work median {np.median(busy):.3f} "
      f"(mean {busy.mean():.3f}, p90 {np.percentile(busy, 90):.3f}), between iterations median {np.median(gap):.3f} (mean {gap.mean():.3f}, p90 {np.percentile(gap, 90):.3f}); "
      f"host inside _graph_train_iter mean {1e3 * np.mean(host[-n:]):.3f} ms")
