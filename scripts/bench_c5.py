#!/usr/bin/env python3
"""Informational timing of BASELINE config c5's per-GPU step (ANPMRShapeNet3D, 8 tasks x (15+15) 3x64x64 images):
ResNet / Bayes-by-backprop path on the run-time-shaped kernels.  Not the headline metric (bench.py)."""
import importlib, os, sys, time, types, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch
import mlhot
from trainer.losses import LossFunc
dev = torch.device("cuda", 0)
T, NC, NQ = 8, 15, 15
cfg = types.SimpleNamespace(device=dev, seed=2578, img_size=[64, 64, 4], tasks_per_batch=T, input_dim=4, output_dim=4,
                            agg_mode="attention", img_agg="reshape", task="shapenet_3d", temperature=0.07, method="ANPMRShapeNet3D")
model = importlib.import_module("networks.ANPMRShapeNet3D").ANPMRShapeNet3D(cfg).to(dev)
g = torch.Generator().manual_seed(1234)
cx, qx = torch.rand(T, NC, 3, 64, 64, generator=g).to(dev), torch.rand(T, NQ, 3, 64, 64, generator=g).to(dev)
cy = torch.nn.functional.normalize(torch.randn(T, NC, 4, generator=g), dim=-1).to(dev)
qy = torch.nn.functional.normalize(torch.randn(T, NQ, 4, generator=g), dim=-1).to(dev)
loss_fn = LossFunc("mse", "shapenet_3d")
def step():
    model.zero_grad(set_to_none=True)
    mu, var, kl = model(cx, cy, qx)
    (loss_fn.calc_loss(mu, var, qy) + 1e-7 * kl).backward()
for _ in range(3): step()
torch.cuda.synchronize()
L = mlhot.lib(); L.prof_begin(65536)
t0 = time.perf_counter()
N = 10
for _ in range(N): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
recs = L.prof_end()
agg = {}
for label, ms in recs:
    a = agg.setdefault(label, [0, 0.0]); a[0] += 1; a[1] += ms
top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]
# ---- the same step as a hipGraph: eps pre-drawn on the CPU generator in the recorded order (networks/bbb/eps.py) ----------
from networks.bbb.eps import StagedEps
import bench
eps = StagedEps(dev)
with eps.recording():
    step()
eps.stage()
def staged_step():
    eps.rewind()
    step()


with eps.active():
    graph = bench._capture(staged_step)
draw_ms = 0.0
for timed in (False, True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        d0 = time.perf_counter()
        eps.stage()
        draw_ms += (time.perf_counter() - d0) * 1e3 if timed else 0.0
        graph.replay()
    torch.cuda.synchronize()
    graph_dt = (time.perf_counter() - t0) / N
print(json.dumps({"workload": "c5 per-GPU: ANPMRShapeNet3D T=8 15+15 3x64x64", "eager_ms_per_step": dt * 1e3, "tasks_per_s_eager": T / dt,
                  "hipgraph_ms_per_step": graph_dt * 1e3, "tasks_per_s": T / graph_dt, "host_eps_draw_ms_per_step": draw_ms / N,
                  "eps_floats_per_step": eps._total,
                  "gpu_busy_ms_per_step": sum(v[1] for v in agg.values()) / N, "launches_per_step": sum(v[0] for v in agg.values()) / N,
                  "top_kernels_ms_per_step": {k: round(v[1] / N, 3) for k, v in top}}))
