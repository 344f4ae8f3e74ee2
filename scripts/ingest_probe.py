#!/usr/bin/env python3
"""Where the time of BatchIngest.stage()/take() goes (host memcpy into pinned staging, H2D, ingest kernels)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mlhot import synth  # noqa: E402
from mlhot.ingest import BatchIngest  # noqa: E402

dev = torch.device("cuda", 0)
hb = synth.get_batch_u8("shapenet_1d", 16, 15, 15, seed=1)
src = torch.from_numpy(hb[0])
pin = torch.empty_like(src).pin_memory()
page = torch.empty_like(src)
d = torch.empty_like(src, device=dev)


def t(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


print("threads", torch.get_num_threads(), "MB", src.numel() / 1e6)
print("torch copy_ pageable->pageable %.3f ms" % t(lambda: page.copy_(src)))
print("torch copy_ pageable->pinned   %.3f ms" % t(lambda: pin.copy_(src)))
pn, sn = pin.numpy(), src.numpy()
print("numpy copyto pageable->pinned  %.3f ms" % t(lambda: np.copyto(pn, sn)))
print("H2D pinned u8 (non_blocking)   %.3f ms" % t(lambda: d.copy_(pin, non_blocking=True)))
print("H2D pageable u8                %.3f ms" % t(lambda: d.copy_(src)))
ing = BatchIngest(dev)
ing.stage(*hb)
ing.take()


def st():
    ing.stage(*hb)
    ing.take()


print("stage+take                     %.3f ms" % t(st))
t0 = time.perf_counter()
for _ in range(20):
    ing.stage(*hb)
    t1 = time.perf_counter()
    ing.take()
    t2 = time.perf_counter()
print("last: stage %.3f ms (host), take %.3f ms (host)" % (1e3 * (t1 - t0) / 20, 1e3 * (t2 - t1)))

# ---- the bench loop, host time per call --------------------------------------------------------------------
import importlib  # noqa: E402
import bench  # noqa: E402
from trainer.losses import LossFunc  # noqa: E402
w = bench.WORKLOADS["c3"]
model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(bench.make_cfg(w, dev)).to(dev)
loss_fn = LossFunc("mse", "shapenet_1d")
ing = BatchIngest(dev)
ing.stage(*hb)
cx, qx, cy, qy = ing.take()


def step():
    model.zero_grad(set_to_none=True)
    loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()


graph = bench._capture(step)
for label, run_step in (("graph", graph.replay), ("eager", step), ("none", lambda: None)):
    ing.stage(*hb)
    acc = [0.0, 0.0, 0.0]
    torch.cuda.synchronize()
    T0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter()
        ing.take()
        b = time.perf_counter()
        run_step()
        c = time.perf_counter()
        ing.stage(*hb)
        e = time.perf_counter()
        acc[0] += b - a
        acc[1] += c - b
        acc[2] += e - c
    torch.cuda.synchronize()
    tot = time.perf_counter() - T0
    ing.take()
    torch.cuda.synchronize()
    print("%-6s total %.3f ms/iter; host: take %.3f step %.3f stage %.3f" % (label, 1e3 * tot / 20, *(1e3 * x / 20 for x in acc)))
