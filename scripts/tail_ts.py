#!/usr/bin/env python3
"""Stage timestamps inside the fused tail kernels (latency hunting, not a benchmark).

Needs the instrumented build:  hipcc ... -DMLHOT_TS mlhot.hip -o csrc/libmlhot_ts.so, then
    MLHOT_LIB=.../libmlhot_ts.so python scripts/tail_ts.py
Workgroup 0 / thread 0 of each instrumented kernel stores wall_clock64() (100 MHz) at stage boundaries.
Reading the numbers: a stamp loads the buffer pointer and waits vmcnt(0) before it stores, i.e. it also waits for every global load
the kernel has in flight at that point (e.g. the conv3 forward's "first unit staged" stamp sits right after the next unit's prefetch
was issued and shows that round trip, ~2.6 k cycles, which the product build does not wait for).
"""
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch  # noqa: E402
import bench  # noqa: E402
import mlhot  # noqa: E402
from mlhot import synth  # noqa: E402
from trainer.losses import LossFunc  # noqa: E402

dev = torch.device("cuda", 0)
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(bench.make_cfg(w, dev)).to(dev)
loss_fn = LossFunc("mse", "shapenet_1d")
cx, qx, cy, qy = synth.get_batch("shapenet_1d", 16, 15, 15, seed=1234, device=dev)
ts = torch.zeros(512, dtype=torch.int64, device=dev)
L = mlhot.lib()
for kv in filter(None, os.environ.get("MLHOT_OPTS", "").split(",")):       # e.g. MLHOT_OPTS=conv2_split=7
    L.set_option(kv.split("=")[0], int(kv.split("=")[1]))
L.c.mlhot_dbg_tsbuf.argtypes = [ctypes.c_void_p]
assert L.c.mlhot_dbg_tsbuf(ts.data_ptr()) == 0
acc = None
N = 10
for it in range(N + 3):
    model.zero_grad(set_to_none=True)
    loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()
    torch.cuda.synchronize()
    if it >= 3:
        v = ts.cpu().double()
        calls = ts.cpu()[200:200 + 8 * 36].view(36, 8).tolist()
        acc = v if acc is None else acc + v
    ts[199] = 0
v = (acc / N).tolist()
for base, name in ((0, "A.fwd"), (16, "A.fwd.keyhead"), (32, "B.fwd"), (64, "C.fwd"), (96, "C.bwd"), (128, "B.bwd"), (160, "A.bwd")):
    seg = [(i, v[base + i]) for i in range(16 if base < 32 else 32) if v[base + i] > 0]
    if len(seg) < 2:
        continue
    print(name, "total %.1f us:" % ((seg[-1][1] - seg[0][1]) / 100.0),
          " ".join("%d:%.1f" % (i, (t - seg[k - 1][1]) / 100.0) for k, (i, t) in enumerate(seg) if k))

rwts = ts.cpu()[300:321].tolist()
if rwts[0]:
    print("trunk conv3x3 (16x16 stride-1, workgroup 0) cycles from kernel entry: prologue done %d; per band [stage start, staged, MFMAs done, stored]:" % (rwts[1] - rwts[0]),
          [[rwts[2 + 4 * k + i] - rwts[0] for i in range(4)] for k in range(4) if rwts[2 + 4 * k]], "exit", rwts[20] - rwts[0])
blk = [(v[220 + i] - v[200 + i]) / 100.0 for i in range(16) if v[220 + i] > 0]
if blk:
    print("A.bwd per block (us):", " ".join("%.1f" % x for x in blk))
print("layer-function calls of workgroup 0 (cycles between stamps: entry | setup | k-loop | fold | epilogue | exit)")
for c, row in enumerate(calls):
    if row[0] and row[5]:
        print(c, [row[i + 1] - row[i] for i in range(5)], "total", row[5] - row[0])

if v[500] > 0:
    print("conv12 fwd main loop of workgroup 0: %.0f shader cycles in %.1f us -> effective clock %.2f GHz" % (v[500], v[501] / 100.0, v[500] / (v[501] * 10.0) ))

# per-wave band timeline of the forward conv12 kernel (7th band of workgroup 0): cycles from the earliest wave's start
raw = ts.cpu()[400:448].view(12, 4).tolist()
if raw[0][0]:
    t0 = min(r[0] for r in raw)
    print("per-wave band timeline [start, MFMA+conv1 done, epilogue done, barrier passed] (cycles):")
    for w, r in enumerate(raw):
        print("  wave %2d" % w, [x - t0 for x in r])

# the same for the conv12 data-gradient kernel (8 waves)
raw = ts.cpu()[448:496].view(8, 6).tolist()
if raw[0][0]:
    t0 = min(r[0] for r in raw)
    print("conv12 dgrad per-wave band timeline [start, row 0 done, next band staged, row 1 done, barrier passed] (cycles):")
    for w, r in enumerate(raw):
        print("  wave %2d" % w, [x - t0 for x in r[:5]])

raw = ts.cpu()[360:420].view(12, 5).tolist()
if raw[0][0]:
    t0 = min(r[0] for r in raw)
    print("conv12 wgrad per-wave band timeline (cycles) - fp32 kernel: [previous band's MFMAs done, barrier passed, conv1 + dY staged, barrier passed, MFMAs done]; "
          "split kernel: [band start, conv1 tiles done, dY cell stored, barrier passed, MFMAs done]:")
    for w, r in enumerate(raw):
        print("  wave %2d" % w, [x - t0 for x in r])
c3t = ts.cpu()[340:358].tolist()
if c3t[0]:
    print("conv3 fwd prologue: weight loads landed %d, staged in LDS (barrier passed) %d; after the gather: halo zeroed %d, wave 0 stashed %d" % (c3t[14] - c3t[0], c3t[13] - c3t[0], c3t[15] - c3t[0], c3t[16] - c3t[0]))
    print("conv3 fwd, workgroup 0 (cycles from entry): weights in registers %d, first unit staged %d, units [MFMAs + store done, barrier passed]:" % (c3t[1] - c3t[0], c3t[2] - c3t[0]),
          [[c3t[3 + 2 * k] - c3t[0], c3t[4 + 2 * k] - c3t[0]] for k in range(4) if c3t[3 + 2 * k]], "exit", c3t[12] - c3t[0])
pro = ts.cpu()[502:507].tolist()
if pro[0]:
    print("prologues of workgroup 0 (cycles from kernel entry to the band loop): conv12 fwd %d, conv12 dgrad %d (its band loop: %d)" % (pro[1] - pro[0], pro[3] - pro[2], pro[4] - pro[3]))

# effective clock under back-to-back hipGraph replay (what bench.py times)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
model.zero_grad(set_to_none=True)
with torch.cuda.graph(graph):
    loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy).backward()
for _ in range(200):
    graph.replay()
torch.cuda.synchronize()
v = ts.cpu().double().tolist()
if v[500] > 0:
    print("graph replay: conv12 fwd main loop %.0f shader cycles in %.1f us -> effective clock %.2f GHz" % (v[500], v[501] / 100.0, v[500] / (v[501] * 10.0)))
