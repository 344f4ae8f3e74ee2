#!/usr/bin/env python3
"""Per-CU timeline of ONE trunk convolution launch (every workgroup stamped, -DMLHOT_TS build, RW_TSALL in csrc/resnet_ws.h): which two
workgroups share a CU, when each is staging / on the matrix pipe / in its epilogue, and how much of the launch a CU spends with NO
workgroup in its MFMA phase.  Geometry = RW_TS_HIN / RW_TS_S of the build (default build below: 32 / 2 = block 1's conv1 + 3x3 skip).
    MLHOT_LIB=build_exp/libmlhot_ts32.so python scripts/dev/trunk_cu_timeline.py"""
import ctypes, os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch
import mlhot
from mlhot import synth
from networks.ANPMRShapeNet3D import ANPMRShapeNet3D
dev = torch.device("cuda:0")
T = 8
cfg = types.SimpleNamespace(device=dev, seed=2578, img_size=[64, 64, 4], tasks_per_batch=T, input_dim=4, output_dim=4,
                            agg_mode="attention", img_agg="reshape", task="shapenet_3d", temperature=0.07)
model = ANPMRShapeNet3D(cfg).to(dev)
cx, qx, cy, qy = synth.get_batch_3d(T, 15, 15, seed=4321, device=dev, task_aug=True)
NW = 1024
ts = torch.zeros(4096 + 32 * NW, dtype=torch.int64, device=dev)
L = mlhot.lib()
L.c.mlhot_dbg_tsbuf.argtypes = [ctypes.c_void_p]
assert L.c.mlhot_dbg_tsbuf(ts.data_ptr()) == 0
for it in range(6):
    ts.zero_()
    with torch.no_grad():
        model(cx, cy, qx)
    torch.cuda.synchronize()
r = ts.cpu().numpy()[4096:].reshape(NW, 32)
r = r[r[:, 1] > 0]
n = len(r)
t0 = r[:, 1].min()
hw = r[:, 0] & 0xffffffff
xcc = (r[:, 0] >> 32) & 0xf
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 20 + cu
us = lambda v: (v - t0) / 100.0
print(f"{n} workgroups on {len(set(key.tolist()))} CUs; launch span {us(r[:, 31].max()):.1f} us (first entry -> last exit); entries spread over {us(r[:, 1].max()):.1f} us")
pro = (r[:, 2] - r[:, 1]) / 100.0
print(f"prologue (weights, patch zeroing): mean {pro.mean():.2f} us, max {pro.max():.2f}")
st, mf, ep, nb = [], [], [], []
for w in r:
    prev = w[2]
    k = 0
    for b in range(7):
        a, m, s = w[3 + 3 * b], w[4 + 3 * b], w[5 + 3 * b]
        if a == 0:
            break
        st.append((a - prev) / 100.0); mf.append((m - a) / 100.0); ep.append((s - m) / 100.0)
        prev = s
        k += 1
    nb.append(k)
st, mf, ep = np.array(st), np.array(mf), np.array(ep)
print(f"bands per workgroup: {np.bincount(nb).tolist()} (index = count); per band: staging {st.mean():.2f} us (p90 {np.percentile(st, 90):.2f}), "
      f"MFMA phase {mf.mean():.2f} us (min {mf.min():.2f}, p90 {np.percentile(mf, 90):.2f}), epilogue {ep.mean():.2f} us (p90 {np.percentile(ep, 90):.2f})")
# per CU: union of the MFMA phases of its workgroups, and the time both / none are in it
span_end = r[:, 31].max()
idle_tot, both_tot, one_tot, ncu = 0.0, 0.0, 0.0, 0
shown = 0
for kk in sorted(set(key.tolist())):
    ws = r[key == kk]
    ev = []
    for w in ws:
        for b in range(7):
            a, m = w[3 + 3 * b], w[4 + 3 * b]
            if a == 0:
                break
            ev.append((a, 1)); ev.append((m, -1))
    ev.sort()
    lo, hi = ws[:, 1].min(), ws[:, 31].max()
    cur, last, t1, t2 = 0, lo, 0.0, 0.0
    for t, d in ev:
        if cur == 1: t1 += t - last
        elif cur >= 2: t2 += t - last
        cur += d; last = t
    tot = (span_end - t0)
    idle_tot += (tot - t1 - t2); both_tot += t2; one_tot += t1; ncu += 1
    if shown < 6:
        shown += 1
        print(f"CU xcc{kk // 1000} se{(kk // 100) % 10} sh{(kk // 20) % 5} cu{kk % 20}: {len(ws)} workgroups " +
              " | ".join("wg%d: in %.1f pro %.1f " % (i, us(w[1]), us(w[2])) + " ".join("[%.1f %.1f %.1f]" % (us(w[3 + 3 * b]), us(w[4 + 3 * b]), us(w[5 + 3 * b])) for b in range(7) if w[3 + 3 * b]) + " out %.1f" % us(w[31])
                         for i, w in enumerate(ws)))
tot = (span_end - t0) / 100.0
print(f"per CU, of the {tot:.1f} us launch: one workgroup in its MFMA phase {one_tot / ncu / 100:.1f} us, two {both_tot / ncu / 100:.1f} us, none {idle_tot / ncu / 100:.1f} us")
wgs_per_cu = np.bincount([int((key == kk).sum()) for kk in set(key.tolist())])
print("workgroups per CU histogram (index = count):", wgs_per_cu.tolist())
