"""Host side of the byte route (mlhot.ingest.ExactU8Feed) on the GPU box: mlhot_host_f32_to_u8_exact on c3's 31.5 MB of fp32 images by
thread count, and ExactU8Feed.stage() (conversion + labels + async H2D) as the trainer calls it.
usage: python scripts/dev/u8_feed_probe.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "what-matters-for-meta-learning_amd"))
import mlhot
from mlhot import synth
from mlhot.ingest import ExactU8Feed

L = mlhot.lib()
hb = synth.get_batch_u8("shapenet_1d", 16, 15, 15, seed=1)
host = (synth.host_convert(hb[0]), synth.host_convert(hb[1]), hb[2], hb[3])
x = torch.cat([host[0].reshape(-1), host[1].reshape(-1)])
dst = np.empty(x.numel(), dtype=np.uint8)
for thr in (1, 2, 4, 8, 16, 32):
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        bad = L.host_f32_to_u8_exact(x.data_ptr(), dst.ctypes.data, x.numel(), threads=thr)
        ts.append(time.perf_counter() - t0)
    print(f"convert {x.numel() / 1e6:.1f} M floats on {thr:2d} threads: median {1e3 * sorted(ts)[6]:.3f} ms (min {1e3 * min(ts):.3f}), inexact {bad}")
if torch.cuda.is_available():
    for thr in (4, 8, 16):
        feed = ExactU8Feed("cuda:0", threads=thr)
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            tk = feed.stage(host)
            ts.append(time.perf_counter() - t0)
            feed.take(tk)
            torch.cuda.synchronize()
        print(f"ExactU8Feed.stage (convert + labels + async H2D issue) on {thr} threads: median {1e3 * sorted(ts)[6]:.3f} ms")
