#!/usr/bin/env python3
"""How should a reference-style loader's fp32 host batch (31.5 MB for c3) reach the device?  Times, per batch:
  a) tensor.to(device) from pageable memory (the reference's route, model_trainer.py:67-70)
  b) np.copyto into a pinned buffer (host memcpy, one thread) - the host-side cost of a staged copy
  c) pinned -> device, async on a copy stream (the DMA itself)
  d) torch copy_ into the pinned buffer (multi-threaded)"""
import time
import numpy as np
import torch

dev = torch.device("cuda", 0)
shapes = [(16, 15, 1, 128, 128), (16, 15, 1, 128, 128), (16, 15, 3), (16, 15, 3)]
host = [torch.rand(s) for s in shapes]
pinned = [torch.empty(s).pin_memory() for s in shapes]
devt = [torch.empty(s, device=dev) for s in shapes]
torch.cuda.synchronize()


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def a():
    for d, h in zip(devt, host):
        d.copy_(h)


def b():
    for p, h in zip(pinned, host):
        np.copyto(p.numpy(), h.numpy())


cs = torch.cuda.Stream()


def c():
    with torch.cuda.stream(cs):
        for d, p in zip(devt, pinned):
            d.copy_(p, non_blocking=True)


def d():
    for p, h in zip(pinned, host):
        p.copy_(h)


mb = sum(t.numel() * 4 for t in host) / 1e6
for name, fn in (("pageable .copy_ to device", a), ("np.copyto into pinned (1 thread)", b), ("pinned -> device async", c), ("torch copy_ into pinned", d)):
    ms = timeit(fn)
    print(f"{name:36s} {ms:7.3f} ms / {mb:.1f} MB = {mb / ms:6.1f} GB/s")
# host memcpy while the GPU DMA of the previous buffer runs (what a prefetching trainer does)
def bc():
    c(); b()
print(f"{'DMA issued, then memcpy (overlap)':36s} {timeit(bc):7.3f} ms")
