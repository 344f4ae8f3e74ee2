#!/usr/bin/env python3
"""Throughput of the generic igemm engine on plain GEMM shapes (the im2col view of the ResNet convolutions)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch
import mlhot
from mlhot import ops
dev = torch.device("cuda", 0)
L = mlhot.lib()
for (M, K, N) in []:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    for _ in range(3): y = L.linear_fwd(x, w, b, "relu") if hasattr(L, "linear_fwd") else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): y = L.linear_fwd(x, w, b, "relu")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"linear_fwd M={M} K={K} N={N}: {dt*1e6:.1f} us  {2*M*K*N/dt/1e12:.1f} TF/s")
    # the conv it stands for (where it is one)
tot = [0.0, 0.0]
for (n, ci, h, co, k, s, p) in [(120, 3, 64, 64, 5, 2, 2), (120, 64, 32, 64, 3, 2, 1), (120, 64, 16, 64, 3, 1, 1), (120, 64, 16, 64, 3, 2, 1),
                                (120, 64, 8, 64, 3, 1, 1), (120, 64, 8, 64, 3, 2, 1), (120, 64, 4, 64, 3, 1, 1), (120, 64, 4, 64, 3, 2, 1),
                                (120, 64, 2, 64, 3, 1, 1), (120, 64, 32, 64, 1, 2, 0)]:
    x = torch.randn(n, ci, h, h, device=dev); w = torch.randn(co, ci, k, k, device=dev); b = torch.randn(co, device=dev)
    for _ in range(3): y = L.conv2d_fwd(x, w, b, s, p, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): y = L.conv2d_fwd(x, w, b, s, p, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    ho = y.shape[-1]
    print(f"conv2d_fwd n={n} ci={ci} h={h} co={co} k={k} s={s}: {dt*1e6:.1f} us  {2*n*ho*ho*co*ci*k*k/dt/1e12:.1f} TF/s")
    dy = torch.randn_like(y)
    need_dx = ci > 3
    for _ in range(3): g = L.conv2d_bwd(x, w, y, dy, s, p, True, need_dx=need_dx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g = L.conv2d_bwd(x, w, y, dy, s, p, True, need_dx=need_dx)
    torch.cuda.synchronize()
    db = (time.perf_counter() - t0) / 20
    print(f"   conv2d_bwd: {db*1e6:.1f} us")
    tot[0] += dt; tot[1] += db
print("sum fwd %.1f us, bwd %.1f us" % (tot[0] * 1e6, tot[1] * 1e6))
