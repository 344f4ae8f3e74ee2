#!/usr/bin/env python3
"""Probe (extras only, never the headline): how exact is conv2 on the bf16 matrix pipe when every fp32 operand is
split into bf16 pieces?  CPU study, no GPU needed.

x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round to nearest even): 3 x 8 = 24 mantissa bits,
fp32's own.  A product x * w is then a sum of piece products; each piece product is EXACT in fp32 (8 x 8 bits), and the matrix core
accumulates in fp32.  Variants:
    6 products : hh, hm, mh, hl, lh, mm          (drops ml, lm ~ 2^-24 each, ll ~ 2^-32)
    8 products : + ml, lm
    9 products : all
against (a) float64 and (b) the fp32 MFMA as the shipped kernels run it (bit-for-bit a k-ordered fp32 fmaf chain, cdna guide
"FP32-input MFMA").  Data: conv2 of the vanilla encoder on real activations (a1 = ReLU(conv1(image)) of seeded images, the
seeded initial weights), K = 288 per output, a few thousand outputs.  Accumulation model for the bf16 path: exact piece products
added one by one in fp32 in k order, the pieces of one k back to back - the hardware adds 32 k's per instruction with at least
this precision, so this is the pessimistic end.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]


def bf16_round(x):
    """float32 -> nearest-even bfloat16, returned as float32"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    h = bf16_round(x)
    m = bf16_round((x - h).astype(np.float32))
    lo = bf16_round((x - h - m).astype(np.float32))
    return h, m, lo


def fma_chain(a, b):
    """sum_k a[.., k] * b[.., k] as a k-ordered fp32 fmaf chain (numpy has no fma: emulate with float64 product + one rounding per step,
    exact because a float32 x float32 product fits a float64)"""
    acc = np.zeros(a.shape[:-1], dtype=np.float32)
    for k in range(a.shape[-1]):
        acc = (acc.astype(np.float64) + a[..., k].astype(np.float64) * b[..., k].astype(np.float64)).astype(np.float32)
    return acc


def main():
    from networks.ANPShapeNet1D import ANPShapeNet1D
    from mlhot.synth import get_batch
    cfg = types.SimpleNamespace(device=torch.device("cpu"), seed=2578, img_size=[128, 128, 1], tasks_per_batch=16, input_dim=3,
                                output_dim=2, agg_mode="attention", img_agg="", dim_w=64, n_hidden_units_r=[100, 100], dim_r=64,
                                dim_z=64, task="shapenet_1d")
    model = ANPShapeNet1D(cfg)
    sd = model.state_dict()
    cx, _, _, _ = get_batch("shapenet_1d", 1, 4, 1, seed=1234)
    with torch.no_grad():
        a1 = torch.relu(torch.nn.functional.conv2d(cx.reshape(-1, 1, 128, 128), sd["encoder_w0.0.weight"], sd["encoder_w0.0.bias"], stride=2, padding=1))
        cols = torch.nn.functional.unfold(a1, 3, padding=1, stride=2)            # [n, 288, 1024]
    rng = np.random.RandomState(0)
    pos = rng.choice(cols.shape[2], 96, replace=False)
    A = cols[:, :, pos].permute(0, 2, 1).reshape(-1, 288).numpy().astype(np.float32)        # [384, 288] im2col rows
    W = sd["encoder_w0.2.weight"].reshape(48, 288).numpy().astype(np.float32)
    a = np.repeat(A[:, None, :], 48, axis=1)                 # [384, 48, 288]
    w = np.repeat(W[None, :, :], A.shape[0], axis=0)
    ref = (a.astype(np.float64) * w.astype(np.float64)).sum(-1)
    scale = np.abs(ref).max()
    out = {"fp32 MFMA (k-ordered fmaf chain)": fma_chain(a, w)}
    ah, am, al = split3(a)
    wh, wm, wl = split3(w)
    combos = {"bf16 x 6": [(ah, wh), (ah, wm), (am, wh), (ah, wl), (al, wh), (am, wm)],
              "bf16 x 8": [(ah, wh), (ah, wm), (am, wh), (ah, wl), (al, wh), (am, wm), (am, wl), (al, wm)],
              "bf16 x 9": [(ah, wh), (ah, wm), (am, wh), (ah, wl), (al, wh), (am, wm), (am, wl), (al, wm), (al, wl)],
              "bf16 x 3 (hh, hm, mh)": [(ah, wh), (ah, wm), (am, wh)]}
    for name, prods in combos.items():
        # small terms first inside one k, then the running sum in k order (fp32)
        acc = np.zeros(a.shape[:-1], dtype=np.float32)
        for k in range(288):
            t = np.zeros_like(acc, dtype=np.float64)
            for pa, pw in reversed(prods):
                t = t + pa[..., k].astype(np.float64) * pw[..., k].astype(np.float64)      # exact piece products, their sum held wide
            acc = (acc.astype(np.float64) + t.astype(np.float32).astype(np.float64)).astype(np.float32)
        out[name] = acc
        # the other end: every piece product added to the fp32 accumulator on its own
        acc2 = np.zeros(a.shape[:-1], dtype=np.float32)
        for pa, pw in prods:
            for k in range(288):
                acc2 = (acc2.astype(np.float64) + pa[..., k].astype(np.float64) * pw[..., k].astype(np.float64)).astype(np.float32)
        out[name + " [piece by piece]"] = acc2
    print(f"conv2 outputs: {ref.size}, K = 288, largest |output| {scale:.4f}")
    print(f"{'variant':42s} {'max |err| / max |out|':>22s} {'rms err / max |out|':>22s}")
    for name, v in out.items():
        e = np.abs(v.astype(np.float64) - ref)
        print(f"{name:42s} {e.max() / scale:22.3e} {np.sqrt((e ** 2).mean()) / scale:22.3e}")


if __name__ == "__main__":
    main()
