"""Adversarial operands for the split-bf16 conv12 kernels (csrc/conv_split.h) and the float64 reference they and the exact-fp32 MFMA
kernels (csrc/conv_tc.h) are both measured against (used by tests/test_gpu_parity.py and
scripts/dev/split_error.py).  Test infrastructure.

The block is conv1 (1 -> 32, 3x3 s2 p1) + ReLU + conv2 (32 -> 48, 3x3 s2 p1) + ReLU + 2x2 max-pool (conv_embedding_model.py:18-31).
conv1 is the same fp32 code in both kernel families, so the cases take its rounding out of the picture: every channel looks at ONE tap
with a 12-bit weight (times a power of two) and the pixels carry 12 bits, so a1 = pixel x weight is EXACT in fp32 - and has the full
24 significant bits the split has to carry.  What varies is what conv2's three reductions see:

  trained_scale      unit channel scales, conv2 weights ~ N(0, 0.06), d p2 ~ N(0, 1)
  range_compensated  a1's channels span 2^-40 .. 2^40 with conv2's weights scaled back (every product O(1), the operands 2^80 apart
                     inside one K reduction); d p2's channels 2^-20 .. 2^20 against the inverse in the weights' rows; pixels and
                     d p2 carry opposite block-wise exponents (0 .. 20) along the positions the weight gradient sums over
  range_raw          the same operand ranges, nothing scaled back: sums dominated by a few huge terms
  cancel_exact       24 of the 32 input channels in pairs with identical a1 and opposite conv2 weights, 2^20 times the rest (the
                     big terms cancel exactly, the result lives in the small ones); 40 of the 48 output channels in pairs with
                     identical rows and opposite d p2, likewise
  cancel_near        the same with the second member of each pair off by 2^-10
  same_sign          all conv2 weights and d p2 positive: every product of a sum has the same sign (the regime where pieces cut by
                     truncation would bias the result)
  end_to_end         dense seeded conv1 (its fp32 rounding is then part of both kernels' errors), trained scales

References are float64 evaluations of the fp32 inputs under the routing (conv1 ReLU bits, pool arg-max, conv2 ReLU) of the kernel
under test - the forward of each family under its own, both backward families under the fp32 forward's."""
import torch
import torch.nn.functional as F

CASES = ("trained_scale", "range_compensated", "range_raw", "cancel_exact", "cancel_near", "same_sign", "end_to_end")


def _pow2(t):
    return torch.pow(torch.tensor(2.0, dtype=torch.float64), t.double()).float()        # exact powers of two


def make(name, n, seed=0):
    g = torch.Generator().manual_seed(1000 * CASES.index(name) + seed)
    ri = lambda lo, hi, *s: torch.randint(lo, hi, s, generator=g)        # noqa: E731
    rn = lambda *s: torch.randn(*s, generator=g)                          # noqa: E731
    x = ri(1, 4096, n, 1, 128, 128).float() / 4096                        # 12-bit pixels in (0, 1)
    w2 = rn(48, 32, 3, 3) * 0.06
    b2 = rn(48) * 0.1
    dp2 = rn(n, 48, 16, 16)
    s_ci = torch.ones(32)                                                 # a1's channel scales (through w1)
    if name == "end_to_end":
        x = torch.rand(n, 1, 128, 128, generator=g)
        return x, rn(32, 1, 3, 3) * 0.3, rn(32) * 0.1, w2, b2, dp2
    if name in ("range_compensated", "range_raw"):
        s_ci = _pow2(ri(-40, 41, 32))
        u_co = _pow2(ri(-20, 21, 48))
        dp2 = dp2 * u_co.view(1, 48, 1, 1)
        if name == "range_compensated":
            w2 = w2 / s_ci.view(1, 32, 1, 1) / u_co.view(48, 1, 1, 1)
            b2 = b2 / u_co
            eb = ri(0, 21, n, 1, 8, 8)                                    # one exponent per 16 x 16 pixel block = 2 x 2 pooled cells
            x = x * _pow2(-eb).repeat_interleave(16, 2).repeat_interleave(16, 3)
            dp2 = dp2 * _pow2(eb).repeat_interleave(2, 2).repeat_interleave(2, 3)
    tap = torch.arange(32) % 9
    m12 = ri(2048, 4096, 32).float() / 4096                               # 12-bit weights in [0.5, 1)
    if name in ("cancel_exact", "cancel_near"):
        off = 1.0 + (2.0 ** -10 if name == "cancel_near" else 0.0)
        big = 2.0 ** 20
        # input channels 0..23 in pairs: same tap, same conv1 weight, opposite conv2 columns, 2^20 times the free channels' a1
        tap[1:24:2] = tap[0:24:2]
        m12[1:24:2] = m12[0:24:2]
        s_ci[:24] = big
        w2[:, 1:24:2] = -w2[:, 0:24:2] * off
        # output channels 0..39 in pairs: identical rows (identical y2, routing and ReLU), opposite d p2, 2^20 times the free ones
        w2[1:40:2] = w2[0:40:2]
        b2[1:40:2] = b2[0:40:2]
        dp2[:, 1:40:2] = -dp2[:, 0:40:2] * off
        dp2[:, :40] *= big
    if name == "same_sign":
        w2, dp2 = w2.abs(), dp2.abs()
    w1 = torch.zeros(32, 1, 3, 3)
    w1[torch.arange(32), 0, tap // 3, tap % 3] = m12 * s_ci
    return x, w1, torch.zeros(32), w2.contiguous(), b2.contiguous(), dp2.contiguous()


def ref64(x, w1, b1, w2, b2, dp2, routes):
    """float64, routing pinned: -> p2, (dw1, db1, dw2, db2) of sum(p2 * dp2)."""
    m1, am2, m2 = routes[0].double(), routes[1].long(), routes[2].double()
    p = [t.double().requires_grad_() for t in (w1, b1, w2, b2)]
    a1 = F.conv2d(x.double(), p[0], p[1], stride=2, padding=1) * m1
    y2 = F.conv2d(a1, p[2], p[3], stride=2, padding=1)
    n = x.shape[0]
    win = y2.view(n, 48, 16, 2, 16, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, 48, 16, 16, 4)
    p2 = torch.gather(win, 4, am2.unsqueeze(-1)).squeeze(-1) * m2
    grads = torch.autograd.grad((p2 * dp2.double()).sum(), p)
    return p2.detach(), grads


def _err(got, ref):
    d = (got.double().cpu() - ref).abs()
    s = ref.abs().max().clamp_min(1e-300)
    return float(d.max() / s), float(d.pow(2).mean().sqrt() / s)


def measure(lib, name, n, split_bits=7):
    """-> {quantity: ((max, rms) of the fp32 kernels, (max, rms) of the split kernels)}, errors relative to the reference's largest
    element.  Quantities: p2 (forward), dw2 / db2 (weight-gradient kernel), dw1 / db1 (data-gradient kernel)."""
    cpu = make(name, n)
    dev = [t.cuda() for t in cpu]
    x, w1, b1, w2, b2, dp2 = dev
    out = {}
    saved32 = routes32 = None
    fwd = {}
    try:
        for tag, bits in (("fp32", 0), ("split", split_bits)):
            lib.set_option("conv2_split", bits)
            p2, _, saved = lib.conv12_fwd(x, w1, b1, w2, b2)
            torch.cuda.synchronize()
            routes = lib.enc_routes(saved, n)
            ref_p2, grads = ref64(*cpu, routes)
            fwd[tag] = _err(p2, ref_p2)
            if tag == "fp32":
                saved32, ref_grads = saved, grads
        out["p2"] = (fwd["fp32"], fwd["split"])
        bwd = {}
        for tag, bits in (("fp32", 0), ("split", split_bits)):
            lib.set_option("conv2_split", bits)
            got = lib.conv12_bwd(x, w1, b1, w2, dp2, saved32)
            torch.cuda.synchronize()
            bwd[tag] = [_err(a, r) for a, r in zip(got, ref_grads)]
        for i, q in enumerate(("dw1", "db1", "dw2", "db2")):
            out[q] = (bwd["fp32"][i], bwd["split"][i])
    finally:
        lib.set_option("conv2_split", 0)
    return out
