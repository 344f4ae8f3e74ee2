"""Index-arithmetic checks without a GPU.  tests/hostsim/libmlhot_hostsim.so is the library's own
sources compiled for the host with every kernel launch replaced by a plain loop over the SAME
problem functors (im2col gathers, stride-2 parity classes, pooled-gradient routing, blocked head
weights, FAVOR+ backward...).  It is test infrastructure: the product never loads it.  Compared
against the CPU oracle / the reference's golden vectors."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_cpu as O
from tests import util as U


def test_linear(hostsim):
    g = torch.Generator().manual_seed(0)
    x, w, b, dy = torch.randn(37, 80, generator=g), torch.randn(100, 80, generator=g) * 0.1, torch.randn(100, generator=g), torch.randn(37, 100, generator=g)
    for act, fn in (("none", lambda t: t), ("relu", torch.relu), ("tanh", torch.tanh)):
        xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        yr = fn(F.linear(xr, wr, br))
        yr.backward(dy)
        y = hostsim.linear_fwd(x, w, b, act)
        dx, dw, db = hostsim.linear_bwd(x, w, y, dy, act)
        for got, want in ((y, yr), (dx, xr.grad), (dw, wr.grad), (db, br.grad)):
            assert U.rel_err(got, want) <= 1e-5


def test_encoder_two_segments(hostsim):
    g = torch.Generator().manual_seed(1)
    shapes = [("0.weight", (32, 1, 3, 3), 0.3), ("0.bias", (32,), 0.1), ("2.weight", (48, 32, 3, 3), 0.06), ("2.bias", (48,), 0.1),
              ("5.weight", (64, 48, 3, 3), 0.05), ("5.bias", (64,), 0.1), ("8.weight", (64, 4096), 0.02), ("8.bias", (64,), 0.1)]
    p = {"encoder_w0." + k: torch.randn(*s, generator=g) * a for k, s, a in shapes}
    x0, x1, df = torch.rand(2, 1, 128, 128, generator=g), torch.rand(1, 1, 128, 128, generator=g), torch.randn(3, 64, generator=g)
    pr = {k: v.clone().requires_grad_() for k, v in p.items()}
    fr = O.vanilla_encoder(torch.cat([x0, x1]), pr)
    fr.backward(df)
    f0, f1, saved = hostsim.enc_vanilla_fwd(x0, x1, list(p.values()), 64)
    assert U.rel_err(torch.cat([f0, f1]), fr) <= 1e-5
    grads = hostsim.enc_vanilla_bwd(x0, x1, list(p.values()), 64, df[:2].contiguous(), df[2:].contiguous(), saved)
    for (k, ref), got in zip(pr.items(), grads):
        assert U.rel_err(got, ref.grad) <= 1e-5, k


@pytest.mark.parametrize("mode", ["mean", "max", "baco"])
def test_aggregators(hostsim, mode):
    g = torch.Generator().manual_seed(2)
    rs, lv, dr = torch.randn(3, 7, 100, generator=g), torch.randn(3, 7, 100, generator=g) * 2, torch.randn(3, 100, generator=g)
    rr, ll = rs.clone().requires_grad_(), lv.clone().requires_grad_()
    ro = O.agg_mean(rr) if mode == "mean" else O.agg_max(rr) if mode == "max" else O.agg_baco(rr, 1e-5 + F.softplus(ll))[0]
    ro.backward(dr)
    r, sigma, amax = hostsim.agg_fwd(mode, rs, lv if mode == "baco" else None)
    drs, dlv = hostsim.agg_bwd(mode, rs, lv if mode == "baco" else None, r, sigma, amax, dr)
    assert U.rel_err(r, ro) <= 1e-5 and U.rel_err(drs, rr.grad) <= 1e-5
    if mode == "baco":
        assert U.rel_err(dlv, ll.grad) <= 1e-5


def test_favor_vs_reference_vectors(hostsim):
    fx = np.load(os.path.join(U.GOLDEN, "favor.npz"))
    for tag in json.loads(str(fx["meta"])):
        proj = torch.from_numpy(fx[f"{tag}/proj"])
        q, k, v, wout = (torch.from_numpy(fx[f"{tag}/{n}"]) for n in ("q", "k", "v", "wout"))
        T, H, Nq, d = q.shape
        qn, kn, vn = (t.permute(0, 2, 1, 3).contiguous() for t in (q, k, v))
        out, ws = hostsim.favor_fwd(qn, kn, vn, proj)
        assert U.rel_err(out.view(T, Nq, d, H).permute(0, 3, 1, 2), fx[f"{tag}/out"]) <= 1e-5
        dq, dk, dv = hostsim.favor_bwd(qn, kn, vn, proj, out, wout.permute(0, 2, 3, 1).reshape(T, Nq, d * H).contiguous(), ws)
        for n, gt in (("dq", dq), ("dk", dk), ("dv", dv)):
            assert U.rel_err(gt.permute(0, 2, 1, 3), fx[f"{tag}/{n}"], floor=1e-12) <= 1e-4, (tag, n)


@pytest.mark.parametrize("name", ["s_anp_shapenet1d_ragged", "s_anp_shapenet1d_nc0", "s_cnp_shapenet1d_baco",
                                  "s_cnp_pascal1d_max", "s_cnp_shapenet1d_nc0"])
def test_whole_model_orchestration_vs_reference(hostsim, name):
    fx, meta = U.load_case(name)
    model = U.build_model(meta, fx=fx)
    c = meta["cfg"]
    cx, qx, cy, qy = U.case_inputs(meta)
    params = {k: p.detach() for k, p in model.named_parameters()}
    proj = model.attn.projection_matrix if model.ATTENTION else None
    dims = hostsim.np_dims(c["tasks_per_batch"], meta["Nc"], meta["Nq"], c["input_dim"], c["output_dim"], c["dim_w"], c["dim_r"],
                           c["dim_z"], c["n_hidden_units_r"], 100, c["agg_mode"], model.OUT_TANH, proj.shape[0] if proj is not None else 0)
    mu, saved, scratch = hostsim.np_vanilla_fwd(dims, params, cx, cy, qx, proj)
    kind = U.loss_kind(c["task"])
    loss = hostsim.loss_fwd(kind, mu, qy)
    grads = hostsim.np_vanilla_bwd(dims, params, cx, cy, qx, mu, hostsim.loss_bwd(kind, mu, qy, torch.tensor(1.0)), saved, scratch, proj)
    assert U.rel_err(mu, fx["mu"]) <= 1e-5
    assert abs(loss.item() - float(fx["loss"])) <= 1e-5
    from mlhot.ops import used_param_keys
    used = used_param_keys(tuple(params), meta["Nc"])
    U.check_grads_against_fixture({k: (g if k in used else None) for k, g in grads.items()}, fx, meta)


def test_adam_step_matches_torch_adam(hostsim):
    """mlhot_adam_step (one launch over a flat buffer) vs torch.optim.Adam over the same values, incl. weight decay and
    the 1/world gradient scale, for several steps."""
    g = torch.Generator().manual_seed(5)
    n = 1037
    p0 = torch.randn(n, generator=g)
    for wd, scale in ((0.0, 1.0), (0.01, 0.5)):
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
        for t in range(1, 6):
            grad = torch.randn(n, generator=g)
            ref.grad = grad * scale
            opt.step()
            hostsim.adam_step(p, grad, m, v, 1e-2, 0.9, 0.999, 1e-8, wd, scale, t)
            assert U.rel_err(p, ref.detach()) <= 2e-6, (wd, t)


def test_ingest_matches_reference_vectors(hostsim):
    """mlhot_ingest_u8_nhwc argument handling + the element formula (host flavour) against the reference-produced vectors."""
    import os
    import numpy as np
    from mlhot.binding import MlhotError
    fx = np.load(os.path.join(U.GOLDEN, "ingest.npz"))
    for C in (1, 2, 3, 4):
        got = hostsim.ingest_u8_nhwc(torch.from_numpy(fx[f"c{C}/u8"]))
        assert np.array_equal(got.numpy(), fx[f"c{C}/f32"]), C
    assert hostsim.ingest_u8_nhwc(torch.zeros(2, 0, 4, 4, 1, dtype=torch.uint8)).shape == (2, 0, 1, 4, 4)
    with pytest.raises(MlhotError):
        hostsim.ingest_u8_nhwc(torch.zeros(2, 4, 4, 1))                 # not uint8
    with pytest.raises(MlhotError):
        hostsim.ingest_u8_nhwc(torch.zeros(1, 4, 4, 1, dtype=torch.uint8), out=torch.zeros(1, 4, 4, 1))   # out not channel-first


@pytest.mark.parametrize("shape", [(2, 3, 12, 12, 8, 5, 2, 2), (2, 4, 9, 9, 6, 3, 2, 1), (1, 5, 8, 8, 7, 3, 1, 1), (2, 4, 8, 8, 5, 1, 2, 0),
                                   (1, 2, 7, 7, 3, 3, 3, 1)])
@pytest.mark.parametrize("relu", [False, True])
def test_conv2d_runtime_shapes_vs_torch(hostsim, shape, relu):
    """Run-time-shaped convolution problems (ConvFwdRT / ConvWgradRT / ConvDgradRT incl. its s*s input-parity classes, batched four
    per launch on the device): index arithmetic of the functors against F.conv2d and its autograd, odd sizes, strides 1-3, k = 1/3/5."""
    N, Cin, H, W, Cout, k, s, p = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, generator=g).requires_grad_(True)
    w = torch.randn(Cout, Cin, k, k, generator=g).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    ref = F.relu(ref) if relu else ref
    y = hostsim.conv2d_fwd(x.detach(), w.detach(), b.detach(), s, p, relu)
    assert U.rel_err(y, ref) <= 1e-5
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    dx, dw, db = hostsim.conv2d_bwd(x.detach(), w.detach(), y, dy, s, p, relu)
    assert U.rel_err(dx, x.grad) <= 1e-5 and U.rel_err(dw, w.grad) <= 1e-5 and U.rel_err(db, b.grad) <= 1e-5


def test_adam_step_counter_matches_adam_step(hostsim):
    """The device-side step count variant (capture-safe) walks the same trajectory as the host-counted one."""
    g = torch.Generator().manual_seed(9)
    n = 517
    p0 = torch.randn(n, generator=g)
    pa, ma, va = p0.clone(), torch.zeros(n), torch.zeros(n)
    pb, mb, vb = p0.clone(), torch.zeros(n), torch.zeros(n)
    counter = torch.zeros(1, dtype=torch.int32)
    for t in range(1, 8):
        grad = torch.randn(n, generator=g)
        hostsim.adam_step(pa, grad, ma, va, 3e-3, 0.9, 0.999, 1e-8, 0.01, 0.5, t)
        hostsim.adam_step_counter(pb, grad, mb, vb, 3e-3, 0.9, 0.999, 1e-8, 0.01, 0.5, counter)
        assert int(counter) == t
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), t
