"""Shared helpers for the parity tests."""
import glob
import hashlib
import importlib
import json
import os
import types

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# north_star: "outputs match the reference PyTorch CPU path within 1e-4 rel fp32".
# Metric (SURVEY.md §7 hard parts): tensor-scale relative error  max|a-b| / max|b|.
RTOL = 1e-4
# Gradients far below the model's largest gradient entry are fp32 cancellation residue (at
# init the FAVOR+ normaliser cancels the Q/K scale, leaving ~1e-9 gradients next to ~1e-1
# ones); they are compared at a floor of GRAD_FLOOR x (largest |grad| entry of the model).
GRAD_FLOOR = 1e-4


def rel_err(a, b, floor=0.0):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    assert a.numel() == b.numel(), (a.shape, b.shape)
    a = a.reshape(b.shape)
    scale = max(b.abs().max().item(), floor, 1e-30)
    return (a - b).abs().max().item() / scale


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def model_case_names(prefix=""):
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, prefix + "*.npz"))
                  if os.path.basename(f)[0] in "cs" and os.path.basename(f)[1] in "123_" and "conv_emb" not in f)


def load_case(name):
    fx = np.load(os.path.join(GOLDEN, name + ".npz"))
    return fx, json.loads(str(fx["meta"]))


def case_config(meta, device="cpu"):
    return types.SimpleNamespace(device=torch.device(device), **meta["cfg"])


def build_model(meta, device="cpu", fx=None):
    """Identically-seeded module of this repo.  Parameters must regenerate bit-exactly (sha-checked
    against the reference); the FAVOR+ projection buffer comes out of a LAPACK QR whose low bits
    depend on the host CPU, so it is checked to 1e-5 and then loaded from the fixture."""
    cfg = case_config(meta, device)
    mod = importlib.import_module("networks." + meta["method"])
    model = getattr(mod, meta["method"])(cfg)
    for k, v in model.state_dict().items():
        if k == "attn.projection_matrix":
            continue
        assert sha(v) == meta["state_sha"][k], f"{k}: seeded init differs from the reference"
    if fx is not None and "projection_matrix" in fx.files:
        ref = torch.from_numpy(fx["projection_matrix"])
        assert sha(ref) == meta["state_sha"]["attn.projection_matrix"]
        assert rel_err(model.attn.projection_matrix, ref) <= 1e-3   # QR low bits are host-LAPACK dependent
        with torch.no_grad():
            model.attn.projection_matrix.copy_(ref)
    return model


def case_inputs(meta):
    c = meta["cfg"]
    T, Nc, Nq = c["tasks_per_batch"], meta["Nc"], meta["Nq"]
    H, W, C = c["img_size"]
    g = torch.Generator().manual_seed(meta["input_seed"])
    cx = torch.rand(T, Nc, C, H, W, generator=g)
    qx = torch.rand(T, Nq, C, H, W, generator=g)
    cy = torch.rand(T, Nc, c["input_dim"], generator=g)
    qy = torch.rand(T, Nq, c["input_dim"], generator=g)
    for k, t in (("cx", cx), ("qx", qx), ("cy", cy), ("qy", qy)):
        assert sha(t) == meta["input_sha"][k], f"input {k} does not regenerate from the seed"
    return cx, qx, cy, qy


def resnet_case_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "r_*.npz")))


def fcl_case_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "f_*.npz")))


def mr_case_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "m_*.npz")))


def resnet_case_inputs(meta, fx):
    c = meta["cfg"]
    T, Nc, Nq, C = c["tasks_per_batch"], meta["Nc"], meta["Nq"], meta["C"]
    H, W, _ = c["img_size"]
    g = torch.Generator().manual_seed(meta["input_seed"])
    cx = torch.rand(T, Nc, C, H, W, generator=g)
    qx = torch.rand(T, Nq, C, H, W, generator=g)
    cy = torch.rand(T, Nc, c["input_dim"], generator=g)
    for k, t in (("cx", cx), ("qx", qx), ("cy", cy)):
        assert sha(t) == meta["input_sha"][k], f"input {k} does not regenerate from the seed"
    return cx, qx, cy, torch.from_numpy(fx["qy"])


def loss_kind(task):
    return {"shapenet_1d": "azimuth", "pascal_1d": "mse", "shapenet_3d": "quaternion", "distractor": "distractor"}[task]


def check_grads_against_fixture(grads, fx, meta, tol=RTOL, head=4096, stride_cap=None):
    """grads: dict key -> tensor (or None).  Compares with the fixture's full / sampled grads."""
    gmax = 0.0
    for k in grads:
        if "grad/" + k in fx:
            gmax = max(gmax, float(np.abs(fx["grad/" + k]).max()))
        elif "gradhead/" + k in fx:
            gmax = max(gmax, float(np.abs(fx["gradhead/" + k]).max()))
    floor = GRAD_FLOOR * gmax
    worst = (0.0, None)
    for k, g in grads.items():
        want_norm = meta["grad_norm"][k]
        if want_norm is None:
            assert g is None, f"{k}: reference has grad=None"
            continue
        assert g is not None, f"{k}: missing gradient"
        g = g.detach().cpu()
        if "grad/" + k in fx:
            e = rel_err(g, fx["grad/" + k], floor)
        else:
            flat = g.reshape(-1)
            strided = flat[1::61] if stride_cap is None else flat[1::61][:stride_cap]
            e = max(rel_err(flat[:head], fx["gradhead/" + k], floor), rel_err(strided, fx["gradstride/" + k], floor))
            assert abs(float(flat.double().norm()) - want_norm) <= tol * max(want_norm, floor), f"{k}: grad norm"
        if e > worst[0]:
            worst = (e, k)
        assert e <= tol, f"{k}: gradient rel err {e:.3e} > {tol}"
    return worst


# ---- pinned routing: the kernels' own ReLU / pool decisions, fed to the oracle; every disagreement with the
# oracle's own decision must sit on a rounding-level tie of the oracle's pre-activations -----------------------------------
TIE = 1e-5


def relu_flips(mask, pre, what="", tie=None):
    """# of ReLU decisions in `mask` that differ from sign(pre); asserts that each of them is a tie (|pre| <= tie * max|pre|,
    tie = TIE unless given)."""
    tie = TIE if tie is None else tie
    mask, pre = torch.as_tensor(mask).cpu().reshape(pre.shape), pre.detach()
    bad = (mask > 0) != (pre > 0)
    n = int(bad.sum())
    if n:
        worst = float((bad * pre.abs()).max() / pre.abs().max())
        assert worst <= tie, f"{what}: ReLU routing differs away from a tie ({worst:.2e} of the layer's largest pre-activation > {tie:.2e})"
    return n


def encoder_flips(route, pre, what="", tie=None):
    """Vanilla encoder: route = (m1, arg2, m2, m3) against the pre-activations of oracle.vanilla_encoder_routed."""
    m1, arg2, m2, m3 = route
    n = relu_flips(m1, pre["y1"], what + " conv1", tie) + relu_flips(m3, pre["y3"], what + " conv3", tie)
    win = torch.relu(pre["y2win"].detach())               # the pool runs on the post-ReLU map
    chosen = torch.gather(win, 4, arg2.long().unsqueeze(-1)).squeeze(-1)
    gap = win.max(dim=4).values - chosen
    assert not bool((gap > (TIE if tie is None else tie) * win.abs().max()).any()), f"{what}: pool arg-max differs away from a tie"
    n += int((gap > 0).sum())
    n += relu_flips(m2, torch.gather(pre["y2win"].detach(), 4, arg2.long().unsqueeze(-1)).squeeze(-1), what + " conv2", tie)
    return n


def self_routes_vanilla(x, p, prefix="encoder_w0."):
    """The oracle's OWN routing decisions on images x, in the form vanilla_encoder_routed takes them."""
    from oracle import ref_cpu as O
    taps = {}
    with torch.no_grad():
        O.vanilla_encoder(x, p, prefix=prefix, taps=taps)
    a2 = taps["a2"]
    n, c, h, w = a2.shape
    win = a2.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    return (taps["a1"] > 0).float(), win.argmax(dim=4).to(torch.uint8), (taps["p2"] > 0).float(), (taps["a3"] > 0).float()
