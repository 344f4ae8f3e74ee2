"""CPU oracle for the task-batched CNP/ANP hot path.  TEST INFRASTRUCTURE ONLY.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``what-matters-for-meta-learning_amd/``) never falls back to it.

It is a straight-line fp32 restatement of the reference's arithmetic on torch-CPU
tensors, written from SURVEY.md Appendix A (A.1-A.7), one function per row of
SURVEY.md §8(a).  Every function cites the reference file:line it restates.  The
element arithmetic (convolution, GEMM, exp) is torch/ATen CPU, which is also what
the reference itself runs on (the reference has no kernels of its own).

Parity pin: the reference holds NO tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
imported in the build container by ``tests/golden/make_fixtures.py`` and committed
as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those vectors (forward taps, loss and gradients).

Parameters are passed as plain dicts keyed exactly like the reference's
``state_dict`` so a fixture's weights drop straight in.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# E1  vanilla encoder                      (ANPShapeNet1D.py:46-56, CNPShapeNet1D.py:46-56)
# --------------------------------------------------------------------------------------


def vanilla_encoder(img, p, prefix="encoder_w0.", taps=None):
    """conv(1->32,k3,s2,p1)+ReLU, conv(32->48)+ReLU, maxpool2, conv(48->64)+ReLU,
    flatten (C-major), Linear(4096->dim_w).  img: [N,C,128,128] -> [N,dim_w]."""
    a1 = F.relu(F.conv2d(img, p[prefix + "0.weight"], p[prefix + "0.bias"], stride=2, padding=1))
    a2 = F.relu(F.conv2d(a1, p[prefix + "2.weight"], p[prefix + "2.bias"], stride=2, padding=1))
    p2 = F.max_pool2d(a2, 2)
    a3 = F.relu(F.conv2d(p2, p[prefix + "5.weight"], p[prefix + "5.bias"], stride=2, padding=1))
    flat = a3.reshape(a3.shape[0], -1)
    out = F.linear(flat, p[prefix + "8.weight"], p[prefix + "8.bias"])
    if taps is not None:
        taps.update(a1=a1, a2=a2, p2=p2, a3=a3)
    return out


def vanilla_encoder_routed(img, p, m1, arg2, m2, m3, prefix="encoder_w0."):
    """Same arithmetic as vanilla_encoder, but every piecewise-linear routing decision (the three
    ReLU masks and the max-pool arg-max, window index 2*dy+dx) is GIVEN instead of being re-derived
    from this function's own pre-activations.  ReLU and max-pool gradients are discontinuous: an
    activation within fp32 rounding of zero (or a pool tie) passes all of its upstream gradient or
    none, so two fp32 evaluations with different summation orders disagree on a handful of the 63 M
    routing decisions of a 480-image batch and each disagreement moves a conv gradient by ~1e-4 of
    its scale (the CPU fp32-vs-fp64 comparison shows the same).  With the routing pinned to the
    kernel's own decisions the gradients must agree to rounding, which is what the full-size test
    checks; the decisions themselves are checked against this function's pre-activations."""
    y1 = F.conv2d(img, p[prefix + "0.weight"], p[prefix + "0.bias"], stride=2, padding=1)
    a1 = y1 * m1
    y2 = F.conv2d(a1, p[prefix + "2.weight"], p[prefix + "2.bias"], stride=2, padding=1)
    n, c, h, w = y2.shape
    win = y2.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    p2 = torch.gather(win, 4, arg2.long().unsqueeze(-1)).squeeze(-1) * m2
    y3 = F.conv2d(p2, p[prefix + "5.weight"], p[prefix + "5.bias"], stride=2, padding=1)
    a3 = y3 * m3
    out = F.linear(a3.reshape(n, -1), p[prefix + "8.weight"], p[prefix + "8.bias"])
    return out, dict(y1=y1, y2win=win, y3=y3)


# --------------------------------------------------------------------------------------
# M1  task-side MLPs                                               (models.py:27-60)
# --------------------------------------------------------------------------------------


def _act(v, masks, pre):
    """ReLU, or (pinned routing) multiplication by the next given {0,1} mask; `pre` receives the pre-activation."""
    if pre is not None:
        pre.append(v.detach())
    return F.relu(v) if masks is None else v * next(masks)


def encoder_fc(x, p, prefix="encoder_r.layers.", n_hidden=2, masks=None, pre=None):
    """EncoderFC: (Linear+ReLU) x n_hidden, then Linear without activation.  `masks` (optional): the n_hidden ReLU
    decisions, given instead of re-derived (see vanilla_encoder_routed); `pre`: a list receiving the pre-activations."""
    idx = 0
    masks = iter(masks) if masks is not None else None
    for _ in range(n_hidden):
        x = _act(F.linear(x, p[f"{prefix}{idx}.weight"], p[f"{prefix}{idx}.bias"]), masks, pre)
        idx += 2
    return F.linear(x, p[f"{prefix}{idx}.weight"], p[f"{prefix}{idx}.bias"])


def decoder_mlp(x, p, prefix="decoder0.", tanh=True, masks=None, pre=None):
    """decoder0: L(128->100) ReLU L(100->100) ReLU L(100->y) [tanh]
    (ANPShapeNet1D.py:65-72; no tanh in CNPVanillaPascal1D.py:67-73).  `masks` / `pre` as in encoder_fc."""
    masks = iter(masks) if masks is not None else None
    x = _act(F.linear(x, p[prefix + "0.weight"], p[prefix + "0.bias"]), masks, pre)
    x = _act(F.linear(x, p[prefix + "2.weight"], p[prefix + "2.bias"]), masks, pre)
    x = F.linear(x, p[prefix + "4.weight"], p[prefix + "4.bias"])
    return torch.tanh(x) if tanh else x


# --------------------------------------------------------------------------------------
# G1  aggregators over the shot axis              (CNPShapeNet1D.py:78-126, A.2)
# --------------------------------------------------------------------------------------


def agg_mean(rs):
    return rs.mean(dim=1)


def agg_max(rs, amax=None):
    """max over the shot axis; `amax` [T,R] (optional): the winning shot, given instead of re-derived."""
    if amax is None:
        return rs.max(dim=1)[0]
    return torch.gather(rs, 1, amax.long().unsqueeze(1)).squeeze(1)


def agg_baco(mu, var):
    """Bayesian context aggregation with prior N(0,1): sigma_z = 1/(1+sum 1/var),
    mu_z = sigma_z * sum(mu/var)          (CNPShapeNet1D.py:78-94)."""
    inv = 1.0 / var
    sigma_z = 1.0 / (1.0 + inv.sum(dim=1))
    mu_z = sigma_z * (inv * mu).sum(dim=1)
    return mu_z, sigma_z


# --------------------------------------------------------------------------------------
# A2/A3  FAVOR+                                              (fast_attention.py:74-156)
# --------------------------------------------------------------------------------------


def favor_features(x, proj, is_query, eps=1e-4):
    """softmax_kernel (fast_attention.py:74-99).  x: [T,H,N,d], proj: [m,d]."""
    d = x.shape[-1]
    m = proj.shape[0]
    c = d ** -0.25
    ratio = m ** -0.5
    dd = torch.einsum("thnd,md->thnm", c * x, proj)
    diag = (x * x).sum(dim=-1, keepdim=True) * 0.5 * (c * c)
    if is_query:
        stab = dd.max(dim=-1, keepdim=True).values
    else:
        stab = dd.max()  # ONE scalar over the whole [T,H,N,m] tensor (fast_attention.py:97)
    return ratio * (torch.exp(dd - diag - stab) + eps)


def linear_attention(qp, kp, v):
    """Non-causal linear attention (fast_attention.py:151-156)."""
    ksum = kp.sum(dim=-2)
    dinv = 1.0 / torch.einsum("thnm,thm->thn", qp, ksum)
    ctx = torch.einsum("thnm,thne->thme", kp, v)
    return torch.einsum("thme,thnm,thn->thne", ctx, qp, dinv)


def favor_attention(q, k, v, proj):
    """FastAttention.forward, default branch (fast_attention.py:187-205)."""
    return linear_attention(favor_features(q, proj, True), favor_features(k, proj, False), v)


def gaussian_orthogonal_random_matrix(nb_rows, nb_cols):
    """Projection-matrix draw, scaling=0 (fast_attention.py:117-146).  Consumes the
    global torch CPU generator exactly like the reference: per block randn(d,d) -> QR
    -> Q^T; then one randn(m,d) whose row norms scale the rows."""
    blocks = []
    full = nb_rows // nb_cols
    for _ in range(full):
        qm, _ = torch.linalg.qr(torch.randn(nb_cols, nb_cols))
        blocks.append(qm.t())
    rem = nb_rows - full * nb_cols
    if rem > 0:
        qm, _ = torch.linalg.qr(torch.randn(nb_cols, nb_cols))
        blocks.append(qm.t()[:rem])
    mat = torch.cat(blocks)
    mult = torch.randn(nb_rows, nb_cols).norm(dim=1)
    return torch.diag(mult) @ mat


# --------------------------------------------------------------------------------------
# A1/A4  multi-head wrapper                                 (ANPShapeNet1D.py:93-116)
# --------------------------------------------------------------------------------------


def multihead_attention(k, v, q, p, n_heads=8, taps=None):
    """8 per-head Linear(h,h) for K(x_ctx), V(rs), Q(x_qry); FAVOR+; merge with
    merged[t,n,d*8+i] = out[t,i,n,d]; _W Linear(8h->h)."""
    def heads(x, name):
        return torch.stack([F.linear(x, p[f"{name}.{i}.linear.weight"], p[f"{name}.{i}.linear.bias"])
                            for i in range(n_heads)], dim=1)
    kh, vh, qh = heads(k, "_W_k"), heads(v, "_W_v"), heads(q, "_W_q")
    out = favor_attention(qh, kh, vh, p["attn.projection_matrix"])
    merged = out.permute(0, 2, 3, 1).reshape(out.shape[0], out.shape[2], -1)
    if taps is not None:
        taps.update(kh=kh, vh=vh, qh=qh, attn_out=out)
    return F.linear(merged, p["_W.linear.weight"], p["_W.linear.bias"])


# --------------------------------------------------------------------------------------
# whole-model forwards (c1-c4)
# --------------------------------------------------------------------------------------


def vanilla_np_forward(p, ctx_x, ctx_y, qry_x, agg_mode, tanh, taps=None, routes=None, pres=None):
    """CNP/ANP vanilla model forward for Pascal1D / ShapeNet1D
    (CNPShapeNet1D.py:96-140, ANPShapeNet1D.py:118-157, CNPVanillaPascal1D.py:98-142,
    ANPVanillaPascal1D.py:136-176).  Returns mu [T,Nq,y].

    `routes` (optional, pinned routing - see vanilla_encoder_routed): a dict with the piecewise-linear decisions of a
    particular evaluation (the HIP kernels'): "enc_qry" / "enc_ctx" = (m1, arg2, m2, m3) of the two image sets, "h" = the
    EncoderFC ReLU masks, "d" = the two decoder0 ReLU masks, "amax" = the winning shot of the max aggregator.  `pres`
    (a dict) then receives this function's own pre-activations under the same keys, so that a caller can prove every
    disagreement to sit on a rounding-level tie."""
    T, Nq = qry_x.shape[:2]
    Nc = ctx_x.shape[1]
    dim_w = p["encoder_w0.8.weight"].shape[0]
    dim_z = p["r_to_z.weight"].shape[0]
    routes = routes or {}

    def enc(x, key):
        x = x.reshape(-1, *x.shape[2:])
        if key not in routes:
            return vanilla_encoder(x, p)
        f, pre = vanilla_encoder_routed(x, p, *routes[key])
        if pres is not None:
            pres[key] = {k_: v.detach() for k_, v in pre.items()}
        return f

    def pre_list(key):
        if pres is None:
            return None
        pres[key] = []
        return pres[key]

    x_qry = enc(qry_x, "enc_qry").reshape(T, Nq, dim_w)
    t = {}
    if Nc:
        x_ctx = enc(ctx_x, "enc_ctx").reshape(T, Nc, dim_w)
        ly = F.linear(ctx_y, p["transform_y.weight"], p["transform_y.bias"])
        n_hidden = sum(1 for k_ in p if k_.startswith("encoder_r.layers.") and k_.endswith(".weight")) - 1
        rs = encoder_fc(torch.cat([x_ctx, ly], dim=2), p, n_hidden=n_hidden, masks=routes.get("h"), pre=pre_list("h"))
        if agg_mode == "attention":
            r = multihead_attention(x_ctx, rs, x_qry, p, taps=t)
            z = F.linear(r, p["r_to_z.weight"], p["r_to_z.bias"])
        else:
            if agg_mode == "mean":
                r = agg_mean(rs)
            elif agg_mode == "max":
                r = agg_max(rs, routes.get("amax"))
                if pres is not None:
                    pres["rs"] = rs.detach()
            elif agg_mode == "baco":
                mu = F.linear(rs, p["rs_to_mu.weight"], p["rs_to_mu.bias"])
                var = 1e-5 + F.softplus(F.linear(rs, p["rs_to_var.weight"], p["rs_to_var.bias"]))
                r, _ = agg_baco(mu, var)
            else:
                raise TypeError(f"agg_mode {agg_mode!r} is not applicable")
            z = F.linear(r, p["r_to_z.weight"], p["r_to_z.bias"])[:, None, :].expand(T, Nq, dim_z)
        t.update(x_ctx=x_ctx, rs=rs, r=r)
    else:
        z = torch.zeros(T, Nq, dim_z)
    mu = decoder_mlp(torch.cat([x_qry, z], dim=-1), p, tanh=tanh, masks=routes.get("d"), pre=pre_list("d"))
    if taps is not None:
        taps.update(t, x_qry=x_qry, z=z)
    return mu


# --------------------------------------------------------------------------------------
# E2 / D2  ResNet encoder, NPDecoder and the ResNet-based CNP / ANP   (models.py:63-192,
#          ResNet.py:58-74, CondNeuralProcess.py:77-119, ANP.py:100-130; Appendix A.5)
# --------------------------------------------------------------------------------------


def resnet_features(img, p, prefix, img_agg, skip_pad=0, route=None, pre=None):
    """5x5 s2 p2 stem + ReLU; 4 x {conv3x3 s2 + ReLU, conv3x3 s1, + skip conv (1x1 s2, or 3x3 p1 s2
    in the BBB twin), ReLU}; then img_agg.  img [n,C,H,W] -> [n,F].
    `route`: optional list of 9 {0,1} masks (stem, then per block: conv1 output, block output) that
    REPLACE the ReLU decisions (pinned routing, see vanilla_encoder_routed); `pre`: optional list that
    receives the 9 pre-activations."""
    masks = iter(route) if route is not None else None

    def act(v):
        if pre is not None:
            pre.append(v.detach())
        return v * next(masks) if masks is not None else F.relu(v)

    x = act(F.conv2d(img, p[prefix + "conv1.weight"], p[prefix + "conv1.bias"], stride=2, padding=2))
    for i in range(1, 5):
        q = f"{prefix}resnet.layer{i}.0."
        out = act(F.conv2d(x, p[q + "conv1.weight"], p[q + "conv1.bias"], stride=2, padding=1))
        out = F.conv2d(out, p[q + "conv2.weight"], p[q + "conv2.bias"], stride=1, padding=1)
        idn = F.conv2d(x, p[q + "downsample.0.weight"], p[q + "downsample.0.bias"], stride=2, padding=skip_pad)
        x = act(out + idn)
    if img_agg in ("max", "baco"):
        x = F.adaptive_max_pool2d(x, (2, 2))
    elif img_agg == "mean":
        x = F.adaptive_avg_pool2d(x, (1, 1))
    return x.reshape(x.shape[0], -1)


def resnet_np_forward(p, ctx_x, ctx_y, qry_x, agg_mode, img_agg, n_heads=8, routes=None, pres=None):
    """CondNeuralProcess / ANP forward.  ctx_x [T,Nc,C,H,W], ctx_y [T,Nc,L], qry_x [T,Nq,C,H,W] -> mu [T,Nq,y].
    `routes` / `pres`: per encoder pass (context, [target,] decoder - in that order) the pinned ReLU
    masks / a list receiving the pre-activations (see resnet_features)."""
    T, Nq = qry_x.shape[:2]
    Nc = ctx_x.shape[1]
    flat = lambda t: t.reshape(-1, *t.shape[2:])
    passes = iter(routes) if routes is not None else None

    def resnet_features(img, p_, prefix, agg):      # noqa: F811 - routed wrapper around the module-level function
        pre = None
        if pres is not None:
            pre = []
            pres.append(pre)
        return globals()["resnet_features"](img, p_, prefix, agg, route=next(passes) if passes is not None else None, pre=pre)

    if Nc:
        x_ctx = resnet_features(flat(ctx_x), p, "img_encoder.", img_agg).reshape(T, Nc, -1)
        if "transform_y.weight" in p:      # *Distractor plugins (CNPDistractor.py:43,89 / ANPDistractor.py:46,114)
            ctx_y = F.linear(ctx_y, p["transform_y.weight"], p["transform_y.bias"])
        h = torch.cat([x_ctx, ctx_y], dim=2)
        for i in (0, 2, 4):
            h = F.relu(F.linear(h, p[f"task_encoder.{i}.weight"], p[f"task_encoder.{i}.bias"]))
        if agg_mode == "attention":
            x_tgt = resnet_features(flat(qry_x), p, "img_encoder.", img_agg).reshape(T, Nq, -1)
            r = multihead_attention(x_ctx, h, x_tgt, p, n_heads=n_heads)
            sample = F.linear(r, p["mu.weight"], p["mu.bias"])
        else:
            if agg_mode == "mean":
                r = agg_mean(h)
            elif agg_mode == "max":
                r = agg_max(h)
            elif agg_mode == "baco":
                mu_l = F.linear(h, p["latent_mu.weight"], p["latent_mu.bias"])
                var = 1e-5 + F.softplus(F.linear(h, p["latent_var.weight"], p["latent_var.bias"]))
                r, _ = agg_baco(mu_l, var)
            else:
                raise TypeError(agg_mode)
            sample = F.linear(r, p["mu.weight"], p["mu.bias"])[:, None, :].expand(T, Nq, -1)
    else:
        sample = torch.zeros(T, Nq, 256, dtype=qry_x.dtype)
    x_dec = resnet_features(flat(qry_x), p, "decoder.", img_agg).reshape(T, Nq, -1)
    h = torch.cat([x_dec, sample], dim=-1)
    h = F.relu(F.linear(h, p["decoder.fc_mu.0.weight"], p["decoder.fc_mu.0.bias"]))
    h = F.relu(F.linear(h, p["decoder.fc_mu.2.weight"], p["decoder.fc_mu.2.bias"]))
    return F.linear(h, p["decoder.fc_mu.4.weight"], p["decoder.fc_mu.4.bias"])


# --------------------------------------------------------------------------------------
# B1  Bayes-by-backprop encoder of ANPMRShapeNet3D      (bbb/BBBConv.py:86-108, bbb/misc.py:36-45,
#     ANPMRShapeNet3D.py:40-90,185-218; Appendix A.6)
# --------------------------------------------------------------------------------------


def bbb_sample(mu, rho):
    """W = mu + eps * log1p(exp(rho)), eps ~ N(0,1) from the torch CPU generator; and the layer's KL as
    the reference literally computes it: calculate_kl(0, 0.1, mu, sigma)."""
    eps = torch.empty(mu.size()).normal_(0, 1).to(mu.dtype)
    sigma = torch.log1p(torch.exp(rho))
    kl = 0.5 * (2 * torch.log(sigma / 0.1) - 1 + (0.1 / sigma).pow(2) + (mu / sigma).pow(2)).sum()
    return mu + eps * sigma, kl


def bbb_resnet_features(img, p, prefix="img_encoder.net.", route=None, pre=None):
    """BBB twin of resnet_features: every conv samples (weight, then bias) on each call, the skip is a
    3x3 p1 s2 conv, output flattened to [n,256].  Returns (features, kl summed over the 13 convs)."""
    masks = iter(route) if route is not None else None
    kls = []

    def act(v):
        if pre is not None:
            pre.append(v.detach())
        return v * next(masks) if masks is not None else F.relu(v)

    def conv(x, q, stride, pad):
        w, klw = bbb_sample(p[q + "W_mu"], p[q + "W_rho"])
        b, klb = bbb_sample(p[q + "bias_mu"], p[q + "bias_rho"])
        kls.append(klw + klb)
        return F.conv2d(x, w, b, stride=stride, padding=pad)

    x = act(conv(img, prefix + "layer1.conv.", 2, 2))
    for i in range(2, 6):
        q = f"{prefix}layer{i}."
        out = act(conv(x, q + "conv1.", 2, 1))
        out = conv(out, q + "conv2.", 1, 1)
        idn = conv(x, q + "downsample.0.", 2, 1)
        x = act(out + idn)
    return x.reshape(-1, 256), sum(kls)


def anpmr3d_forward(p, ctx_x, ctx_y, qry_x, img_agg="reshape", n_heads=8, routes=None, pres=None):
    """ANPMRShapeNet3D.forward with a non-empty context -> (mu [T,Nq,y], kl)."""
    T, Nq = qry_x.shape[:2]
    Nc = ctx_x.shape[1]
    flat = lambda t: t.reshape(-1, *t.shape[2:])
    passes = iter(routes) if routes is not None else None

    def nxt():
        pre = None
        if pres is not None:
            pre = []
            pres.append(pre)
        return dict(route=next(passes) if passes is not None else None, pre=pre)

    x_ctx, _ = bbb_resnet_features(flat(ctx_x), p, **nxt())
    x_tgt, kl = bbb_resnet_features(flat(qry_x), p, **nxt())      # second, independent sample; its kl is returned
    x_ctx, x_tgt = x_ctx.reshape(T, Nc, -1), x_tgt.reshape(T, Nq, -1)
    h = torch.cat([x_ctx, ctx_y], dim=2)
    for i in (0, 2, 4):
        h = F.relu(F.linear(h, p[f"task_encoder.{i}.weight"], p[f"task_encoder.{i}.bias"]))
    sample = F.linear(multihead_attention(x_ctx, h, x_tgt, p, n_heads=n_heads), p["mu.weight"], p["mu.bias"])
    x_dec = resnet_features(flat(qry_x), p, "decoder.", img_agg, **nxt()).reshape(T, Nq, -1)
    h = torch.cat([x_dec, sample], dim=-1)
    h = F.relu(F.linear(h, p["decoder.fc_mu.0.weight"], p["decoder.fc_mu.0.bias"]))
    h = F.relu(F.linear(h, p["decoder.fc_mu.2.weight"], p["decoder.fc_mu.2.bias"]))
    return F.linear(h, p["decoder.fc_mu.4.weight"], p["decoder.fc_mu.4.bias"]), kl


def bbb_vanilla_encoder(img, p, prefix="encoder_w0.net.", route=None, pre=None):
    """MR twin of vanilla_encoder (ANPMR.py:40-53): the same stack with every weight and bias re-sampled
    (layer1.conv, layer2.conv, layer3.conv, linear; weight then bias).  Returns (features, kl of the call).
    `route` = (m1, arg2, m2, m3): pinned routing as in vanilla_encoder_routed; `pre` (a dict) receives the pre-activations."""
    kls, w = [], {}
    for name in ("layer1.conv.", "layer2.conv.", "layer3.conv.", "linear."):
        w[name + "w"], kw = bbb_sample(p[prefix + name + "W_mu"], p[prefix + name + "W_rho"])
        w[name + "b"], kb = bbb_sample(p[prefix + name + "bias_mu"], p[prefix + name + "bias_rho"])
        kls.append(kw + kb)
    if route is not None:
        q = {"e.0.weight": w["layer1.conv.w"], "e.0.bias": w["layer1.conv.b"], "e.2.weight": w["layer2.conv.w"],
             "e.2.bias": w["layer2.conv.b"], "e.5.weight": w["layer3.conv.w"], "e.5.bias": w["layer3.conv.b"],
             "e.8.weight": w["linear.w"], "e.8.bias": w["linear.b"]}
        f, pr = vanilla_encoder_routed(img, q, *route, prefix="e.")
        if pre is not None:
            pre.update({k_: v.detach() for k_, v in pr.items()})
        return f, sum(kls)
    a1 = F.relu(F.conv2d(img, w["layer1.conv.w"], w["layer1.conv.b"], stride=2, padding=1))
    a2 = F.relu(F.conv2d(a1, w["layer2.conv.w"], w["layer2.conv.b"], stride=2, padding=1))
    a3 = F.relu(F.conv2d(F.max_pool2d(a2, 2), w["layer3.conv.w"], w["layer3.conv.b"], stride=2, padding=1))
    return F.linear(a3.reshape(a3.shape[0], -1), w["linear.w"], w["linear.b"]), sum(kls)


def vanilla_mr_forward(p, ctx_x, ctx_y, qry_x, agg_mode, attention, tanh, routes=None, pres=None):
    """ANPMR / ANPMRShapeNet1D (attention=True: targets encoded first, ANPMR.py:183-186) and CNPMR /
    CNPMRShapeNet1D (context first, targets last, CNPMR.py:136-166).  Returns (mu, kl of the target pass).
    `routes` / `pres`: per encoder call, in call order, the pinned routing (m1, arg2, m2, m3) / a list receiving one dict
    of pre-activations per call."""
    T, Nq = qry_x.shape[:2]
    Nc = ctx_x.shape[1]
    dim_w = p["encoder_w0.net.linear.W_mu"].shape[0]
    dim_z = p["r_to_z.weight"].shape[0]
    passes = iter(routes) if routes is not None else None

    def enc(x, n):
        pre = None
        if pres is not None:
            pre = {}
            pres.append(pre)
        f, kl_ = bbb_vanilla_encoder(x.reshape(T * n, *x.shape[2:]), p, route=next(passes) if passes is not None else None, pre=pre)
        return f.reshape(T, n, dim_w), kl_

    if attention:
        x_qry, kl = enc(qry_x, Nq)
    if Nc:
        x_ctx, _ = enc(ctx_x, Nc)
        ly = F.linear(ctx_y, p["transform_y.weight"], p["transform_y.bias"])
        n_hidden = sum(1 for k_ in p if k_.startswith("encoder_r.layers.") and k_.endswith(".weight")) - 1
        rs = encoder_fc(torch.cat([x_ctx, ly], dim=2), p, n_hidden=n_hidden)
        if attention:
            z = F.linear(multihead_attention(x_ctx, rs, x_qry, p), p["r_to_z.weight"], p["r_to_z.bias"])
        else:
            if agg_mode == "mean":
                r = agg_mean(rs)
            elif agg_mode == "max":
                r = agg_max(rs)
            elif agg_mode == "baco":
                r, _ = agg_baco(F.linear(rs, p["rs_to_mu.weight"], p["rs_to_mu.bias"]),
                                1e-5 + F.softplus(F.linear(rs, p["rs_to_var.weight"], p["rs_to_var.bias"])))
            else:
                raise TypeError(f"agg_mode {agg_mode!r} is not applicable")
            z = F.linear(r, p["r_to_z.weight"], p["r_to_z.bias"])[:, None, :].expand(T, Nq, dim_z)
    else:
        z = torch.zeros(T, Nq, dim_z)
    if not attention:
        x_qry, kl = enc(qry_x, Nq)
    return decoder_mlp(torch.cat([x_qry, z], dim=-1), p, tanh=tanh), kl


# --------------------------------------------------------------------------------------
# L1  losses                                                   (trainer/losses.py:32-80)
# --------------------------------------------------------------------------------------


def azimuth_loss(gt, pr):
    return ((gt[..., :2] - pr) ** 2).sum(dim=-1).mean()


def mean_square_loss(gt, pr):
    return ((gt - pr) ** 2).mean()


def quaternion_loss(gt, pr):
    pr = pr / pr.pow(2).sum(dim=-1, keepdim=True).sqrt()
    pos = (gt - pr).abs().sum(dim=-1)
    neg = (-gt - pr).abs().sum(dim=-1)
    return torch.minimum(pos, neg).mean()


def degree_loss(gt, pr):
    g = torch.rad2deg(gt[..., -1])
    ang = torch.acos(pr[..., 0])
    ang = torch.where(pr[..., 1] < 0, 2 * math.pi - ang, ang)
    d = torch.rad2deg(ang)
    err = torch.stack([(g - d).abs(), (g + 360.0 - d).abs(), (g - (d + 360.0)).abs()], dim=-1)
    return err.min(dim=-1)[0].mean()


def distractor_loss(gt, pr):
    return ((gt - pr) ** 2).sum(dim=-1).sqrt().mean()


def calc_loss(task, mu, gt, test=False):
    if task == "shapenet_1d":
        return degree_loss(gt, mu) if test else azimuth_loss(gt, mu)
    if task == "pascal_1d":
        return mean_square_loss(gt, mu)
    if task == "shapenet_3d":
        return quaternion_loss(gt, mu)
    if task == "distractor":
        return distractor_loss(gt, mu)
    raise TypeError(task)


# --------------------------------------------------------------------------------------
# X1  ConvEmbeddingModel (MMAML task embedding)      (conv_embedding_model.py:99-184)
# --------------------------------------------------------------------------------------


def conv_embedding_forward(x, p, num_conv=4, pooling="avg", bn_eps=1e-5):
    """4x{conv3x3 s2 p1, train-mode batch norm over the shots of ONE task, ReLU},
    spatial mean, Linear+ReLU, avg/max pool over the shot axis, one Linear per head."""
    for i in range(1, num_conv + 1):
        x = F.conv2d(x, p[f"conv.conv{i}.weight"], p[f"conv.conv{i}.bias"], stride=2, padding=1)
        mean = x.mean(dim=(0, 2, 3), keepdim=True)
        var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        x = (x - mean) / torch.sqrt(var + bn_eps)
        x = x * p[f"conv.bn{i}.weight"].view(1, -1, 1, 1) + p[f"conv.bn{i}.bias"].view(1, -1, 1, 1)
        x = F.relu(x)
    x = x.reshape(x.shape[0], x.shape[1], -1).mean(dim=2)
    hid = F.relu(F.linear(x, p["linear.weight"], p["linear.bias"]))
    emb = hid.mean(dim=0, keepdim=True) if pooling == "avg" else hid.max(dim=0, keepdim=True)[0]
    n_heads = sum(1 for k_ in p if k_.startswith("_embeddings.") and k_.endswith(".weight"))
    return [F.linear(emb, p[f"_embeddings.{i}.weight"], p[f"_embeddings.{i}.bias"]) for i in range(n_heads)]


# --------------------------------------------------------------------------------------
# Batch ingest (the loaders' host-side image conversion)   (dataset/shapenet_1d.py:189-196, utils/utils.py:26-30)
# --------------------------------------------------------------------------------------


def ingest_images(u8, div=255.0):
    """uint8 [T, N, H, W, C] -> float32 [T, N, C, H, W]: numpy `astype(float32) / 255.0` (shapenet_1d.py:189-190, same
    line in pascal_1d.py / shapenet_3d.py / distractor.py), then convert_channel_last_np_to_tensor's
    `permute(0, 1, 4, 2, 3).contiguous()` (utils/utils.py:26-30)."""
    x = np.asarray(u8).astype(np.float32) / div
    return torch.from_numpy(x).type(torch.FloatTensor).permute(0, 1, 4, 2, 3).contiguous()


# --------------------------------------------------------------------------------------
# Functional contrastive learning (FCL*): NT-Xent on task embeddings   (trainer/losses.py:82-99, FCLCNPShapeNet1D.py:101-159,
# FCLCNPDistractor.py:82-147, FCLANP.py:108-137).  The loss itself lives in the un-vendored dependency
# pytorch_metric_learning (requirements.txt:12, no version pinned; absent from this image): `nt_xent` restates the published
# NTXentLoss algorithm (cosine similarity, all same-label pairs positive, all different-label pairs negatives of their anchor,
# mean over positive pairs) literally, pair by pair.  PARITY OF THIS TERM IS UNPINNED: the fixture generator has to stub the
# missing package with this same restatement; everything around it (embeddings, mu, regression loss) is pinned by the reference.
# --------------------------------------------------------------------------------------


def nt_xent(z, labels, t=0.07):
    zn = z / z.norm(dim=1, keepdim=True).clamp_min(1e-12)
    sim = zn @ zn.t()
    n = z.shape[0]
    losses = []
    tiny = torch.finfo(z.dtype).tiny
    for a in range(n):
        negs = [sim[a, k] / t for k in range(n) if int(labels[k]) != int(labels[a])]
        if not negs:
            continue
        negs = torch.stack(negs)
        for q in range(n):
            if q == a or int(labels[q]) != int(labels[a]):
                continue
            pos = sim[a, q] / t
            m = torch.max(pos, negs.max()).detach()
            num = torch.exp(pos - m)
            den = torch.exp(negs - m).sum() + num
            losses.append(-torch.log(num / den + tiny))
    return torch.stack(losses).mean() if losses else z.sum() * 0.0


def contrastive_loss(z_1, z_2, t=0.07):
    labels = list(range(z_1.shape[0])) + list(range(z_2.shape[0]))
    return nt_xent(torch.cat((z_1, z_2), dim=0), labels, t)


def contrastive_loss_anp(z, t=0.07):
    labels = [i for i in range(z.shape[0]) for _ in range(z.shape[1])]
    return nt_xent(z.reshape(-1, z.shape[-1]), labels, t)


def fcl_cnp_vanilla_forward(p, ctx_x, ctx_y, qry_x, qry_y, agg_mode, test=False):
    """FCLCNPShapeNet1D.forward: the CNPShapeNet1D forward plus the target-set embedding (always max-aggregated, line 147)."""
    taps = {}
    mu = vanilla_np_forward(p, ctx_x, ctx_y, qry_x, agg_mode, tanh=True, taps=taps)
    if test:
        return mu, 0.0
    n_hidden = sum(1 for k_ in p if k_.startswith("encoder_r.layers.") and k_.endswith(".weight")) - 1
    z_0 = F.linear(taps["r"], p["r_to_z.weight"], p["r_to_z.bias"])
    ly = F.linear(qry_y, p["transform_y.weight"], p["transform_y.bias"])
    rq = agg_max(encoder_fc(torch.cat([taps["x_qry"], ly], dim=2), p, n_hidden=n_hidden))
    z_q = F.linear(rq, p["r_to_z.weight"], p["r_to_z.bias"])
    return mu, contrastive_loss(z_0, z_q)


def fcl_resnet_forward(p, ctx_x, ctx_y, qry_x, qry_y, agg_mode, img_agg, temperature=0.07, test=False, routes=None, pres=None):
    """FCLANP.forward / FCLCNPDistractor.forward: the ResNet-encoder ANP / CNP forward plus the contrastive term.
    Encoder passes in call order: context, target (ANP: attention queries; CNP: the target-set embedding, skipped when `test`),
    decoder - `routes` / `pres` as in resnet_np_forward."""
    T, Nq = qry_x.shape[:2]
    Nc = ctx_x.shape[1]
    flat = lambda t_: t_.reshape(-1, *t_.shape[2:])
    passes = iter(routes) if routes is not None else None

    def features(img, prefix):
        pre = None
        if pres is not None:
            pre = []
            pres.append(pre)
        return resnet_features(img, p, prefix, img_agg, route=next(passes) if passes is not None else None, pre=pre)

    def task_features(x_img, labels):
        if "transform_y.weight" in p:
            labels = F.linear(labels, p["transform_y.weight"], p["transform_y.bias"])
        h_ = torch.cat([x_img, labels], dim=2)
        for i in (0, 2, 4):
            h_ = F.relu(F.linear(h_, p[f"task_encoder.{i}.weight"], p[f"task_encoder.{i}.bias"]))
        return h_

    def embed(h_, quirk):
        if agg_mode == "mean":
            r = agg_mean(h_)
        elif agg_mode == "max":
            r = agg_max(h_)
        else:
            mu_l = F.linear(h_, p["latent_mu.weight"], p["latent_mu.bias"])
            # the target-set path applies latent_var to latent_mu's output (FCLCNPDistractor.py:133-134)
            var = 1e-5 + F.softplus(F.linear(mu_l if quirk else h_, p["latent_var.weight"], p["latent_var.bias"]))
            r, _ = agg_baco(mu_l, var)
        return F.linear(r, p["mu.weight"], p["mu.bias"])

    x_ctx = features(flat(ctx_x), "img_encoder.").reshape(T, Nc, -1)
    h = task_features(x_ctx, ctx_y)
    contra = 0.0
    if agg_mode == "attention":
        x_tgt = features(flat(qry_x), "img_encoder.").reshape(T, Nq, -1)
        sample = F.linear(multihead_attention(x_ctx, h, x_tgt, p), p["mu.weight"], p["mu.bias"])
        if not test:
            contra = contrastive_loss_anp(sample, temperature)
    else:
        z_0 = embed(h, False)
        sample = z_0[:, None, :].expand(T, Nq, -1)
        if not test:
            x_qry = features(flat(qry_x), "img_encoder.").reshape(T, Nq, -1)
            contra = contrastive_loss(z_0, embed(task_features(x_qry, qry_y), True))
    x_dec = features(flat(qry_x), "decoder.").reshape(T, Nq, -1)
    hd = torch.cat([x_dec, sample], dim=-1)
    hd = F.relu(F.linear(hd, p["decoder.fc_mu.0.weight"], p["decoder.fc_mu.0.bias"]))
    hd = F.relu(F.linear(hd, p["decoder.fc_mu.2.weight"], p["decoder.fc_mu.2.bias"]))
    return F.linear(hd, p["decoder.fc_mu.4.weight"], p["decoder.fc_mu.4.bias"]), contra
