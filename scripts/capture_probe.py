#!/usr/bin/env python3
"""Which op family of the ResNet / Bayes-by-backprop path survives hipGraph capture (each case in its own process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["enc_kl_bwd", "model_step"]

if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        print(f"{c:14s} rc={r.returncode} {tail[:150]}", flush=True)
    sys.exit(0)

sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import types  # noqa: E402
import torch  # noqa: E402
from mlhot import ops  # noqa: E402
dev = torch.device("cuda", 0)
case = sys.argv[1]
g = torch.Generator().manual_seed(0)


def R(*s, grad=False):
    return torch.randn(*s, generator=g).to(dev).requires_grad_(grad)


if case in ("conv", "conv_bwd"):
    x, w, b = R(8, 3, 64, 64, grad=True), R(64, 3, 5, 5, grad=True), R(64, grad=True)

    def fn():
        y = ops.Conv2dFunction.apply(x, w, b, 2, 2, True)
        if case == "conv_bwd":
            x.grad = w.grad = b.grad = None
            y.sum().backward()
        return y
elif case == "bbb":
    mu, rho, eps = R(64, 64, 3, 3, grad=True), R(64, 64, 3, 3, grad=True), R(64, 64, 3, 3)

    def fn():
        mu.grad = rho.grad = None
        w, kl = ops.BBBSampleFunction.apply(mu, rho, eps)
        (w.sum() + kl).backward()
        return w
elif case in ("bbb_kl_only", "bbb_twice"):
    mu, rho = R(64, 64, 3, 3, grad=True), R(64, 64, 3, 3, grad=True)
    e1, e2 = R(64, 64, 3, 3), R(64, 64, 3, 3)

    def fn():
        mu.grad = rho.grad = None
        w, kl = ops.BBBSampleFunction.apply(mu, rho, e1)
        if case == "bbb_kl_only":
            (1e-7 * kl).backward()
        else:
            w2, kl2 = ops.BBBSampleFunction.apply(mu, rho, e2)
            (w.sum() + w2.sum() + 1e-7 * (kl + kl2)).backward()
        return w
elif case in ("favor", "favor_bwd"):
    from networks.fast_attention import FastAttention
    att = FastAttention(256, 1419).to(dev)
    q, k, v = R(2, 8, 15, 256, grad=True), R(2, 8, 15, 256, grad=True), R(2, 8, 15, 256, grad=True)

    def fn():
        o = att(q, k, v)
        if case == "favor_bwd":
            q.grad = k.grad = v.grad = None
            o.sum().backward()
        return o
elif case == "linear":
    x, w, b = R(120, 512, grad=True), R(256, 512, grad=True), R(256, grad=True)

    def fn():
        x.grad = w.grad = b.grad = None
        y = ops.LinearFunction.apply(x, w, b, "relu")
        y.sum().backward()
        return y
elif case == "addrelu_pool":
    a, b2 = R(8, 64, 16, 16, grad=True), R(8, 64, 16, 16, grad=True)
    names = [n for n in dir(ops) if "Relu" in n or "Pool" in n]
    print("ops:", names)

    def fn():
        a.grad = b2.grad = None
        y = ops.AddReluFunction.apply(a, b2)
        z = ops.MaxPool2Function.apply(y)
        z.sum().backward()
        return z
else:
    import importlib
    from trainer.losses import LossFunc
    from networks.bbb.eps import StagedEps
    T = 2
    cfg = types.SimpleNamespace(device=dev, seed=2578, img_size=[64, 64, 4], tasks_per_batch=T, input_dim=4, output_dim=4,
                                agg_mode="attention", img_agg="reshape", task="shapenet_3d", temperature=0.07, method="ANPMRShapeNet3D")
    model = importlib.import_module("networks.ANPMRShapeNet3D").ANPMRShapeNet3D(cfg).to(dev)
    cx, qx = R(T, 5, 3, 64, 64), R(T, 5, 3, 64, 64)
    cy = torch.nn.functional.normalize(R(T, 5, 4), dim=-1)
    qy = torch.nn.functional.normalize(R(T, 5, 4), dim=-1)
    loss_fn = LossFunc("mse", "shapenet_3d")
    st = StagedEps(dev)

    def raw():
        if case == "model_fwd":
            with torch.no_grad():
                return model(cx, cy, qx)[0]
        if case != "model_bwd_mu_nozero":
            model.zero_grad(set_to_none=True)
        if case in ("enc_bwd", "enc_kl_bwd"):
            f, kl = model.img_encoder(cx.reshape(-1, 3, 64, 64))
            (f.sum() if case == "enc_bwd" else 1e-7 * kl).backward()
            return f
        if case == "dec_bwd":
            smp = torch.ones(T, 5, 256, device=dev, requires_grad=True)
            out, _ = model.decoder(qx, smp)
            out.sum().backward()
            return out
        if case == "attn_bwd":
            a, b_, c_ = (torch.ones(T, 5, 256, device=dev, requires_grad=True) for _ in range(3))
            o = model._multihead_attention(a, b_, c_)
            o.sum().backward()
            return o
        if case == "mlp_bwd":
            from networks._resnet_np import _mlp3
            a = torch.ones(T, 5, 260, device=dev, requires_grad=True)
            o = _mlp3(a, model.task_encoder, last_relu=True)
            o.sum().backward()
            return o
        mu, _, kl = model(cx, cy, qx)
        if case == "model_fwd_grad":
            return mu
        if case in ("model_bwd_mu", "model_bwd_mu_nozero"):
            mu.sum().backward()
        elif case == "model_bwd_kl":
            (1e-7 * kl).backward()
        elif case == "model_bwd_loss":
            loss_fn.calc_loss(mu, None, qy).backward()
        else:
            (loss_fn.calc_loss(mu, None, qy) + 1e-7 * kl).backward()
        return mu

    with st.recording():
        raw()
    if st.shapes:
        st.stage()
        ctx = st.active()
        ctx.__enter__()

    def fn():
        st.rewind()
        return raw()

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fn()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("eager ok", flush=True)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = fn()
print("captured", flush=True)
graph.replay()
torch.cuda.synchronize()
print("replayed ok", float(out.float().abs().sum()), flush=True)
