#!/usr/bin/env python3
"""Per-parameter gradient error of the tail flavours against the fp64 oracle (one step, ANPShapeNet1D T=2 5+5)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch
import mlhot
from mlhot import ops
from mlhot.synth import get_batch
from networks.ANPShapeNet1D import ANPShapeNet1D
from trainer.losses import LossFunc
from oracle import ref_cpu as O
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests import test_gpu_parity as TP

DEV = "cuda:0"
cfg = types.SimpleNamespace(device=torch.device(DEV), seed=2578, img_size=[128, 128, 1], tasks_per_batch=2, input_dim=3,
                            output_dim=2, agg_mode="attention", img_agg="", dim_w=64, n_hidden_units_r=[100, 100], dim_r=64,
                            dim_z=64, task="shapenet_1d")
model = ANPShapeNet1D(cfg).to(DEV)
L = mlhot.lib()
cx, qx, cy, qy = get_batch("shapenet_1d", 2, 5, 5, seed=100)
res = {}
for opt in [int(x) for x in sys.argv[1:]] or [0, 47, 63]:
    L.set_option("tail_spec", opt)
    model.zero_grad(set_to_none=True)
    ops.saved_taps = []
    mu = model(cx.to(DEV), cy.to(DEV), qx.to(DEV))[0]
    LossFunc("mse", "shapenet_1d").calc_loss(mu, None, qy.to(DEV)).backward()
    routes = TP._vanilla_routes(L, ops.saved_taps, 2, 5, 5, "attention")
    ops.saved_taps = None
    res[opt] = ({k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}, routes)
p = {k: v.detach().cpu().double().requires_grad_(v.is_floating_point() and "projection" not in k) for k, v in model.state_dict().items()}
r = next(iter(res.values()))[1]
r64 = {k: (tuple(t.double() if t.is_floating_point() else t for t in v) if isinstance(v, tuple) else [t.double() for t in v] if isinstance(v, list) else v) for k, v in r.items()}
mu = O.vanilla_np_forward(p, cx.double(), cy.double(), qx.double(), "attention", tanh=True, routes=r64, pres={})
O.calc_loss("shapenet_1d", mu, qy.double()).backward()
gmax = max(p[k].grad.abs().max().item() for k, _ in model.named_parameters())
print("gmax", gmax)
for k, _ in model.named_parameters():
    g = p[k].grad
    line = f"{k:28s} max|g|/gmax {g.abs().max().item() / gmax:9.2e}  err/tensor-max:"
    for opt, (gr, _) in res.items():
        line += f"  [{opt}] {(gr[k] - g).abs().max().item() / g.abs().max().item():9.2e}"
    print(line)
