#!/usr/bin/env python3
"""Which torch-side device ops (fills, copies) ride along with one step?  (diagnostic)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch
import bench
from mlhot import synth
from trainer.losses import LossFunc
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["c3"]
model = getattr(importlib.import_module("networks." + w["method"]), w["method"])(bench.make_cfg(w, dev)).to(dev)
loss_fn = LossFunc("mse", "shapenet_1d")
cx, qx, cy, qy = synth.get_batch("shapenet_1d", 16, 15, 15, seed=1234, device=dev)
def step():
    model.zero_grad(set_to_none=True)
    loss = loss_fn.calc_loss(model(cx, cy, qx)[0], None, qy)
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA or "Memcpy" in e.name or "copy" in e.name.lower() or "fill" in e.name.lower():
        print(e.name[:90], "| cpu_parent:", (e.cpu_parent.name if e.cpu_parent else None), "| dur us", e.device_time_total or e.cpu_time_total)
