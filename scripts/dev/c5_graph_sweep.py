#!/usr/bin/env python3
"""One hipGraph per context size of the ShapeNet3D training draw; capture and replay each, synchronising after every step."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch
import bench
from mlhot import synth
from networks.ANPMRShapeNet3D import ANPMRShapeNet3D
from networks.bbb.eps import StagedEps
from trainer.losses import LossFunc
dev = torch.device("cuda:0")
T = 8
cfg = types.SimpleNamespace(device=dev, seed=2578, img_size=[64, 64, 4], tasks_per_batch=T, input_dim=4, output_dim=4,
                            agg_mode="attention", img_agg="reshape", task="shapenet_3d", temperature=0.07)
model = ANPMRShapeNet3D(cfg).to(dev)
cx, qx, cy, qy = synth.get_batch_3d(T, 15, 15, seed=4321, device=dev, task_aug=True)
px, py = torch.cat([cx, qx], 1), torch.cat([cy, qy], 1)
loss_fn = LossFunc("mse", "shapenet_3d")
eps = StagedEps(dev, source=sys.argv[1] if len(sys.argv) > 1 else "host")

def fb(b):
    model.zero_grad(set_to_none=True)
    mu, var, kl = model(b[0], b[2], b[1])
    (loss_fn.calc_loss(mu, var, b[3]) + 1e-7 * kl).backward()

torch.manual_seed(1)
b15 = (px[:, :15].contiguous(), px[:, 15:].contiguous(), py[:, :15].contiguous(), py[:, 15:].contiguous())
with eps.recording():
    fb(b15)
torch.cuda.synchronize()
graphs, keep = {}, []
order = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(1, 16))
for nc in order:
    b = (px[:, :nc].contiguous(), px[:, nc:].contiguous(), py[:, :nc].contiguous(), py[:, nc:].contiguous())
    keep.append(b)
    def step(b=b):
        eps.rewind()
        fb(b)
    eps.stage()
    with eps.active():
        graphs[nc] = bench._capture(step)
    torch.cuda.synchronize()
    print("captured", nc, flush=True)
for rep in range(2):
    for nc in order:
        eps.stage()
        graphs[nc].replay()
        torch.cuda.synchronize()
        print("replayed", nc, flush=True)
print("done")
