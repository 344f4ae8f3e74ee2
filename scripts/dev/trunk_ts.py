#!/usr/bin/env python3
"""Band timeline of the trunk's conv3x3 kernel (16x16 stride-1 geometry, workgroup 0) from a -DMLHOT_TS build:
MLHOT_LIB=build_exp/libmlhot_ts.so python scripts/dev/trunk_ts.py"""
import ctypes, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch
import mlhot
from mlhot import synth
from networks.ANPMRShapeNet3D import ANPMRShapeNet3D
from trainer.losses import LossFunc
dev = torch.device("cuda:0")
T = 8
cfg = types.SimpleNamespace(device=dev, seed=2578, img_size=[64, 64, 4], tasks_per_batch=T, input_dim=4, output_dim=4,
                            agg_mode="attention", img_agg="reshape", task="shapenet_3d", temperature=0.07)
model = ANPMRShapeNet3D(cfg).to(dev)
cx, qx, cy, qy = synth.get_batch_3d(T, 15, 15, seed=4321, device=dev, task_aug=True)
ts = torch.zeros(512, dtype=torch.int64, device=dev)
L = mlhot.lib()
L.c.mlhot_dbg_tsbuf.argtypes = [ctypes.c_void_p]
assert L.c.mlhot_dbg_tsbuf(ts.data_ptr()) == 0
for it in range(4):
    model.zero_grad(set_to_none=True)
    with torch.no_grad():
        mu, var, kl = model(cx, cy, qx)
    torch.cuda.synchronize()
    r = ts.cpu()[300:321].tolist()
    print("prologue %d cycles; bands [stage start, staged, MFMAs done, stored]:" % (r[1] - r[0]),
          [[r[2 + 4 * k + i] - r[0] for i in range(4)] for k in range(3) if r[2 + 4 * k]], "exit", r[20] - r[0],
          "| epilogue of band 0 / last: loads issued", r[16] - r[0], r[17] - r[0], "first tile stored", r[18] - r[0], r[19] - r[0])
