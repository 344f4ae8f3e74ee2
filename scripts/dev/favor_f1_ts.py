"""Per-workgroup wall-clock stamps of FAVOR+ F1 in one c5 forward (-DMLHOT_TS build):
MLHOT_LIB=build_exp/libmlhot_ts.so python scripts/dev/favor_f1_ts.py"""
import ctypes, os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "what-matters-for-meta-learning_amd"), ROOT]
import torch
import mlhot
from mlhot import synth
from networks.ANPMRShapeNet3D import ANPMRShapeNet3D
dev = torch.device("cuda:0")
T = 8
cfg = types.SimpleNamespace(device=dev, seed=2578, img_size=[64, 64, 4], tasks_per_batch=T, input_dim=4, output_dim=4,
                            agg_mode="attention", img_agg="reshape", task="shapenet_3d", temperature=0.07)
model = ANPMRShapeNet3D(cfg).to(dev)
cx, qx, cy, qy = synth.get_batch_3d(T, 15, 15, seed=4321, device=dev, task_aug=True)
ts = torch.zeros(4096, dtype=torch.int64, device=dev)
L = mlhot.lib()
L.c.mlhot_dbg_tsbuf.argtypes = [ctypes.c_void_p]
assert L.c.mlhot_dbg_tsbuf(ts.data_ptr()) == 0
for it in range(3):
    with torch.no_grad():
        model(cx, cy, qx)
    torch.cuda.synchronize()
w = ts.cpu()[1024:1024 + 4 * 512].view(512, 4).numpy().astype(np.float64)
w = w[w[:, 0] > 0]
t0 = w[:, 0].min()
w = (w - t0) / 100.0
print("F1 workgroups %d: start us med/max %.1f %.1f | loop done med/max %.1f %.1f | dd stored med/max %.1f %.1f | exit med/max %.1f %.1f" % (
    len(w), np.median(w[:, 0]), w[:, 0].max(), np.median(w[:, 1]), w[:, 1].max(), np.median(w[:, 2]), w[:, 2].max(), np.median(w[:, 3]), w[:, 3].max()))
print("per-workgroup durations: loop med %.1f, stores med %.1f, tree med %.1f" % (np.median(w[:, 1] - w[:, 0]), np.median(w[:, 2] - w[:, 1]), np.median(w[:, 3] - w[:, 2])))

f2 = ts.cpu()[3200:3200 + 8 * 64].view(64, 8).numpy().astype(np.float64)
f2 = f2[f2[:, 0] > 0]
if len(f2):
    f2 = (f2 - f2[:, 0].min()) / 100.0
    d = np.diff(f2[:, :5], axis=1)
    print("F2 workgroups %d: start med/max %.1f %.1f, exit med/max %.1f %.1f us | phases (median us): preamble %.1f, feature loop %.1f, fold + D %.1f, out %.1f" % (
        len(f2), np.median(f2[:, 0]), f2[:, 0].max(), np.median(f2[:, 4]), f2[:, 4].max(), *np.median(d, axis=0)))
