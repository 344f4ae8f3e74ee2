"""Timeline of workgroup 0 of the trunk's 32x32 stride-2 weight-gradient kernel (block-1 conv1) in one c5 backward; needs a
-DMLHOT_TS build:  MLHOT_LIB=build_exp/libmlhot_ts.so python scripts/dev/trunk_wgrad_ts.py"""
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
loss_fn = LossFunc("mse", "shapenet_3d")
cx, qx, cy, qy = synth.get_batch_3d(T, 15, 15, seed=4321, device=dev, task_aug=True)
ts = torch.zeros(4096, dtype=torch.int64, device=dev)
L = mlhot.lib()
L.c.mlhot_dbg_tsbuf.argtypes = [ctypes.c_void_p]
assert L.c.mlhot_dbg_tsbuf(ts.data_ptr()) == 0
for it in range(3):
    model.zero_grad(set_to_none=True)
    mu, var, kl = model(cx, cy, qx)
    (loss_fn.calc_loss(mu, var, qy) + 1e-7 * kl).backward()
    torch.cuda.synchronize()
    r = ts.cpu()[420:448].tolist()
    print("bands of this job %d, slab rows %d; cycles from entry: first fetch issued at 0; bands [loop top, barrier passed, staged + barrier, MFMAs done]:" % (r[26], r[27]),
          [[r[1 + 4 * k + i] - r[0] for i in range(4)] for k in range(5) if r[1 + 4 * k]], "loop exit", r[24] - r[0], "slab stored", r[25] - r[0])

w = ts.cpu()[1024:1024 + 2048].view(1024, 2)
w = w[w[:, 0] > 0]
t0 = int(w[:, 0].min())
import numpy as np
st, en = (w[:, 0] - t0).numpy() / 100.0, (w[:, 1] - t0).numpy() / 100.0
print("workgroups %d: start us min/median/max %.1f %.1f %.1f | end us min/median/max %.1f %.1f %.1f | duration median %.1f" % (
    len(st), st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), np.median(en - st)))
print("starts by block index (every 32nd):", [round(float(x), 1) for x in st[::32]])
