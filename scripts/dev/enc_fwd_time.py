"""scripts/dev/enc_fwd_time.py: the vanilla encoder's forward (480 images) with the fp32 and the split-precision conv12 kernel:
largest difference of the features, time per call (device events over 50 calls).  Per-kernel time: run under
`rocprofv3 --kernel-trace --stats`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "what-matters-for-meta-learning_amd"))
from mlhot import lib  # noqa: E402

L = lib()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [((32, 1, 3, 3), 0.3), ((32,), 0.1), ((48, 32, 3, 3), 0.06), ((48,), 0.1), ((64, 48, 3, 3), 0.05), ((64,), 0.1), ((64, 4096), 0.02), ((64,), 0.1)]
plist = [(torch.randn(*s, generator=g) * a).to(dev) for s, a in shapes]
x = torch.rand(480, 1, 128, 128, generator=g).to(dev)
out = {}
for split in (0, 1, 0, 1):
    L.set_option("conv2_split", split)
    f, _, saved = L.enc_vanilla_fwd(x, None, plist, 64)
    out[split] = f.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        L.enc_vanilla_fwd(x, None, plist, 64)
    e1.record()
    torch.cuda.synchronize()
    print(f"conv2_split={split}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per encoder forward")
print("max |split - fp32| / max |fp32| =", float((out[1] - out[0]).abs().max() / out[0].abs().max()))
