"""Host eps draw of c5's plan (896 k normals per step) on 1 / 2 / 4 / 8 / 16 threads: ms per draw, and that numbers + final generator
state equal the one-thread draw's.  usage: python scripts/dev/eps_threads_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "what-matters-for-meta-learning_amd"))
from networks.bbb import eps as E

shapes = []
for _ in range(2):
    shapes += [(64, 3, 5, 5), (64,)]
    for _ in range(12):
        shapes += [(64, 64, 3, 3), (64,)]
ref = None
for threads in (1, 2, 4, 8, 16):
    st = E.StagedEps("cpu", threads=threads)
    st.shapes = list(shapes)
    st._plan()
    out = torch.zeros(st._total)
    torch.manual_seed(11)
    st.draw_host(out)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        st.draw_host(out)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    torch.manual_seed(11)
    st.draw_host(out)
    if ref is None:
        ref = (out.clone(), torch.get_rng_state())
    same = torch.equal(out, ref[0]) and torch.equal(torch.get_rng_state(), ref[1])
    print(f"threads {threads:2d}: pieces {len(st._pieces) if st._pieces else 1}  median {1e3 * ts[len(ts) // 2]:.3f} ms  min {1e3 * ts[0]:.3f} ms  same numbers and state: {same}", flush=True)
print("cpus", os.cpu_count())
