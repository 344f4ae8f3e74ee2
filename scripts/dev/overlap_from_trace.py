"""Kernel-trace CSV (rocprofv3 --kernel-trace) -> how much of the busy time has two or more kernels in flight, and the per-kernel
overlap with its predecessor in start order, over the last `n` dispatches.  usage: overlap_from_trace.py <kernel_trace.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))[-n:]
t0 = ev[0][0]
busy = two = 0
pts = sorted([(s, 1) for s, e, *_ in ev] + [(e, -1) for s, e, *_ in ev])
depth, last = 0, pts[0][0]
for t, d in pts:
    if depth >= 1: busy += t - last
    if depth >= 2: two += t - last
    depth += d; last = t
print(f"{len(ev)} dispatches over {(ev[-1][1] - t0) / 1e3:.1f} us: busy {busy / 1e3:.1f} us, two or more in flight {two / 1e3:.1f} us")
prev_end = ev[0][0]
for s, e, name, q, st in ev[-120:]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} gap {(s - prev_end) / 1e3:7.1f} q={q} s={st} {name[:70]}")
    prev_end = max(prev_end, e)
