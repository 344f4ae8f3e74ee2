"""Timeline of one replayed step out of a rocprofv3 --kernel-trace CSV: kernels in start order with the idle time in front of each
(which stream is waiting for which), busy / wall per step.  usage: step_timeline.py <kernel_trace.csv> [first-kernel-fragment]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "stem_kernel"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"] and "wgrad" not in r["Kernel_Name"]]
full = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if any("wsum" in r["Kernel_Name"] or "wgrad" in r["Kernel_Name"] for r in rows[a:b])]
walls = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1000 for a, b in full]
print("steps", len(full), "wall us (last 8):", [round(w, 1) for w in walls[-8:]])
a, b = full[-3]
t0 = int(rows[a]["Start_Timestamp"])
end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1000:8.1f} {(e - s) / 1000:7.1f} idle {(s - end) / 1000:7.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].split('(')[0].replace('mlhot::', '')[-64:]}")
    end = max(end, e)
