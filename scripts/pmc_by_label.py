#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv of `bench.py --no-graph` (eager steps) -> per LAUNCH LABEL sums per step.

usage: pmc_by_label.py <label_sequence.json> <out.json> <workload> <counter_collection.csv> [<counter_collection.csv> ...]

The library's kernels run in the same order every step, and bench.py (MLHOT_BENCH_SEQ) writes the launch labels of one step in
that order.  The mlhot:: dispatches of the LAST whole steps of the profiled run are folded onto that sequence (position modulo the
number of launches per step), so every counter can be reported per label - also for the ResNet trunk, where one label covers
several template instantiations and one instantiation serves several labels.  FETCH_SIZE is doubled (gfx950 tallies 128-byte read
requests at 64 B, MI355X_MICROARCH.md) and both traffic counters are in KiB."""
import collections
import csv
import json
import sys

seq = json.load(open(sys.argv[1]))
out_path, workload, paths = sys.argv[2], sys.argv[3], sys.argv[4:]
n = len(seq)
res = {"_workload": workload,
       "_note": "per launch label and STEP: sums over the label's launches, means over the profiled steps; rocprofv3 --pmc passes of "
                "`bench.py --no-graph` (eager); hbm_bytes = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE (separate passes)"}
per_label = collections.defaultdict(lambda: collections.defaultdict(float))
for path in paths:
    rows = collections.defaultdict(dict)            # dispatch id -> counter -> value
    names = {}
    for row in csv.DictReader(open(path)):
        if "mlhot::" not in row["Kernel_Name"]:
            continue
        d = int(row["Dispatch_Id"])
        rows[d][row["Counter_Name"]] = rows[d].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        names[d] = row["Kernel_Name"]
    ids = sorted(rows)
    steps = len(ids) // n
    if steps < 1:
        raise SystemExit(f"{path}: {len(ids)} mlhot dispatches, fewer than one step of {n}")
    ids = ids[len(ids) - steps * n:]                # the last whole steps (the first dispatches are warm-up / recording calls)
    for k, d in enumerate(ids):
        label = seq[k % n]
        for c, v in rows[d].items():
            per_label[label][c] += v / steps
        per_label[label]["_launches"] += 1.0 / steps
for label, v in per_label.items():
    e = {"launches_per_step": round(v.pop("_launches") / max(len(paths), 1), 2)}
    e.update({c: val for c, val in v.items()})
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        e["fetch_bytes"], e["write_bytes"] = 2.0 * 1024.0 * v.get("FETCH_SIZE", 0.0), 1024.0 * v.get("WRITE_SIZE", 0.0)
        e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    if v.get("GRBM_GUI_ACTIVE"):
        e["mfma_util"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if v.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in v:
                e["frac_" + c[3:].lower()] = v[c] / v["SQ_WAVE_CYCLES"]
    res[label] = e
json.dump(res, open(out_path, "w"), indent=1)
top = sorted(((k, v) for k, v in res.items() if not k.startswith("_")), key=lambda kv: -kv[1].get("hbm_bytes", kv[1].get("SQ_BUSY_CYCLES", 0)))[:12]
for k, v in top:
    print(k, {a: (round(b, 3) if isinstance(b, float) and b < 100 else b) for a, b in v.items() if a in ("launches_per_step", "hbm_bytes", "mfma_util", "frac_wait_any")})
