#!/bin/bash
# scripts/dev/fetch_by_kernel.sh: HBM fetch bytes per launch of the conv kernels (rocprofv3 --pmc FETCH_SIZE, c3, eager)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/fetch_q
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/fetch_q -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 0 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/fetch_q/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k:60s} launches {len(v):3d}  fetch per launch {2 * 1024 * sum(v) / len(v) / 1e6:8.1f} MB")
PY
