#!/bin/bash
# rocprofv3 kernel stats of the bench command (c3 by default): per-kernel average duration + the per-step sum -> gpurun_out/kstats_r5.txt
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/kstats_r5
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY' | tee $REPO/gpurun_out/kstats_r5.txt
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(path)))
calls = max(int(r["Calls"]) for r in rows if "conv12_fwd" in r["Name"] or "conv3x3" in r["Name"] or "stem" in r["Name"]) if rows else 1
tot = 0.0
for r in rows:
    c, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
    if c < 20: continue
    per_step = avg * c / calls
    tot += per_step
    print(f"{r['Name'][:86]:86s} calls {c:5d}  avg {avg:8.1f} us  per step {per_step:7.1f}")
print(f"sum per step {tot:.1f} us (steps counted: {calls})")
PY
