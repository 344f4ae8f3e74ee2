#!/bin/bash
# usage (GPU box): bash scripts/dev/kstats_opt.sh <tag> [bench.py arguments]  -> gpurun_out/ks_<tag>/..., top kernels on stdout
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$REPO/gpurun_out/ks_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --prof-steps 0 "$@" > $OUT/log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-80s %4s %8.1f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1000))
PY
