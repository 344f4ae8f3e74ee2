#!/bin/bash
# usage (GPU box): bash scripts/dev/dbgrun2.sh <kernel-name-substring> <dbg values...>  - rocprofv3 kernel time under dbg knock-outs
L=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for d in "$@"; do
  rm -rf /tmp/kk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kk -o s -- python3 $REPO/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extras --prof-steps 0 --dbg $d $DBG_EXTRA > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("/tmp/kk/**/*kernel_stats.csv", recursive=True)[0]
print($d, [(r["Name"][:50], round(float(r["AverageNs"]) / 1000, 2)) for r in csv.DictReader(open(f)) if "$L" in r["Name"]])
PY
done
