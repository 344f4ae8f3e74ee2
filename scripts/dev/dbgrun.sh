#!/bin/bash
# usage (GPU box): bash scripts/dev/dbgrun.sh <label-substring> <dbg values...>   - per-label time of one kernel under dbg knock-outs (the kernel has to read g_opt.dbg: temporary plumbing, see profiles/r04_final_split_dgrad_knockouts.txt)
L=$1; shift
for d in "$@"; do
  MLHOT_BENCH_KERNELS=/tmp/k_$d.json python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --opt conv2_split=7 --dbg $d > /dev/null 2>&1
  python - <<PY
import json
k = json.load(open("/tmp/k_$d.json"))
print($d, {a: round(b["avg_us"], 1) for a, b in k.items() if "$L" in a})
PY
done
