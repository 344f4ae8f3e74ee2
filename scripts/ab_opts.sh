#!/bin/bash
# A/B of run-time options: scripts/ab_opts.sh <workload> "<opts>" ["<opts>" ...]  -> one line per option string with the step time
# and the per-step microseconds of the kernel labels matching $AB_FILTER (default: tail).  Runs every variant twice, interleaved.
wl=$1; shift
for rep in 1 2; do
for o in "$@"; do
  MLHOT_OPTS="$o" MLHOT_BENCH_KERNELS=gpurun_out/kx.json python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bx.json 2>/dev/null
  python - "$o" <<'PY'
import json, os, sys
k = json.load(open('gpurun_out/kx.json')); b = json.load(open('gpurun_out/bx.json'))
f = os.environ.get("AB_FILTER", "tail")
sel = {n: v['us_per_step'] for n, v in k.items() if f in n}
print(f"[{sys.argv[1]}]", round(b['ms_per_step'], 4), "sum=%.1f" % sum(sel.values()), ' '.join(f"{n}={v:.1f}" for n, v in sel.items()))
PY
done
done
