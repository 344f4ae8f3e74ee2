#!/bin/bash
# A/B of compile-time kernel variants: scripts/kexp_variants.sh <workload> <lib>...  -> one line per library with the
# per-step microseconds of the heaviest kernel labels (bench.py's MLHOT_BENCH_KERNELS dump), or of the labels containing $KSHOW.
wl=$1; shift
for so in "$@"; do
  MLHOT_LIB=$PWD/$so MLHOT_BENCH_KERNELS=gpurun_out/kx.json python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bx.json 2>/dev/null
  python - "$so" <<'PY'
import json, os, sys
k = json.load(open('gpurun_out/kx.json')); b = json.load(open('gpurun_out/bx.json'))
top = sorted(k.items(), key=lambda kv: -kv[1]['us_per_step'])[:6]
flt = os.environ.get("KSHOW")
if flt:
    top = [kv for kv in sorted(k.items()) if any(f in kv[0] for f in flt.split(","))]
print(sys.argv[1], round(b['ms_per_step'], 4), ' '.join(f"{n}={v['us_per_step']:.1f}" for n, v in top))
PY
done
