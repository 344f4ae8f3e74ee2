#!/bin/bash
# scripts/ab_r5.sh: A/B lines of one gpurun call (box-to-box spread is +-3 %, so only numbers of ONE call compare).
# usage: scripts/ab_r5.sh "<label>:[LIB=<other build of the library> ]<bench args>" ...   -> gpurun_out/ab_r5.txt
out=gpurun_out/ab_r5.txt; : > $out
for rep in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%:*}; args=${spec#*:}
    unset MLHOT_LIB
    if [[ "$args" == LIB=* ]]; then export MLHOT_LIB=$PWD/${args%% *}; MLHOT_LIB=${MLHOT_LIB/LIB=/}; args=${args#* }; [[ "$args" == LIB=* ]] && args=""; fi
    python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 $args > gpurun_out/ab_tmp.json 2>/dev/null
    python - "$label" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d.get("kernel_us_per_step") or {}
fwd = (d.get("roofline") or {}).get("forward") or {}
print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:.4f}  event median {d['ms_per_step_event_median']:.4f}  fwd {fwd.get('fwd_ms', float('nan')):.4f}  launches {d.get('launches_per_step')}  "
      + " ".join(f"{n}={v}" for n, v in k.items() if n.startswith('tail') or 'dgrad2' in n or 'conv3' in n or 'wgrad' in n or 'wsum' in n or 'trunk.conv' in n or 'conv2.dgrad' in n or n in ('slab_reduce', 'loss_fwd', 'loss_bwd', 'enc.linear')))
PY
  done
done
cat $out
