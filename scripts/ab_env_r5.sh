#!/bin/bash
# scripts/ab_env_r5.sh: like ab_r5.sh, for switches that are environment variables.
# usage: scripts/ab_env_r5.sh "<bench args>" "<label>:<VAR=value ...>" ...   -> gpurun_out/ab_env_r5.txt
out=gpurun_out/ab_env_r5.txt; : > $out
args=$1; shift
for rep in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%:*}; envs=${spec#*:}
    env $envs python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 10 $args > gpurun_out/ab_tmp.json 2>gpurun_out/ab_tmp.err || tail -5 gpurun_out/ab_tmp.err >> $out
    python - "$label" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
fwd = (d.get("roofline") or {}).get("forward") or {}
print(f"{sys.argv[1]:24s} ms/step {d['ms_per_step']:.4f}  event median {d['ms_per_step_event_median']:.4f}  fwd {fwd.get('fwd_ms', float('nan')):.4f}  launches {d.get('launches_per_step')}  loss {d.get('final_loss')}")
PY
  done
done
cat $out
