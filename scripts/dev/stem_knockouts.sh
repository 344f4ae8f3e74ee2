out=gpurun_out/stem_knockouts.txt; : > $out
for rep in 1 2; do
for v in base nostore noload; do
  unset MLHOT_LIB
  [ $v != base ] && export MLHOT_LIB=$PWD/build_exp/libmlhot_stem_$v.so
  python bench.py --workload c5 --no-configs --no-cpu-baseline --no-extras --steps 60 --warmup 10 > gpurun_out/ab_tmp.json 2>/dev/null
  python - $v >> $out <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab_tmp.json")); k=d.get("kernel_us_per_step") or {}
print(f"{sys.argv[1]:10s} step {d['ms_per_step_event_median']:.4f} ms  trunk.stem {k.get('trunk.stem')}  stem.wgrad {k.get('trunk.bwd.stem.wgrad')}  conv1.b1 {k.get('trunk.conv1.b1')}")
PY
done; done; cat $out
