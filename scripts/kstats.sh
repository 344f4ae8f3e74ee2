#!/bin/bash
# quick per-kernel table (rocprofv3 --kernel-trace --stats of the hipGraph-replayed bench); run on the GPU box
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-steps 0 ${1:-} > $OUT/log 2>&1
tail -1 $OUT/log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), round(d['ms_per_step'],4))
except Exception as e: print('no bench line', e)"
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows:
    calls=int(r['Calls']); per=calls/28.0
    us=float(r['AverageNs'])/1000
    tot+=us*per
    print(f"{us:8.1f} us x{per:4.1f}  {r['Name'].split('(')[0][-60:]}")
print('sum per step us', round(tot,1))
PY
