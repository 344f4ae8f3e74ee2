#!/bin/bash
# per-kernel table of the c5 step (rocprofv3 --kernel-trace --stats over scripts/bench_c5.py); run on the GPU box
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/kstats_c5
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $REPO/scripts/bench_c5.py > $OUT/log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time ms', tot/1e6)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}%  {float(r['AverageNs'])/1000:8.1f} us x{int(r['Calls']):6d}  {r['Name'].split('(')[0][-90:]}")
PY
