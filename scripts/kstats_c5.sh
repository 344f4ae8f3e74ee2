#!/bin/bash
# rocprofv3 per-kernel statistics of the c5 workload (bench.py --workload c5), for profiles/<tag>_kernel_stats_c5.csv
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r02}
OUT=$REPO/gpurun_out/kstats_c5_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 $REPO/bench.py --workload c5 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --prof-steps 0 > $OUT/run.log 2>&1
find $OUT -name "*kernel_stats.csv" | head -3
