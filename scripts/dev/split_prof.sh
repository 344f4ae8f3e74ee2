#!/bin/bash
# scripts/dev/split_prof.sh: kernel times of the encoder forward with both conv12 kernels (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
python scripts/dev/enc_fwd_time.py 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/split_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/split_prof -o sp -- python scripts/dev/enc_fwd_time.py > /dev/null 2>&1
grep -E "conv12|Name" gpurun_out/split_prof/sp_kernel_stats.csv | cut -c1-200
