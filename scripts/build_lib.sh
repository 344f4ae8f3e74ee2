#!/bin/bash
# scripts/build_lib.sh [out.so] [extra hipcc flags...]: one-command build of the library with the per-kernel resource table of the
# kernels whose mangled name contains $KFILTER (registers, spills, scratch).
cd "$(dirname "$0")/../what-matters-for-meta-learning_amd/csrc" || exit 1
out=${1:-libmlhot.so}; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-comment -Wno-division-by-zero -Wno-pass-failed -mllvm -pragma-unroll-threshold=40000 -Rpass-analysis=kernel-resource-usage "$@" mlhot.hip -o "$out" 2> /tmp/build_lib.log
rc=$?
# the default output is the product library: record which sources it was built from (mlhot/build.py decides "stale" by that)
if [ $rc -eq 0 ] && [ "$out" = "libmlhot.so" ] && [ $# -eq 0 ]; then (cd .. && python3 -c "from mlhot.build import stamp_product; stamp_product()"); fi
grep -E "error" -A6 /tmp/build_lib.log | head -40
if [ -n "$KFILTER" ]; then
  grep -A12 "Function Name: .*$KFILTER" /tmp/build_lib.log | grep -E "Function Name|    VGPRs:|VGPRs Spill|ScratchSize" | sed 's/\[-Rpass.*//' | sed 's/.*remark: //'
fi
exit $rc
