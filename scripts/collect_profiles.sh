#!/bin/bash
# Run on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and HBM-traffic counters of the headline
# workload.  Everything lands under gpurun_out/prof/; scripts/summarise_profiles.py turns it into profiles/*.
# PMC passes are separate runs with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# EVERY <tag>_* summary comes out of THIS run: the raw directory is emptied first, the hash of the library's SOURCES (the box may
# rebuild the .so: file times do not survive the copy, and hipcc's output differs from box to box), the .so's own sha256 and the start
# time go into $OUT/MANIFEST.json, and scripts/summarise_profiles.py refuses raw files older than that or sources other than the tree's
# (round 4 committed a "final" set that mixed two library states).
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof
TAG=${1:-r06_final}
rm -rf $OUT
mkdir -p $OUT
cd $REPO
LIB=$REPO/what-matters-for-meta-learning_amd/csrc/libmlhot.so
python3 -c "import sys; sys.path.insert(0, '$REPO/what-matters-for-meta-learning_amd'); import mlhot; mlhot.build_product()"   # the box rebuilds when the copy's file times say so: do it BEFORE hashing
python3 - "$LIB" "$OUT/MANIFEST.json" "$TAG" "$REPO" <<'PY'
import hashlib, json, sys, time
sys.path.insert(0, sys.argv[4] + "/what-matters-for-meta-learning_amd")
from mlhot.build import source_sha256
json.dump({"tag": sys.argv[3], "src_sha256": source_sha256(), "lib_sha256": hashlib.sha256(open(sys.argv[1], "rb").read()).hexdigest(),
           "started_at": time.time()}, open(sys.argv[2], "w"))
PY
python bench.py --steps 50 --warmup 10 > $OUT/${TAG}_bench_c3.json 2> $OUT/bench_c3.err
python bench.py --steps 50 --warmup 10 --workload c2 --no-cpu-baseline > $OUT/${TAG}_bench_c2.json 2> $OUT/bench_c2.err
MLHOT_BENCH_KERNELS=$OUT/${TAG}_kernels_c3.json python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
# the bench command itself (pre-warm replays, timed region, and the roofline leg's eager steps, each behind a few replays)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
# the same without pre-warm and with the GPU idle in front of the eager steps (what the numbers of rounds 1-3 were taken in)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cold -o stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-prewarm > $OUT/stats_cold.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 0 > $OUT/pmc_write.log 2>&1
# SQ / GRBM pass: MFMA-pipe busy cycles and the wave-cycle split (active / issue-stalled / parked) of the hot kernels
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 0 > $OUT/pmc_sq.log 2>&1
# instruction mix and LDS bank conflicts of the same library (round 4's copies of these two were collected hours before its final build)
cd $REPO
bash scripts/inst_mix.sh c3 > $OUT/${TAG}_inst_mix_c3.txt 2>&1
bash scripts/lds_conflicts.sh c3 > $OUT/${TAG}_lds_conflicts_c3.txt 2>&1
# the parity log the GPU tests keep (flip counts, decisions, worst errors against the oracle and against the reference's fixtures)
rm -f $REPO/gpurun_out/parity_flips.txt
MLHOT_PARITY_LOG=$OUT/${TAG}_parity_flips.txt python -m pytest tests -m gpu -q -p no:cacheprovider -k "baseline_configs_vs_reference or resnet_models_vs_reference or anpmr_shapenet3d_vs_reference or c5_full_size or mr_vanilla_models or fcl_models or mid_size_case or edge_cases" > $OUT/parity_tests.log 2>&1
# BASELINE config c5's per-GPU step (ANPMRShapeNet3D, bench.py --workload c5): bench line, per-label kernel times, kernel stats
cd $REPO
python bench.py --workload c5 --steps 30 --warmup 5 > $OUT/${TAG}_bench_c5.json 2> $OUT/bench_c5.err
# the same step with the eps stream continued on the device (4 jump-ahead sub-streams): the mode the c5 kernel numbers are quoted in
python bench.py --workload c5 --eps device --steps 40 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_c5_device_eps.json 2> $OUT/bench_c5d.err
MLHOT_BENCH_KERNELS=$OUT/${TAG}_kernels_c5.json python bench.py --workload c5 --eps device --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o stats -- python3 $REPO/bench.py --workload c5 --eps device --steps 20 --warmup 5 --no-cpu-baseline --no-extras --prof-steps 0 > $OUT/stats_c5.log 2>&1
# c5 counters per launch label (scripts/pmc_by_label.py folds the dispatch list onto the label sequence of one step)
cd $REPO
MLHOT_BENCH_SEQ=$OUT/seq_c5.json python bench.py --workload c5 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 2 > /dev/null 2>&1
cd /tmp
C5="python3 $REPO/bench.py --workload c5 --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_c5 -o fetch -- $C5 > $OUT/pmc_fetch_c5.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_c5 -o write -- $C5 > $OUT/pmc_write_c5.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_c5 -o sq -- $C5 > $OUT/pmc_sq_c5.log 2>&1
cd $REPO
python3 scripts/pmc_by_label.py $OUT/seq_c5.json $OUT/${TAG}_pmc_traffic_c5.json c5 $(find $OUT/pmc_fetch_c5 -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write_c5 -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_by_label.py $OUT/seq_c5.json $OUT/${TAG}_pmc_sq_c5.json c5 $(find $OUT/pmc_sq_c5 -name "*counter_collection.csv" | head -1)
# the opt-in split-precision conv12 kernels (extras, never `value`): an A/B bench line, kernel stats, instruction mix, LDS conflicts,
# SQ counters and the error table against float64.  Frozen since round 5: SKIP_SPLIT=1 leaves the section out.
cd $REPO
if [ "${SKIP_SPLIT:-0}" = "1" ]; then ls -la $OUT; exit 0; fi
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras --opt conv2_split=7 > $OUT/${TAG}_split_bench_c3.json 2> $OUT/bench_split.err
python scripts/dev/split_error.py 8 > $OUT/${TAG}_split_error_vs_float64.txt 2> /dev/null
bash scripts/inst_mix.sh c3 --opt conv2_split=7 > $OUT/${TAG}_split_inst_mix.txt 2>&1
bash scripts/lds_conflicts.sh c3 --opt conv2_split=7 > $OUT/${TAG}_split_lds_conflicts.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_split -o stats -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --prof-steps 0 --opt conv2_split=7 > $OUT/stats_split.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_split -o sq -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 0 --opt conv2_split=7 > $OUT/pmc_sq_split.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -20
ls -la $OUT
