#!/bin/bash
# LDS bank-conflict share per kernel: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE over a few eager steps of a workload.
# usage (on the GPU box): bash scripts/lds_conflicts.sh c3|c5  -> gpurun_out/lds_<workload>/..., summary on stdout
REPO=${GRAFT_REPO_ROOT:-$PWD}
WL=${1:-c3}
OUT=$REPO/gpurun_out/lds_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT -o lds -- python3 $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 0 ${@:2} > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
path = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(path)):
    k = row["Kernel_Name"].replace("mlhot::", "")[:90]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"])
# conflict cycles per LDS instruction: a 2-way ds_write_b32 / ds_read_b32 adds one pass per instruction, so a kernel whose conflicts are
# all 2-way stores shows conflict_cycles / (conflicted instructions) ~ a constant, and SQ_WAIT_INST_LDS (waves stalled on the LDS
# queue) says whether anybody waited for it
print("%-92s %6s %12s %8s %12s %10s %12s %12s" % ("kernel", "calls", "lds_active", "conflict", "insts_lds", "cyc/inst", "confl/inst", "wait_lds"))
for k, v in rows[:40]:
    a = v["SQ_LDS_IDX_ACTIVE"]
    c = max(n[k], 1)
    il = v.get("SQ_INSTS_LDS", 0.0)
    print("%-92s %6d %12.0f %7.1f%% %12.0f %10.2f %12.2f %12.0f" % (k, n[k], a / c, 100.0 * v["SQ_LDS_BANK_CONFLICT"] / a if a else 0.0, il / c,
          a / il if il else 0.0, v["SQ_LDS_BANK_CONFLICT"] / il if il else 0.0, v.get("SQ_WAIT_INST_LDS", 0.0) / c))
PY
