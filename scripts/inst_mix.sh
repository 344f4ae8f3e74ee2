#!/bin/bash
# Instruction mix per kernel (wave-instructions per launch, and per MFMA): rocprofv3 --pmc SQ_INSTS_* over a few eager steps.
# usage (GPU box): bash scripts/inst_mix.sh c3|c5 [extra bench.py arguments, e.g. --opt conv2_split=7]
REPO=${GRAFT_REPO_ROOT:-$PWD}
WL=${1:-c3}
OUT=$REPO/gpurun_out/mix_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT -o mix -- python3 $REPO/bench.py --workload $WL --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-extras --prof-steps 0 ${@:2} > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
path = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(path)):
    k = row["Kernel_Name"].replace("mlhot::", "")[:70]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
rows = sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_MFMA"])
names = ["SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"]
print("%-72s %5s %11s  per MFMA: %6s %6s %6s %6s %6s" % ("kernel", "calls", "mfma/launch", "valu-m", "salu", "lds", "vm_rd", "vm_wr"))
for k, v in rows[:30]:
    m = v["SQ_INSTS_MFMA"]
    if m <= 0: continue
    c = max(n[k], 1)
    print("%-72s %5d %11.0f            %6.2f %6.2f %6.2f %6.3f %6.3f" % (k, c, m / c, (v["SQ_INSTS_VALU"] - m) / m, v["SQ_INSTS_SALU"] / m, v["SQ_INSTS_LDS"] / m, v["SQ_INSTS_VMEM_RD"] / m, v["SQ_INSTS_VMEM_WR"] / m))
PY
