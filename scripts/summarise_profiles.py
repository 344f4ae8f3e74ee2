#!/usr/bin/env python3
"""gpurun_out/prof/ (scripts/collect_profiles.sh) -> profiles/<tag>_*: bench lines, rocprofv3 kernel-stats CSV and
per-launch HBM traffic of the hot kernels (FETCH_SIZE doubled per MI355X_MICROARCH.md: gfx950 tallies 128-B read
requests at 64 B; WRITE_SIZE as read; counter unit KiB)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r06_final"
force = "--force" in sys.argv

# ---- one library build behind every summary ----------------------------------------------------------------------------------
import hashlib
import time
try:
    MAN = json.load(open(os.path.join(SRC, "MANIFEST.json")))
except OSError:
    sys.exit("summarise_profiles: gpurun_out/prof/MANIFEST.json is missing - collect with scripts/collect_profiles.sh (it records the library's sha256)")
sys.path.insert(0, os.path.join(ROOT, "what-matters-for-meta-learning_amd"))
from mlhot.build import source_sha256
local_sha = source_sha256()
if MAN.get("tag") != tag:
    sys.exit(f"summarise_profiles: the raw directory was collected as {MAN.get('tag')!r}, not {tag!r}")
if local_sha != MAN.get("src_sha256") and not force:
    sys.exit(f"summarise_profiles: the profiles were collected from library sources {str(MAN.get('src_sha256'))[:12]}, the tree now holds {local_sha[:12]}: "
             "re-collect (or --force to summarise the older sources' numbers, which the manifest will say)")
# gpurun MERGES the box's gpurun_out/ into the local one, so leftovers of earlier collections sit beside this one's files: anything
# older than this collection's start is not part of it and is never read below (fresh() / find())
def fresh(path):
    return os.path.isfile(path) and os.path.getsize(path) > 0 and os.path.getmtime(path) >= MAN["started_at"] - 5


stale = [p for p in glob.glob(os.path.join(SRC, "**", "*"), recursive=True) if os.path.isfile(p) and not fresh(p) and not p.endswith("MANIFEST.json")]
if stale:
    print(f"summarise_profiles: ignoring {len(stale)} raw files older than this collection's start (leftovers of earlier collections)", file=sys.stderr)
WRITTEN = []
_copy = shutil.copy


def _tracked_copy(src, dst):
    WRITTEN.append(os.path.basename(dst))
    return _copy(src, dst)


shutil.copy = _tracked_copy
import atexit


def _manifest():
    json.dump({"tag": tag, "src_sha256": MAN.get("src_sha256"), "lib_sha256": MAN["lib_sha256"], "sources_match_tree": local_sha == MAN.get("src_sha256"),
               "collected_at": time.strftime("%Y-%m-%d %H:%M:%S", time.gmtime(MAN["started_at"])) + " UTC",
               "files": sorted(set(WRITTEN + [n for n in os.listdir(DST) if n.startswith(tag + "_") and n.endswith((".json", ".txt", ".csv"))
                                               and os.path.getmtime(os.path.join(DST, n)) >= MAN["started_at"]])),
               "note": "every file listed was produced by ONE run of scripts/collect_profiles.sh with the library built from the sources above "
                       "(src_sha256: csrc/*.hip, csrc/*.h, include/mlhot.h; lib_sha256: the .so that ran - the GPU box may rebuild it from the same sources)"},
              open(os.path.join(DST, f"{tag}_MANIFEST.json"), "w"), indent=1)


atexit.register(_manifest)

KERNEL_LABEL = {            # kernel-name fragment -> bench.py label
    "conv12_fwd_pool_kernel": "enc.conv12", "conv12_wgrad_kernel": "enc.bwd.conv12.wgrad", "conv12_dgrad_kernel": "enc.bwd.conv12.dgrad",
    "conv3_fwd_kernel": "enc.conv3", "conv3_wgrad_kernel": "enc.bwd.conv3.wgrad", "conv3_dgrad_kernel": "enc.bwd.conv3.dgrad",
    "conv3_bwd_kernel": "enc.bwd.conv3",
}


def find(pattern):
    hits = [h for h in sorted(glob.glob(os.path.join(SRC, pattern), recursive=True)) if fresh(h)]
    return hits[0] if hits else None


for name in (f"{tag}_bench_c3.json", f"{tag}_bench_c2.json", f"{tag}_kernels_c3.json", f"{tag}_bench_c5.json", f"{tag}_bench_c5_device_eps.json", f"{tag}_kernels_c5.json",
             f"{tag}_pmc_traffic_c5.json", f"{tag}_pmc_sq_c5.json", f"{tag}_inst_mix_c3.txt", f"{tag}_lds_conflicts_c3.txt", f"{tag}_parity_flips.txt"):
    p = os.path.join(SRC, name)
    if fresh(p):
        shutil.copy(p, os.path.join(DST, name))
if fresh(os.path.join(SRC, f"{tag}_pmc_traffic_c5.json")):
    shutil.copy(os.path.join(SRC, f"{tag}_pmc_traffic_c5.json"), os.path.join(DST, "pmc_traffic_c5.json"))      # what bench.py --workload c5 reads
stats = find("stats/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(DST, f"{tag}_kernel_stats.csv"))
cold = find("stats_cold/**/*kernel_stats.csv")
if cold:
    shutil.copy(cold, os.path.join(DST, f"{tag}_kernel_stats_cold.csv"))
stats_c5 = find("stats_c5/**/*kernel_stats.csv")
if stats_c5:
    shutil.copy(stats_c5, os.path.join(DST, f"{tag}_kernel_stats_c5.csv"))


def pmc(dirname, counter):
    path = find(f"{dirname}/**/*counter_collection.csv")
    per = {}
    if not path:
        return per
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            kname = row.get("Kernel_Name", "")
            for frag, label in KERNEL_LABEL.items():
                if frag in kname:
                    a = per.setdefault(label, [0, 0.0])
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    return {k: v[1] / v[0] for k, v in per.items()}


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
out = {"_workload": "c3", "_src_sha256": MAN.get("src_sha256"), "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1 --no-graph`, c3 workload, "
                "480 images. Counter unit KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); "
                "WRITE_SIZE as read. Bytes per launch (mean over the launches of the run)."}
for label in sorted(set(fetch) | set(write)):
    fb, wb = 2.0 * 1024.0 * fetch.get(label, 0.0), 1024.0 * write.get(label, 0.0)
    out[label] = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                  "fetch_raw_kib": fetch.get(label), "write_raw_kib": write.get(label)}
if len(out) > 1:
    for name in (f"{tag}_pmc_traffic.json", "pmc_traffic.json"):
        with open(os.path.join(DST, name), "w") as f:
            json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))


# ---- SQ / GRBM counters of the hot kernels: MFMA-pipe utilisation and where the wave cycles go -----------------------------
SQ = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
      "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"]
sq = {c: pmc("pmc_sq", c) for c in SQ}
labels = sorted(set().union(*[set(v) for v in sq.values()]))
if labels:
    N_SIMD, N_XCD = 256 * 4, 8
    res = {"_src_sha256": MAN.get("src_sha256"), "_note": "rocprofv3 --pmc (one pass, 7 SQ + 1 GRBM counter) of `bench.py --steps 3 --warmup 1 --no-graph`, c3 workload; per-launch means. "
                    "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): SQ_VALU_MFMA_BUSY_CYCLES is summed over "
                    "all SIMDs in shader cycles (it equals 32 x the kernel's v_mfma_f32_16x16x4_f32 count: 238.9 M for enc.conv12 = 7.47 M MFMAs), "
                    "GRBM_GUI_ACTIVE is summed over the 8 XCDs (1/8 of it x 1/2.4 GHz is the kernel's duration). wave-cycle split: WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + "
                    "ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, MI355X_MICROARCH.md)."}
    for lb in labels:
        v = {c: sq[c].get(lb) for c in SQ}
        d = dict(v)
        if v["GRBM_GUI_ACTIVE"]:
            d["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / N_XCD * N_SIMD)
        if v["SQ_WAVE_CYCLES"]:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                d["frac_" + c[3:].lower()] = v[c] / v["SQ_WAVE_CYCLES"]
        res[lb] = d
    with open(os.path.join(DST, f"{tag}_pmc_sq.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {kk: (round(vv, 4) if isinstance(vv, float) and vv < 10 else vv) for kk, vv in v.items()} for k, v in res.items() if not k.startswith("_")}, indent=1))


# ---- the opt-in split-precision conv12 kernels (extras) -----------------------------------------------------------------------
for name in (f"{tag}_split_bench_c3.json", f"{tag}_split_error_vs_float64.txt", f"{tag}_split_inst_mix.txt", f"{tag}_split_lds_conflicts.txt"):
    p = os.path.join(SRC, name)
    if fresh(p):
        shutil.copy(p, os.path.join(DST, name))
st = find("stats_split/**/*kernel_stats.csv")
if st:
    shutil.copy(st, os.path.join(DST, f"{tag}_split_kernel_stats.csv"))
SPLIT_LABEL = {"conv12_fwd_split_kernel": "enc.conv12.split", "conv12_wgrad_split_kernel": "enc.bwd.conv12.wgrad.split",
               "conv12_dgrad_split_kernel": "enc.bwd.conv12.dgrad.split"}
KERNEL_LABEL.clear()
KERNEL_LABEL.update(SPLIT_LABEL)
sq = {c: pmc("pmc_sq_split", c) for c in SQ}
labels = sorted(set().union(*[set(v) for v in sq.values()]))
if labels:
    res = {"_note": "as <tag>_pmc_sq.json, `bench.py ... --opt conv2_split=7`.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): "
                    "the share of SIMD cycles in which either matrix pipe (bf16 conv2, fp32 conv1 / conv1's gradient) is busy."}
    for lb in labels:
        v = {c: sq[c].get(lb) for c in SQ}
        d = dict(v)
        if v["GRBM_GUI_ACTIVE"]:
            d["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if v["SQ_WAVE_CYCLES"]:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                d["frac_" + c[3:].lower()] = v[c] / v["SQ_WAVE_CYCLES"]
        res[lb] = d
    with open(os.path.join(DST, f"{tag}_split_pmc_sq.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: {kk: (round(vv, 4) if isinstance(vv, float) and vv < 10 else vv) for kk, vv in v.items()} for k, v in res.items() if not k.startswith("_")}, indent=1))
