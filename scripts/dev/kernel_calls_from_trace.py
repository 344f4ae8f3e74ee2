"""Per-kernel duration histogram of the graph replays in a rocprofv3 kernel trace: for kernels whose name contains one of the given
substrings, the durations of the calls of ONE step in launch order (median over steps).  usage: kernel_calls_from_trace.py <trace.csv> <per_step_launches> name..."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
names = sys.argv[2:]
for nm in names:
    sel = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if nm in r["Kernel_Name"])
    durs = [d for _, d in sel]
    # group: find the period by autocorrelation over small candidates
    best = None
    for per in range(1, 13):
        if len(durs) < 4 * per:
            continue
        tail = durs[-(len(durs) // per // 2) * per:]
        cols = [tail[i::per] for i in range(per)]
        spread = sum(statistics.pstdev(c) for c in cols) / per
        if best is None or spread < best[0] * 0.8:
            best = (spread, per, [statistics.median(c) / 1e3 for c in cols])
    print(nm, "calls", len(durs), "period", best[1], "us per call in order:", " ".join(f"{x:.1f}" for x in best[2]))
