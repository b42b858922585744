"""Per-step view of a rocprofv3 kernel trace of bench.py: step span and the big GEMM's duration, step by step (start-up transient)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps, cur = [], None
for s, e, k in rows:
    if "step_prologue" in k:
        cur = {"t0": s, "gemm": [], "busy": 0, "n": 0}
        steps.append(cur)
    if cur is not None:
        cur["t1"] = e; cur["busy"] += e - s; cur["n"] += 1
        if "gemm_streamk" in k: cur["gemm"].append((e - s) / 1e3)
for i, st in enumerate(steps[:-1]):
    nxt = steps[i + 1]["t0"]
    if i < 40 or i % 10 == 0:
        print(i, "period %.1f us" % ((nxt - st["t0"]) / 1e3), "busy %.1f" % (st["busy"] / 1e3), "launches", st["n"], "streamk", ["%.1f" % g for g in st["gemm"]])
# per-launch comparison: an early step vs a late one
def launches(i):
    t0, t1 = steps[i]["t0"], steps[i + 1]["t0"]
    return [((e - s) / 1e3, k.split("(")[0][:70]) for s, e, k in rows if t0 <= s < t1]
a, b = launches(5), launches(60)
print("launch-by-launch, step 5 vs step 60")
for (da, ka), (db, kb) in zip(a, b):
    print("%7.1f %7.1f  %+6.1f%%  %s" % (da, db, 100 * (da - db) / db, ka))
