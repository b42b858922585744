#!/bin/bash
# shader clock held during the big GEMM: GRBM_GUI_ACTIVE (cycles, per XCD) / kernel duration, same rocprofv3 run
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
O=gpurun_out/clk; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/t -- python3 tools/one_shape.py 4096 15000 128 > $O/log 2>&1
python3 - <<'PY'
import csv, glob, collections
O = "gpurun_out/clk/t"
dur = collections.defaultdict(list); cyc = collections.defaultdict(list)
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Kernel_Name"][:50], r.get("Dispatch_Id") or r.get("Correlation_Id"))] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(rows[0].keys())
    acc = collections.defaultdict(float)
    for r in rows:
        acc[(r["Kernel_Name"][:50], r.get("Dispatch_Id") or r.get("Correlation_Id"))] += float(r["Counter_Value"])
    for k, v in list(acc.items())[-12:]:
        d = dur.get(k)
        print(k[0], "cycles/XCD %.0f" % (v / 8), "dur_us", d, "GHz %.3f" % (v / 8 / d / 1e3) if d else "")
PY
rm -rf $O/t
