#!/usr/bin/env python3
"""One steady-state training step as an ordered launch list from a rocprofv3 --kernel-trace CSV directory:
start offset, duration and the idle gap in front of every launch.  Steps are delimited by the sampler launch.
    python3 tools/prof_timeline.py <trace_dir> [out.txt] [marker_kernel_substring]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    marker = sys.argv[3] if len(sys.argv) > 3 else "step_prologue_kernel"
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(marks) < 4 and len(sys.argv) <= 3:
        # (round 6: the headline step has no prologue launch any more) a step = the launches behind one AdamW launch up to and including the next
        marks = [i + 1 for i, r in enumerate(rows[:-1]) if "adamw_kernel" in r["Kernel_Name"]]
    if len(marks) < 4:
        sys.exit(f"fewer than 4 '{marker}' launches in the trace")
    # the median-length step among the last ones
    spans = [(int(rows[marks[i + 1]]["Start_Timestamp"]) - int(rows[marks[i]]["Start_Timestamp"]), i) for i in range(len(marks) // 2, len(marks) - 1)]
    spans.sort()
    span, i = spans[len(spans) // 2]
    seg = rows[marks[i]:marks[i + 1]]
    t0 = int(seg[0]["Start_Timestamp"])
    busy = 0
    prev_end = t0
    print(f"step of {span / 1e3:.1f} us, {len(seg)} launches (median of {len(spans)} steady-state steps)", file=out)
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  {'grid':>8} {'vgpr':>4}  kernel", file=out)
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void snag::", "").replace("snag::", "")
        cut = name.find("(")
        name = (name[:cut] if cut > 0 else name)[:110]
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {(s - prev_end) / 1e3:7.2f}  {grid:8d} {r.get('VGPR_Count', ''):>4}  {name}", file=out)
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"busy {busy / 1e3:.1f} us of {span / 1e3:.1f} us ({100.0 * busy / span:.1f} %)", file=out)


if __name__ == "__main__":
    main()
