#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small per-kernel table (for profiles/)."""
import csv
import glob
import sys

d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(r)
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
lines = [f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}"]
for r in rows[:40]:
    name = r["Name"]
    if len(name) > 100:
        name = name[:97] + "..."
    lines.append(f"{name:100s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.2f} "
                 f"{float(r['MinNs']) / 1e3:9.2f} {float(r['MaxNs']) / 1e3:9.2f} {100 * float(r['TotalDurationNs']) / tot:6.2f}")
text = "\n".join(lines)
print(text)
if out:
    open(out, "w").write(text + "\n")
