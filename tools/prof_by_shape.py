#!/usr/bin/env python3
"""Per-(kernel, grid) table from rocprofv3 CSV output: durations from *_kernel_trace.csv and, when present, PMC counters from
*_counter_collection.csv (FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950)."""
import collections
import csv
import glob
import sys


def load(pattern):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def short(name):
    name = name.replace("void snag::", "").replace("snag::", "")
    cut = name.find("(")
    return (name[:cut] if cut > 0 else name)[:78]


def split_bimodal(vals):
    """Two launches of one kernel with the same grid but different shapes (the visual and the text projection both run as 512
    stream-K workgroups) share a (kernel, grid) key: split such a group at the geometric mean of its extremes."""
    if len(vals) < 4 or min(vals) <= 0 or max(vals) / min(vals) < 2.5:
        return None
    cut = (min(vals) * max(vals)) ** 0.5
    hi, lo = [v for v in vals if v >= cut], [v for v in vals if v < cut]
    return (hi, lo) if hi and lo else None


def main():
    out = open(sys.argv[-1], "w") if sys.argv[-1].endswith(".txt") else None
    dirs = [a for a in sys.argv[1:] if not a.endswith(".txt")]
    dur = collections.defaultdict(list)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for r in load(d + "/**/*kernel_trace.csv"):
            if "Grid_Size_X" in r:
                grid = str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
            else:
                grid = r.get("Grid_Size", "?")
            dur[(short(r["Kernel_Name"]), grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for r in load(d + "/**/*counter_collection.csv"):
            ctr[(short(r["Kernel_Name"]), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in [k for k in dur if "gemm" in k[0]]:
        sp = split_bimodal(dur[k])
        if sp:
            del dur[k]
            dur[(k[0][:70] + " [long]", k[1])], dur[(k[0][:70] + " [short]", k[1])] = sp
    for k in [k for k in ctr if "gemm" in k[0]]:
        sps = {name: split_bimodal(vals) for name, vals in ctr[k].items()}
        if sps and any(sps.values()):          # a counter that does not separate the two shapes is quoted as their mean on both rows
            cs = ctr.pop(k)
            for tag, idx in ((" [long]", 0), (" [short]", 1)):
                for name in cs:
                    ctr[(k[0][:70] + tag, k[1])][name] = sps[name][idx] if sps[name] else cs[name]
    lines = []
    if dur:
        tot = sum(sum(v) for v in dur.values())
        lines.append(f"{'kernel':80s} {'grid':>9s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'pct':>6s}")
        for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:45]:
            lines.append(f"{k[0]:80s} {k[1]:>9s} {len(v):6d} {sum(v) / len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {sum(v) / 1e3:9.3f} {100 * sum(v) / tot:6.2f}")
    if ctr:
        lines.append("")
        lines.append(f"{'kernel':80s} {'grid':>9s} {'n':>4s}  counters (per launch; FETCH_SIZE/WRITE_SIZE in MB, FETCH corrected x2)")
        for k, cs in sorted(ctr.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values()))[:45]:
            parts = []
            for name, vals in sorted(cs.items()):
                avg = sum(vals) / len(vals)
                if name == "FETCH_SIZE":
                    parts.append(f"FETCH {2 * avg / 1024:9.2f} MB")
                elif name == "WRITE_SIZE":
                    parts.append(f"WRITE {avg / 1024:9.2f} MB")
                else:
                    parts.append(f"{name} {avg:.4g}")
            lines.append(f"{k[0]:80s} {k[1]:>9s} {len(next(iter(cs.values()))):4d}  " + "  ".join(parts))
    text = "\n".join(lines)
    print(text)
    if out:
        out.write(text + "\n")


main()
