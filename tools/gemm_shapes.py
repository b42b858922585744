"""Aggregate `SNAG_GEMM_LOG=1` stderr lines by shape: python tools/gemm_shapes.py < log"""
import sys, collections
c = collections.Counter(l.strip() for l in sys.stdin if l.startswith("snag_gemm "))
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    f = dict(t.split("=") for t in k.split()[1:])
    gf = 2.0 * int(f["M"]) * int(f["N"]) * int(f["K"]) / 1e9
    print(f"{v:6d}  {gf:8.2f} GF  {k}")
