"""The visual projection [15000 x 4096] x [4096 x 128] in the four operand-layout combinations (standalone loop): tools/probe_layouts.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import ops
def timed(fn, n=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
N, fd, D = 15000, 4096, 128
x = torch.randn(N, fd, device="cuda"); ldt = (N + 31) // 32 * 32
xt = torch.zeros(fd, ldt, device="cuda")[:, :N]; xt.copy_(x.t())
w = torch.randn(D, fd, device="cuda"); wt = w.t().contiguous(); out = torch.empty(N, D, device="cuda")
big = torch.empty(300_000_000, device="cuda")          # 1.2 GB: flushed through between launches so that A comes from HBM, as inside the step
def flush(): big.add_(1.0)
for name, a, akc, b, bkc in (("A kc   B kc  ", x, True, w, True), ("A kc   B rowc", x, True, wt, False), ("A rowc B kc  ", xt, False, w, True), ("A rowc B rowc", xt, False, wt, False)):
    t = timed(lambda: ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out, act="relu"))
    def cold():
        flush(); ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out, act="relu")
    tc = timed(cold) - timed(flush)
    print(f"{name}: back-to-back {t:6.1f} us   behind a 1.2 GB sweep {tc:6.1f} us", flush=True)
for layout in (1, 2):
    xa = ops.tile_a(x, kcontig=True, layout=layout)
    for name, b, bkc in (("B kc  ", w, True), ("B rowc", wt, False)):
        t = timed(lambda: ops.gemm(xa, b, a_tiled=(N, fd), a_layout=layout, b_kcontig=bkc, out=out, act="relu"))
        def cold():
            flush(); ops.gemm(xa, b, a_tiled=(N, fd), a_layout=layout, b_kcontig=bkc, out=out, act="relu")
        tc = timed(cold) - timed(flush)
        print(f"A tiled{layout} {name}: back-to-back {t:6.1f} us   behind a 1.2 GB sweep {tc:6.1f} us", flush=True)
