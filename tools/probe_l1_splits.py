"""Is the long-K eight-wave launch paced per CU (time ~ k-tiles per workgroup) or by the chip (time ~ total bytes)?  The visual first projection
[15000, 4096] x [128, 4096]^T alone with 2 / 3 / 4 K splits (118 / 177 / 236 workgroups of 128 / 86 / 64 k-tiles), cold operands:
    python tools/probe_l1_splits.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import _lib
from snag_amd._lib import call, ptr, stream
N, D, fd = 15000, 128, 4096
lib = _lib.load()
big = torch.empty(300_000_000, device="cuda")
x = torch.randn(N, fd, device="cuda"); w = torch.randn(D, fd, device="cuda"); out = torch.empty(N, D, device="cuda")


def run(M, s, reps=12):
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.a_kcontig, d.b_kcontig, d.act = M, D, fd, 1, 1, 1
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = ptr(x), fd, ptr(w), fd, ptr(out), D
    ws = torch.empty(max(64, (s + 1) * M * D), device="cuda")
    d.alpha, d.workspace, d.workspace_floats, d.splits, d.defer_reduce = 1.0, ptr(ws), ws.numel(), s, 1
    ts = []
    for _ in range(reps):
        big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("snag_gemm_f32", ctypes.byref(d), stream())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


for rnd in range(2):
    for M, s in ((15000, 2), (15000, 3), (15000, 4), (7500, 4), (7500, 8), (15000, 1)):
        wg = -(-M // 256) * s
        print(f"M {M} splits {s}: {wg:3d} workgroups x {-(-256 // s):3d} k-tiles: {run(M, s):6.1f} us", flush=True)
