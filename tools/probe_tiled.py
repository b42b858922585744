"""Plain vs tiled A for the step's long-K projection shapes (standalone loop): tools/probe_tiled.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import ops  # noqa: E402


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


N, D = 15000, 128
ldt = (N + 31) // 32 * 32
for fd in (4096, 768):
    x = torch.randn(N, fd, device="cuda")
    xt = torch.zeros(fd, ldt, device="cuda")[:, :N]
    xt.copy_(x.t())
    w_t = torch.randn(fd, D, device="cuda")          # [K, N] row-contiguous (today's storage)
    w = w_t.t().contiguous()                          # [N, K] K-contiguous (nn.Linear layout)
    dmid = torch.randn(N, D, device="cuda")
    dmid_t = dmid.t().contiguous()
    mid = torch.empty(N, D, device="cuda")
    dw = torch.empty(fd, D, device="cuda")
    dw_n = torch.empty(D, fd, device="cuda")
    x_tf = ops.tile_a(x, kcontig=True)               # forward operand: rows = entities, k = features
    x_tb = ops.tile_a(x, kcontig=False)              # backward operand: rows = features, k = entities
    res = {}
    res["fwd plain (A rowc, B rowc)"] = timed(lambda: ops.gemm(xt, w_t, a_kcontig=False, b_kcontig=False, out=mid, act="relu"))
    res["fwd tiled A, B rowc"] = timed(lambda: ops.gemm(x_tf, w_t, a_tiled=(N, fd), b_kcontig=False, out=mid, act="relu"))
    res["fwd tiled A, B kc"] = timed(lambda: ops.gemm(x_tf, w, a_tiled=(N, fd), b_kcontig=True, out=mid, act="relu"))
    res["bwd plain (A rowc, B rowc)"] = timed(lambda: ops.gemm(x, dmid, a_kcontig=False, b_kcontig=False, out=dw))
    res["bwd tiled A, B rowc"] = timed(lambda: ops.gemm(x_tb, dmid, a_tiled=(fd, N), b_kcontig=False, out=dw))
    res["bwd tiled A, B kc, C^T"] = timed(lambda: ops.gemm(x_tb, dmid_t, a_tiled=(fd, N), b_kcontig=True, out=dw_n, c_transposed=True))
    for k, v in res.items():
        print(f"fd={fd}: {k:32s} {v:7.1f} us  ({2.0 * N * fd * D / v / 1e6:6.1f} TF)", flush=True)
    w_img = ops.presplit_b(w_t, kcontig=False)
    w_img_kc = ops.presplit_b(w, kcontig=True)
    print(f"fd={fd}: presplit_b                       {timed(lambda: ops.presplit_b(w_t, kcontig=False, out=w_img)):7.1f} us", flush=True)
    for k, fn in (("fwd plain A, B image", lambda: ops.gemm(xt, w_t, a_kcontig=False, b_kcontig=False, out=mid, act="relu", b_image=w_img)),
                  ("fwd tiled A, B image", lambda: ops.gemm(x_tf, w_t, a_tiled=(N, fd), b_kcontig=False, out=mid, act="relu", b_image=w_img))):
        v = timed(fn)
        print(f"fd={fd}: {k:32s} {v:7.1f} us  ({2.0 * N * fd * D / v / 1e6:6.1f} TF)", flush=True)
