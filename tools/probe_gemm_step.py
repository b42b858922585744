"""Time the step's large GEMM shapes through snag_gemm_f32 on one device (environment switches SNAG_GEMM_* select variants)."""
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


def main():
    N, D = 15000, 128
    ldt = (N + 31) // 32 * 32
    out = {}
    for fd in (4096, 768):
        xt = torch.randn(fd, ldt, device="cuda")[:, :N]            # transposed feature table [fd, N], padded pitch
        x = torch.randn(N, fd, device="cuda")
        w = torch.randn(fd, D, device="cuda")
        dmid = torch.randn(N, D, device="cuda")
        mid = torch.empty(N, D, device="cuda")
        dw = torch.empty(fd, D, device="cuda")
        t_f = timed(lambda: ops.gemm(xt, w, a_kcontig=False, b_kcontig=False, out=mid, act="relu"))
        t_b = timed(lambda: ops.gemm(x, dmid, a_kcontig=False, b_kcontig=False, out=dw))
        fl = 2.0 * N * fd * D
        out[fd] = (t_f, t_b)
        print(f"fd={fd}: l1 fwd {t_f:.1f} us ({fl / t_f / 1e6:.1f} TF, {fl / t_f / 1e6 / 157.3:.3f}), dl1 {t_b:.1f} us ({fl / t_b / 1e6:.1f} TF, {fl / t_b / 1e6 / 157.3:.3f})", flush=True)
    # Mformer shapes
    x = torch.randn(45000, D, device="cuda")
    wkv = torch.randn(2 * D, D, device="cuda")
    o = torch.empty(45000, 2 * D, device="cuda")
    print(f"KV [45000x256x128]: {timed(lambda: ops.gemm(x, wkv, out=o)):.1f} us", flush=True)
    dkv = torch.randn(45000, 2 * D, device="cuda")
    dx = torch.empty(45000, D, device="cuda")
    print(f"dX = dkv Wkv [45000x128x256]: {timed(lambda: ops.gemm(dkv, wkv, b_kcontig=False, out=dx)):.1f} us", flush=True)
    dwkv = torch.empty(2 * D, D, device="cuda")
    print(f"dWkv [256x128x45000]: {timed(lambda: ops.gemm(dkv, x, a_kcontig=False, b_kcontig=False, out=dwkv)):.1f} us", flush=True)


if __name__ == "__main__":
    main()
