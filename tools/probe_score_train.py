"""Time the scorer's training forms on one device: fused one-launch (snag_score_margin_fwd_bwd) vs forward + backward + reduce.
(the waves-per-triple of the fused kernel are chosen by the library from the batch size)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import ops  # noqa: E402
from snag_amd._lib import call, ptr, stream  # noqa: E402


def timed(fn, n=50, warm=5):
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
    dev = "cuda"
    for (N, D, K, B) in ((15000, 128, 32, 1024), (15000, 128, 32, 8192), (15000, 128, 128, 1024), (1_000_000, 256, 128, 1024)):
        g = torch.Generator().manual_seed(1)
        ent = torch.randn(N, D, generator=g).to(dev)
        rel = torch.randn(558, D, generator=g).to(dev)
        h, r, t = (torch.randint(0, hi, (B,), generator=g, dtype=torch.int32).to(dev) for hi in (N, 558, N))
        neg = torch.randint(0, N, (B, K), generator=g, dtype=torch.int32).to(dev)
        d_ent, d_rel = torch.zeros_like(ent), torch.zeros_like(rel)
        pos, ng, lp, loss = torch.empty(B, device=dev), torch.empty(B, K, device=dev), torch.empty(B, device=dev), torch.empty((), device=dev)
        tick = ops._ticket(torch.device(dev, torch.cuda.current_device()))
        args = (ptr(ent), D, N, ptr(rel), 558, D, ptr(h), ptr(r), ptr(t), ptr(neg), 4, B, K, 12.0)
        fused = lambda: call("snag_score_margin_fwd_bwd", *args, 1.0, ptr(pos), ptr(ng), ptr(lp), ptr(loss), ptr(tick), ptr(d_ent), D, ptr(d_rel), stream())
        fused_noloss = lambda: call("snag_score_margin_fwd_bwd", *args, 1.0, ptr(pos), ptr(ng), ptr(lp), None, None, ptr(d_ent), D, ptr(d_rel), stream())
        fwd = lambda: call("snag_score_margin_fwd", *args, ptr(pos), ptr(ng), ptr(lp), stream())
        bwd = lambda: call("snag_score_margin_bwd", *args, 1.0, None, None, ptr(d_ent), D, ptr(d_rel), stream())
        red = lambda: call("snag_reduce_sum", ptr(lp), B, ptr(loss), stream())
        from snag_amd import _lib
        nb = _lib.load().snag_score_seg_work_bytes(N, 558, B, K, D)
        work = torch.zeros(nb, device=dev, dtype=torch.uint8)
        seg = lambda: call("snag_score_margin_fwd_bwd_seg", *args, 1.0, ptr(pos), ptr(ng), ptr(lp), ptr(loss), ptr(tick), ptr(d_ent), D, ptr(d_rel), 1, ptr(work), nb, stream())
        t_seg = timed(seg)
        by = B * ((K + 3) * D * 4 * 3 + (K + 3) * 4 + (K + 1) * 4)
        print(f"    segmented (no float atomics, overwrite): {t_seg:.2f} us = {by / t_seg / 8e6:.3f} of 8 TB/s on the same algorithmic bytes", flush=True)
        tf = timed(fused)
        print(f"N={N} D={D} K={K} B={B}: fused {tf:.2f} us ({by / tf / 1e3:.0f} GB/s algorithmic = "
              f"{by / tf / 8e6:.3f} of 8 TB/s; atomic floor {B * (K + 3) * D * 4 / 1.3e6:.1f} us), fused without loss sum {timed(fused_noloss):.2f}, "
              f"fwd {timed(fwd):.2f} + bwd {timed(bwd):.2f} + reduce {timed(red):.2f}", flush=True)


if __name__ == "__main__":
    main()
