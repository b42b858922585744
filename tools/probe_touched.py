"""configs[4] training scorer (snag_score_margin_fwd_bwd_touched_bf16 on the 1M x 256 bf16 table, B = 1024, K = 128, uniform indices): HIP-event time
of the entry, 30 back-to-back calls.  Run under several libraries (SNAG_HIP_LIB) to see where its launches' time goes (tools/ab_touched.sh)."""
import os, sys, torch
sys.path.insert(0, '.')
from snag_amd import ops
dev = 'cuda'
g = torch.Generator().manual_seed(5)
N, R, D, K = 1_000_000, 1000, 256, 128
e16 = torch.empty(N, D, device=dev).normal_().to(torch.bfloat16)
r16 = torch.randn(R, D, generator=g).to(dev).to(torch.bfloat16)
d_ent, d_rel = torch.zeros(N, D, device=dev), torch.zeros(R, D, device=dev)
for B in (1024, 8192):
    idx = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).to(dev) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
    run = lambda: ops.score_margin_fwd_bwd_touched(e16, r16, *idx, 12.0, d_ent=d_ent, d_rel=d_rel)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    by = B * ((K + 3) * D * 2 + (K + 3) * 4 + (K + 1) * 4) + B * (K + 3) * D * 4 * 2
    print(f"{os.environ.get('SNAG_HIP_LIB', 'in-tree').split('_')[-1]:>14s} B={B}: {us:7.1f} us  {by / us / 1e3:7.0f} GB/s  frac {by / us / 1e3 / 8000:.3f}")
