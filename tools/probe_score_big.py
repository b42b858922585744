"""Score kernel on a table far larger than the 256 MB Infinity Cache (config-5 shape, fp32): true HBM-bound gather."""
import sys, torch
sys.path.insert(0, '.')
from snag_amd import _lib
dev = 'cuda'
g = torch.Generator().manual_seed(1)
N, R, D, K = 1_000_000, 1000, 256, 128
ent = torch.empty(N, D, device=dev).normal_(); rel = torch.randn(R, D, generator=g).to(dev)
for B in (1024, 8192, 32768):
    h = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(dev); r = torch.randint(0, R, (B,), generator=g, dtype=torch.int32).to(dev)
    t = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(dev); neg = torch.randint(0, N, (B, K), generator=g, dtype=torch.int32).to(dev)
    pos = torch.empty(B, device=dev); ng = torch.empty(B, K, device=dev); lp = torch.empty(B, device=dev)
    st = _lib.stream()
    fn = lambda: _lib.call("snag_score_margin_fwd", ent.data_ptr(), D, N, rel.data_ptr(), R, D, h.data_ptr(), r.data_ptr(), t.data_ptr(), neg.data_ptr(), 4, B, K, 12.0, pos.data_ptr(), ng.data_ptr(), lp.data_ptr(), st)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    by = B * ((K + 3) * D * 4 + (K + 3) * 4 + (K + 1) * 4)
    print(f"N=1M D=256 K=128 fp32 B={B}: {us:.1f} us/launch, algorithmic {by / us / 1e6:.2f} TB/s = {by / us / 1e6 / 8 * 100:.1f}% of 8 TB/s, {B / us:.2f} Mtriples/s")

eb = ent.to(torch.bfloat16); rb = rel.to(torch.bfloat16)
for B in (1024, 8192, 32768):
    h = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(dev); r = torch.randint(0, R, (B,), generator=g, dtype=torch.int32).to(dev)
    t = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(dev); neg = torch.randint(0, N, (B, K), generator=g, dtype=torch.int32).to(dev)
    pos = torch.empty(B, device=dev); ng = torch.empty(B, K, device=dev); lp = torch.empty(B, device=dev)
    st = _lib.stream()
    fn = lambda: _lib.call("snag_score_margin_fwd_bf16", eb.data_ptr(), D, N, rb.data_ptr(), R, D, h.data_ptr(), r.data_ptr(), t.data_ptr(), neg.data_ptr(), 4, B, K, 12.0, pos.data_ptr(), ng.data_ptr(), lp.data_ptr(), st)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    by = B * ((K + 3) * D * 2 + (K + 3) * 4 + (K + 1) * 4)
    print(f"N=1M D=256 K=128 bf16 B={B}: {us:.1f} us/launch, algorithmic {by / us / 1e6:.2f} TB/s = {by / us / 1e6 / 8 * 100:.1f}% of 8 TB/s, {B / us:.2f} Mtriples/s")
