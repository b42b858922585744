"""MKGC filtered tail prediction at DB15K scale: all-entity scoring of B test triples (difference-form scorer) + count-less-than ranks."""
import sys, torch
sys.path.insert(0, '.')
from snag_amd import evalsim
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator().manual_seed(0)
N, R, D = 15000, 558, 128
ent = torch.randn(N, D, generator=g).cuda(); rel = torch.randn(R, D, generator=g).cuda()
for B in (1024, 4096, 16384):
    h = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).cuda(); r = torch.randint(0, R, (B,), generator=g, dtype=torch.int32).cuda()
    tt = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).cuda()
    filt = (torch.rand(B, N, generator=g) < 0.001).to(torch.uint8).cuda()
    ms = t(lambda: evalsim.link_prediction(ent, rel, h, r, tt, filt))
    ms2 = t(lambda: evalsim.link_prediction(ent, rel, h, r, tt, filt, method="gemm"))
    print(f"B={B}: exact {ms:.2f} ms ({B * N / ms / 1e6:.1f} G scores/s)   gemm {ms2:.2f} ms ({B * N / ms2 / 1e6:.1f} G scores/s)")
