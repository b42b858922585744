import sys, torch
sys.path.insert(0, '.')
from snag_amd import ops
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for k in (300, 1200):
    z = torch.randn(7000, k, device='cuda')
    out = torch.empty(7000, 7000, device='cuda')
    print(k, "gemm", t(lambda: ops.gemm(z, z, alpha=10.0, out=out)), "gram", t(lambda: ops.gram(z, alpha=10.0, out=out)))
for k in (300, 1200):
    z = torch.randn(7000, k, device='cuda')
    outp = torch.empty(7000, 7008, device='cuda')[:, :7000]
    print(k, "padded ld 7008: gemm", t(lambda: ops.gemm(z, z, alpha=10.0, out=outp)), "gram", t(lambda: ops.gram(z, alpha=10.0, out=outp)))
