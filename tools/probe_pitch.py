"""How much do row pitches that are not whole 128-byte lines cost the MMEA-shaped GEMMs?  (A [T,300] / out [T,900] vs padded 320 / 928)"""
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
T = 158376
for (K, N) in ((300, 900), (300, 300), (400, 300), (300, 400)):
    w = torch.randn(N, K, device='cuda')
    for pa in (K, (K + 31) // 32 * 32):
        for pc in (N, (N + 31) // 32 * 32):
            a = torch.randn(T, pa, device='cuda')[:, :K]
            c = torch.empty(T, pc, device='cuda')[:, :N]
            wp = torch.randn(N, pa, device='cuda')[:, :K]
            print(f"K={K} N={N} lda={pa} ldc={pc}: {t(lambda: ops.gemm(a, wp, out=c)):.1f} us")
