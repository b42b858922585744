"""One product through the B-resident streaming form (SNAG_GEMM_STREAM=1 SNAG_GEMM_STREAM_SINGLE=1) against fp64: tools/dbg_stream.py M N K [b_kcontig]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
bkc = len(sys.argv) < 5 or sys.argv[4] == "1"
g = torch.Generator().manual_seed(1)
a = torch.randn(M, K, generator=g).cuda(); b = torch.randn(N, K, generator=g).cuda() if bkc else torch.randn(K, N, generator=g).cuda()
out = torch.zeros(M, N, device="cuda")
ops.gemm(a, b, a_kcontig=True, b_kcontig=bkc, out=out)
torch.cuda.synchronize()
ref = a.double() @ (b.double().t() if bkc else b.double())
err = (out.double() - ref).abs()
bad = err > 1e-3
print("bad", int(bad.sum()), "of", bad.numel())
if bad.any():
    idx = bad.nonzero()
    print("rows", idx[:, 0].min().item(), idx[:, 0].max().item(), "cols", sorted(set((idx[:, 1] // 32).tolist()))[:20])
    r = idx[:, 0] % 32
    print("row-in-block histogram", torch.bincount(r, minlength=32).tolist())
    print("col-in-32 histogram", torch.bincount(idx[:, 1] % 32, minlength=32).tolist())
