"""Run one GEMM shape a few times (for rocprofv3 kernel traces): python tools/one_shape.py KROWS MCOLS NCOLS (row-contiguous A^T [K,M], B [K,N])."""
import sys, torch
sys.path.insert(0, ".")
from snag_amd import ops
K, M, N = (int(x) for x in sys.argv[1:4])
a = torch.randn(K, M, device="cuda"); b = torch.randn(K, N, device="cuda"); out = torch.empty(M, N, device="cuda")
for _ in range(30): ops.gemm(a, b, a_kcontig=False, b_kcontig=False, out=out)
torch.cuda.synchronize()
