"""Run the projection GEMM in the layouts the engine streams (A [M, K] and B [N, K] both K-contiguous) a few times, for rocprofv3:
python tools/one_shape_kc.py M K N"""
import sys, torch
sys.path.insert(0, ".")
from snag_amd import ops
M, K, N = (int(x) for x in sys.argv[1:4])
a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
for _ in range(30): ops.gemm(a, b, out=out, act="relu")
torch.cuda.synchronize()
