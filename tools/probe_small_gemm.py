"""The 15000 x 128 x 128 GEMM of the read-out chain, standalone (back-to-back launches; SNAG_GEMM_DBG ablations with an ABLATE build)."""
import sys, torch
sys.path.insert(0, ".")
from snag_amd import ops
x, w, o = torch.randn(15000, 128, device="cuda"), torch.randn(128, 128, device="cuda"), torch.empty(15000, 128, device="cuda")
for _ in range(10): ops.gemm(x, w, out=o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): ops.gemm(x, w, out=o)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) * 10:.2f} us per launch")
