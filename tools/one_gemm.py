import sys, torch
sys.path.insert(0, ".")
from snag_amd import ops
N = 15000
which = sys.argv[1] if len(sys.argv) > 1 else "l1_fwd"
shapes = {"l1_fwd": ((N, 4096), (128, 4096), True, True), "l1_bwd": ((N, 128), (N, 4096), False, False),
          "qkv": ((45000, 128), (384, 128), True, True)}
sa, sb, akc, bkc = shapes[which]
a = torch.randn(*sa, device="cuda"); b = torch.randn(*sb, device="cuda")
M = sa[0] if akc else sa[1]; Nn = sb[0] if bkc else sb[1]
out = torch.empty(M, Nn, device="cuda")
for _ in range(12): ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out)
torch.cuda.synchronize()
