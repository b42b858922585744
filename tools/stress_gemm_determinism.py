"""Race screen for the GEMM loops: the same launch many times, every result compared bit for bit with the first (a k-loop that read
an LDS image before it was complete, or overwrote one still in use, shows up as a differing run)."""
import sys, torch
sys.path.insert(0, ".")
from snag_amd import ops
torch.manual_seed(0)
cases = [("l1 fwd (ping-pong)", (4096, 15000), (4096, 128), False, False), ("dl1 (ping-pong)", (15000, 4096), (15000, 128), False, False),
         ("text l1", (768, 15000), (768, 128), False, False), ("tall128 N=300", (7000, 7000), (7000, 300), True, False),
         ("KV 64x128", (45000, 128), (256, 128), True, True), ("dX", (45000, 256), (256, 128), True, False),
         ("dW pair shape", (15000, 128), (15000, 128), False, False), ("ragged", (1501, 4099), (1501, 130), False, False)]
bad = 0
for name, sa, sb, akc, bkc in cases:
    a, b = torch.randn(*sa, device="cuda"), torch.randn(*sb, device="cuda")
    ref = ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc).clone()
    n_diff = 0
    for i in range(150):
        c = ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc)
        if not torch.equal(c, ref):
            n_diff += 1
    print(f"{name}: {n_diff} of 150 runs differ", flush=True)
    bad += n_diff
ref64 = None
print("OK" if bad == 0 else "RACE SUSPECTED")
sys.exit(1 if bad else 0)
