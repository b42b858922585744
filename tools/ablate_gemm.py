"""Ablation of the dominant GEMM's k-loop (results are wrong in the ablated modes; timing only):
   SNAG_GEMM_DBG=0 real kernel; 1 = no global loads after the first two tiles; 2 = additionally no per-k-tile barrier."""
import os, sys, subprocess
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, ".")
    from snag_amd import ops
    a = torch.randn(4096, 15008, device="cuda")[:, :15000]; b = torch.randn(4096, 128, device="cuda"); out = torch.empty(15000, 128, device="cuda")
    for _ in range(5): ops.gemm(a, b, a_kcontig=False, b_kcontig=False, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.gemm(a, b, a_kcontig=False, b_kcontig=False, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{us:.1f} us incl. fix-up  ({2 * 15000 * 4096 * 128 / us / 1e6:.1f} TFLOP/s)")
else:
    for dbg in ("0", "1", "2", "0", "1", "2"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SNAG_GEMM_DBG=dbg), capture_output=True, text=True)
        print("dbg", dbg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
