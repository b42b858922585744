"""A/B on one device: XCD-aware tile mapping on (default) vs off (SNAG_GEMM_XCD=2)."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ".")
    from snag_amd import ops
    N, T = 15000, 158376
    shapes = {"l1_fwd_TT": ((4096, 15008), (4096, 128), False, False), "l1_bwd": ((N, 128), (N, 4096), False, False), "qkv": ((45000, 128), (384, 128), True, True),
              "dx_qkv": ((45000, 384), (384, 128), True, False), "wo": ((45000, 128), (128, 128), True, True),
              "S300": ((7000, 300), (7000, 300), True, True), "dZ300": ((7000, 7000), (7000, 300), True, False),
              "mmea_qkv": ((T, 300), (900, 300), True, True), "mmea_out": ((T, 300), (300, 300), True, True), "mmea_dx": ((T, 900), (900, 300), True, False),
              "proj_img": ((39594, 2048), (300, 2048), True, True), "sim": ((10500, 1200), (10500, 1200), True, True)}
    res = {}
    for name, (sa, sb, akc, bkc) in shapes.items():
        a = torch.randn(*sa, device="cuda"); b = torch.randn(*sb, device="cuda")
        if name == "l1_fwd_TT": a = a[:, :N]
        M = a.shape[0] if akc else a.shape[1]; Nn = sb[0] if bkc else sb[1]; K = a.shape[1] if akc else a.shape[0]
        out = torch.empty(M, Nn, device="cuda")
        for _ in range(3): ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out)
        e1.record(); torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
    z = torch.randn(7000, 300, device="cuda"); o = torch.empty(7000, 7000, device="cuda")
    for _ in range(3): ops.gram(z, out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gram(z, out=o)
    e1.record(); torch.cuda.synchronize()
    res["gram300"] = round(e0.elapsed_time(e1) * 1e3 / 20, 1)
    print(json.dumps(res))
else:
    for mode in ("2", "1", "2", "1"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, SNAG_GEMM_XCD=mode), capture_output=True, text=True)
        print("xcd", "off" if mode == "2" else "on ", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])
