"""Time snag_gemm_f32 on the MMEA step's shapes (N = 300-wide outputs) per forced tile configuration (SNAG_GEMM_TILE: 0 auto, 1 64x128,
2 128x128, 3 128x160); each child also checks the result against torch.matmul in fp64."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ".")
    from snag_amd import ops
    T, N = 158376, 39594
    shapes = {"proj_img": ((N, 2048), (300, 2048), True, True), "proj_rel": ((N, 1000), (300, 1000), True, True),
              "qkv": ((T, 300), (900, 300), True, True), "out": ((T, 300), (300, 300), True, True), "ffn_up": ((T, 300), (400, 300), True, True),
              "ffn_dn": ((T, 400), (300, 400), True, True), "dx_qkv": ((T, 900), (900, 300), True, False), "dx_up": ((T, 400), (400, 300), True, False),
              "dx_dn": ((T, 300), (300, 400), True, False), "dZ300": ((7000, 7000), (7000, 300), True, False), "dZ1200": ((7000, 7000), (7000, 1200), True, False),
              "S300": ((7000, 300), (7000, 300), True, True), "dW_out": ((T, 300), (T, 300), False, False), "dW_up": ((T, 300), (T, 400), False, False)}
    res = {}
    for name, (sa, sb, akc, bkc) in shapes.items():
        a = torch.randn(*sa, device="cuda"); b = torch.randn(*sb, device="cuda")
        M = sa[0] if akc else sa[1]; Nn = sb[0] if bkc else sb[1]; K = sa[1] if akc else sa[0]
        out = torch.empty(M, Nn, device="cuda")
        for _ in range(2): ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out)
        rows = torch.randint(0, M, (64,), device="cuda")
        A = (a if akc else a.t())[rows].double(); Bm = (b if bkc else b.t()).double()
        err = float((out[rows].double() - A @ Bm.t()).abs().max() / (A.abs().max() * Bm.abs().max() * K ** 0.5))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        res[name] = (round(us, 1), round(2.0 * M * Nn * K / us / 1e6, 1), f"{err:.1e}")
        del a, b, out
    print(json.dumps(res))
else:
    for tile in [int(x) for x in os.environ.get('TILES', '0,1,2,3').split(',')]:
        env = dict(os.environ, SNAG_GEMM_TILE=str(tile))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(f"tile={tile}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:])
