"""Time snag_gemm_f32 on the step's big shapes under forced tile/split settings (env SNAG_GEMM_TILE / SNAG_GEMM_SPLITS)."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ".")
    from snag_amd import ops
    N = 15000
    shapes = {"l1_fwd": ((N, 4096), (128, 4096), True, True), "l1_fwd_xT": ((4096, N), (128, 4096), False, True),
              "l1_fwd_TT": ((4096, N), (4096, 128), False, False), "qkv_TT": ((128, 45000), (128, 384), False, False), "txt_fwd": ((N, 768), (128, 768), True, True), "l1_bwd": ((N, 128), (N, 4096), False, False),
              "qkv": ((45000, 128), (384, 128), True, True), "dx": ((45000, 384), (384, 128), True, False),
              "dwqkv": ((45000, 384), (45000, 128), False, False)}
    res = {}
    for name, (sa, sb, akc, bkc) in shapes.items():
        a = torch.randn(*sa, device="cuda"); b = torch.randn(*sb, device="cuda")
        M = sa[0] if akc else sa[1]; Nn = sb[0] if bkc else sb[1]; K = sa[1] if akc else sa[0]
        out = torch.empty(M, Nn, device="cuda")
        for _ in range(3): ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, b, a_kcontig=akc, b_kcontig=bkc, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res[name] = (round(us, 1), round(2.0 * M * Nn * K / us / 1e6, 1))
    print(json.dumps(res))
else:
    for tile in [int(x) for x in os.environ.get('TILES','1,2').split(',')]:
        for splits in [int(x) for x in os.environ.get('SPLITS','1,2,4,8,16,32').split(',')]:
            env = dict(os.environ, SNAG_GEMM_TILE=str(tile), SNAG_GEMM_SPLITS=str(splits))
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(f"tile={tile} splits={splits:2d}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
