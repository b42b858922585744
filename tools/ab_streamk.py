"""A/B on one device: split-K (SNAG_GEMM_SK=1) vs stream-K (default) on the shapes that split; child prints us per call."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ".")
    from snag_amd import ops
    N, T = 15000, 158376
    shapes = {"l1_fwd_TT": ((4096, N), (4096, 128), False, False), "l1_bwd": ((N, 128), (N, 4096), False, False),
              "txt_fwd_TT": ((768, N), (768, 128), False, False), "txt_bwd": ((N, 128), (N, 768), False, False),
              "dwqkv": ((45000, 384), (45000, 128), False, False), "dwo": ((45000, 128), (45000, 128), False, False),
              "l2_dw": ((N, 128), (N, 128), False, False), "mmea_dW_up": ((T, 300), (T, 400), False, False), "mmea_dW_img": ((39594, 300), (39594, 2048), False, False)}
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
    for mode, blocks in (("1", "0"), ("0", "1024"), ("0", "768"), ("0", "1280"), ("2", "1024")):
        env = dict(os.environ, SNAG_GEMM_SK=mode, SNAG_GEMM_SK_BLOCKS=blocks)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(f"SK={mode} blocks={blocks}", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
