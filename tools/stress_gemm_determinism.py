"""Race screen for the GEMM loops: the same launch many times, every result compared bit for bit with the first (a k-loop that read
an LDS image before it was complete, or overwrote one still in use, shows up as a differing run)."""
import sys, torch
sys.path.insert(0, ".")
from snag_amd import ops
torch.manual_seed(0)
cases = [("l1 fwd (ping-pong, K-contiguous operands: the engine's layout)", (15000, 4096), (128, 4096), True, True),
         ("l1 fwd (ping-pong)", (4096, 15000), (4096, 128), False, False), ("dl1 (ping-pong)", (15000, 4096), (15000, 128), False, False),
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
# the grouped launches of the step (snag_gemm_f32_group): both first projections on eight-wave tiles with balanced K splits, and a group of
# 64 x 128 tiles with K splits
import ctypes
from snag_amd import _lib
from snag_amd._lib import call, ptr, stream
from snag_amd.mkgc import MKGCHotPath
for name, shapes, akc, bkc, hint in (("group: both first projections", [(15000, 128, 4096), (15000, 128, 768)], True, True, True),
                                     ("group: both weight gradients", [(4096, 128, 15000), (768, 128, 15000)], False, False, True),
                                     ("group: dWqkv, dWo, dWf x 2", [(384, 128, 15000), (128, 128, 15000), (256, 128, 15000), (256, 128, 15000)], False, False, False)):
    ops_ = [(torch.randn(*((M, K) if akc else (K, M)), device="cuda"), torch.randn(*((N, K) if bkc else (K, N)), device="cuda")) for M, N, K in shapes]
    splits = MKGCHotPath._balanced_splits([(M, K) for M, N, K in shapes]) if hint else [0] * len(shapes)
    outs = [torch.empty(M, N, device="cuda") for M, N, K in shapes]
    ws = [torch.empty(max(64, _lib.load().snag_gemm_workspace_floats(M, N, K), (s_ + 1) * M * N), device="cuda") for (M, N, K), s_ in zip(shapes, splits)]
    descs = []
    for (M, N, K), (a, b), o, w, s_ in zip(shapes, ops_, outs, ws, splits):
        d = _lib.GemmDesc()
        d.M, d.N, d.K, d.a_kcontig, d.b_kcontig, d.alpha, d.splits = M, N, K, int(akc), int(bkc), 1.0, s_
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc = ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(o), o.stride(0)
        d.workspace, d.workspace_floats = ptr(w), w.numel()
        descs.append(d)
    arr = (ctypes.c_void_p * len(descs))(*[ctypes.addressof(d) for d in descs])
    ref, n_diff = None, 0
    for i in range(100):
        call("snag_gemm_f32_group", arr, len(descs), stream())
        cur = [o.clone() for o in outs]
        if ref is None:
            ref = cur
        elif not all(torch.equal(x, y) for x, y in zip(cur, ref)):
            n_diff += 1
    print(f"{name}: {n_diff} of 99 runs differ", flush=True)
    bad += n_diff
print("OK" if bad == 0 else "RACE SUSPECTED")
sys.exit(1 if bad else 0)
