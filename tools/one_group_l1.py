"""Both first projections (visual [N, 4096] and text [N, 768], weights in the nn.Linear layout) as the engine launches them: ONE grouped launch of
eight-wave tiles with the K splits of MKGCHotPath._balanced_splits + the visual one's reduce, a few times behind a 1.2 GB sweep (cold operands, as
inside the step), for rocprofv3 / tools/collect_gemm_pmc.sh:   python tools/one_group_l1.py [N] [bwd]
bwd: the two weight gradients X^T d_mid instead (A = the tables read row-contiguous, transposed store)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import _lib
from snag_amd._lib import call, ptr, stream
from snag_amd.mkgc import MKGCHotPath
N = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
D, dims = 128, (4096, 768)
lib = _lib.load()
feats = [torch.randn(N, fd, device="cuda") for fd in dims]
big = torch.empty(300_000_000, device="cuda")
descs, keep = [], []
shapes = [(fd, N) for fd in dims] if bwd else [(N, fd) for fd in dims]
splits = MKGCHotPath._balanced_splits(shapes)
for fd, x, s in zip(dims, feats, splits):
    d = _lib.GemmDesc()
    if bwd:          # dW^T [fd, D] = X^T d_mid, stored transposed into [D, fd]
        dm, out = torch.randn(N, D, device="cuda"), torch.empty(D, fd, device="cuda")
        d.M, d.N, d.K, d.a_kcontig, d.b_kcontig, d.c_transposed = fd, D, N, 0, 0, 1
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc = ptr(x), fd, ptr(dm), D, ptr(out), fd
    else:            # mid = relu(X W^T)
        dm, out = torch.randn(D, fd, device="cuda"), torch.empty(N, D, device="cuda")
        d.M, d.N, d.K, d.a_kcontig, d.b_kcontig, d.act = N, D, fd, 1, 1, 1
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc = ptr(x), fd, ptr(dm), fd, ptr(out), D
    ws = torch.empty(max(64, (s + 1) * d.M * d.N), device="cuda")
    d.alpha, d.workspace, d.workspace_floats, d.splits, d.defer_reduce = 1.0, ptr(ws), ws.numel(), s, 1
    descs.append(d); keep += [dm, out, ws]
arr = (ctypes.c_void_p * 2)(*[ctypes.addressof(d) for d in descs])
for _ in range(12):
    big.add_(1.0)
    call("snag_gemm_f32_group", arr, 2, stream())
    call("snag_gemm_reduce_pair", ctypes.byref(descs[0]), ctypes.byref(descs[1]), stream())
torch.cuda.synchronize()
print("splits", splits)
