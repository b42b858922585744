"""Does the row pitch of the feature tables matter to the first projections?  A [N, 4096] table has a 16 KiB pitch: at any moment the
~60,000 row streams of the launch sit at the same k offset of rows 16 KiB apart -- the same few HBM channels if the channel is picked by
address bits below 16 KiB.  Both first projections (or, "bwd", both weight gradients) as the engine's ONE grouped launch, cold (a 1.2 GB
sweep in front of every launch), for pitches fd + pad floats:   python tools/probe_pitch_l1.py [bwd] [pad ...]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import _lib
from snag_amd._lib import call, ptr, stream
from snag_amd.mkgc import MKGCHotPath
args = sys.argv[1:]
bwd = bool(args) and args[0] == "bwd"
pads = [int(x) for x in args[1 if bwd else 0:]] or [0, 32, 64, 96, 160, 288, 544]
N, D, dims = 15000, 128, (4096, 768)
lib = _lib.load()
big = torch.empty(300_000_000, device="cuda")


def run(pad, reps=12):
    feats = [torch.randn(N, fd + pad, device="cuda")[:, :fd] for fd in dims]
    descs, keep = [], []
    shapes = [(fd, N) for fd in dims] if bwd else [(N, fd) for fd in dims]
    splits = MKGCHotPath._balanced_splits(shapes)
    for fd, x, s in zip(dims, feats, splits):
        d = _lib.GemmDesc()
        if bwd:
            dm, out = torch.randn(N, D, device="cuda"), torch.empty(D, fd, device="cuda")
            d.M, d.N, d.K, d.a_kcontig, d.b_kcontig, d.c_transposed = fd, D, N, 0, 0, 1
            d.A, d.lda, d.B, d.ldb, d.C, d.ldc = ptr(x), x.stride(0), ptr(dm), D, ptr(out), fd
        else:
            dm, out = torch.randn(D, fd, device="cuda"), torch.empty(N, D, device="cuda")
            d.M, d.N, d.K, d.a_kcontig, d.b_kcontig, d.act = N, D, fd, 1, 1, 1
            d.A, d.lda, d.B, d.ldb, d.C, d.ldc = ptr(x), x.stride(0), ptr(dm), fd, ptr(out), D
        ws = torch.empty(max(64, (s + 1) * d.M * d.N), device="cuda")
        d.alpha, d.workspace, d.workspace_floats, d.splits, d.defer_reduce = 1.0, ptr(ws), ws.numel(), s, 1
        descs.append(d); keep += [dm, out, ws]
    arr = (ctypes.c_void_p * 2)(*[ctypes.addressof(d) for d in descs])
    ts = []
    for _ in range(reps):
        big.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("snag_gemm_f32_group", arr, 2, stream())
        e1.record()
        call("snag_gemm_reduce_pair", ctypes.byref(descs[0]), ctypes.byref(descs[1]), stream())
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2], ts[0]


for rnd in range(2):
    for pad in pads:
        med, best = run(pad)
        print(f"{'bwd' if bwd else 'fwd'} pad {pad:4d} floats (pitch {4 * (4096 + pad)} B): median {med:6.1f} us  best {best:6.1f} us", flush=True)
