"""s_memtime stamps of the B-resident streaming GEMM (-DSNAG_GEMM_STAMP=1 variant via SNAG_HIP_LIB; SNAG_GEMM_STREAM_SINGLE=1):
workgroup 7, waves 0 and 5: B images | barrier | first row block: multiply, epilogue | rest."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import _lib, ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (15000, 128, 128)
bkc = os.environ.get("STAMP_BKC", "1") == "1"
a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda") if bkc else torch.randn(K, N, device="cuda")
out = torch.empty(M, N, device="cuda")
big = torch.empty(300_000_000, device="cuda")
for _ in range(10):
    big.add_(1.0)
    ops.gemm(a, b, a_kcontig=True, b_kcontig=bkc, out=out)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_uint64 * 72)()
lib.snag_debug_read_gemm_stamps.argtypes = [ctypes.c_void_p]
lib.snag_debug_read_gemm_stamps(buf)
for w in (0, 1):
    s = list(buf)[9 * w:9 * w + 7]
    print(f"wave {'0' if w == 0 else '5'}: images {s[1]-s[0]:6d} | barrier {s[2]-s[1]:6d} | to first multiply {s[3]-s[2]:6d} | multiply {s[4]-s[3]:6d} | epilogue {s[5]-s[4]:6d} | rest {s[6]-s[5]:6d} | total {s[6]-s[0]:6d} cycles (100 MHz counter x ?)")
