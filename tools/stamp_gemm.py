"""In-kernel cycle stamps of the eight-wave ping-pong GEMM loop (needs a -DSNAG_GEMM_STAMP=1 variant selected with SNAG_HIP_LIB):
per iteration kt = 24..27 of workgroup (row tile 17, K split 1), wave 0 (group 0) and wave 4 (group 1):
  memory half (top -> reads landed), wait at barrier 1, matrix half, wait at barrier 2.   tools/stamp_gemm.py [M K N]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import _lib, ops  # noqa: E402

M, K, N = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (15000, 4096, 128)
ldt = (M + 31) // 32 * 32
xt = torch.randn(K, ldt, device="cuda")[:, :M]
w = torch.randn(K, N, device="cuda")
out = torch.empty(M, N, device="cuda")
KC = os.environ.get("STAMP_KC", "0") == "1"          # both operands K-contiguous (the engine's natural layout)
x, wk = xt.t().contiguous(), w.t().contiguous()
COLD = os.environ.get("STAMP_COLD", "0") == "1"      # a 1.2 GB sweep in front of every launch: A arrives from HBM, as inside the step
big = torch.empty(300_000_000, device="cuda") if COLD else None
for _ in range(30):
    if COLD:
        big.add_(1.0)
    if KC:
        ops.gemm(x, wk, a_kcontig=True, b_kcontig=True, out=out, act="relu")
    else:
        ops.gemm(xt, w, a_kcontig=False, b_kcontig=False, out=out, act="relu")
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_uint64 * 72)()
lib.snag_debug_read_gemm_stamps.argtypes = [ctypes.c_void_p]
rc = lib.snag_debug_read_gemm_stamps(buf)
st = list(buf)
print("rc", rc)
for g in range(2):
    for i in range(4):
        s = st[(g * 4 + i) * 9:(g * 4 + i) * 9 + 9]
        nxt = st[(g * 4 + i + 1) * 9] if i < 3 else None
        print(f"group {g} iter {i}: [A split {s[5] - s[0]:4d} | A loads {s[6] - s[5]:4d} | B split+write {s[7] - s[6]:4d} | B loads {s[8] - s[7]:4d} | lgkm wait {s[1] - s[8]:4d}] "
              f"memory half {s[1] - s[0]:5d}  barrier1 wait {s[2] - s[1]:5d}  matrix half {s[3] - s[2]:5d}  barrier2 wait {s[4] - s[3]:5d}"
              + (f"  loop overhead {nxt - s[4]:4d}  iteration {nxt - s[0]:5d}" if nxt else ""))
print("group 1 start - group 0 start (iter 0):", st[36] - st[0])
