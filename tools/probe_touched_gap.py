"""Where the configs[4] scorer's launch boundaries lose time: the entry (pass A, pass B, hot rows) followed by two one-thread launches, 40 times;
run under rocprofv3 --kernel-trace and read with tools/prof_timeline.py <dir> <out> touched_a_kernel."""
import sys, torch
sys.path.insert(0, '.')
from snag_amd import ops
from snag_amd._lib import call, ptr, stream
dev = 'cuda'
g = torch.Generator().manual_seed(5)
N, R, D, K, B = 1_000_000, 1000, 256, 128, 1024
e16 = torch.empty(N, D, device=dev).normal_().to(torch.bfloat16)
r16 = torch.randn(R, D, generator=g).to(dev).to(torch.bfloat16)
d_ent, d_rel = torch.zeros(N, D, device=dev), torch.zeros(R, D, device=dev)
tick = torch.zeros(1, device=dev, dtype=torch.int64)
idx = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).to(dev) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
for _ in range(40):
    ops.score_margin_fwd_bwd_touched(e16, r16, *idx, 12.0, d_ent=d_ent, d_rel=d_rel)
    call("snag_tick", ptr(tick), stream())
    call("snag_tick", ptr(tick), stream())
torch.cuda.synchronize()
