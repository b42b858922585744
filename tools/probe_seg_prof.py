import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snag_amd import ops
g = torch.Generator().manual_seed(1)
N, D, K, B = 15000, 128, 32, 1024
ent = torch.randn(N, D, generator=g).cuda(); rel = torch.randn(558, D, generator=g).cuda()
h, r, t = (torch.randint(0, hi, (B,), generator=g, dtype=torch.int32).cuda() for hi in (N, 558, N))
neg = torch.randint(0, N, (B, K), generator=g, dtype=torch.int32).cuda()
de, dr = torch.empty_like(ent), torch.empty_like(rel)
for _ in range(30):
    ops.score_margin_fwd_bwd_seg(ent, rel, h, r, t, neg, 12.0, d_ent=de, d_rel=dr)
torch.cuda.synchronize()
