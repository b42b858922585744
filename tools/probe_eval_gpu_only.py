"""rank_metrics at the config-4 shape, GPU only (for rocprofv3 kernel traces)."""
import sys, torch
sys.path.insert(0, '.')
from snag_amd import evalsim as E
n, d = 10500, 1200
g = torch.Generator().manual_seed(4)
left = torch.randn(n, d, generator=g)
right = left + 10.0 * torch.randn(n, d, generator=g)
xl = torch.nn.functional.normalize(left).cuda(); xr = torch.nn.functional.normalize(right).cuda()
for _ in range(6): res = E.rank_metrics(xl, xr, csls_k=3)
torch.cuda.synchronize()
print(res["hits_l2r"])
