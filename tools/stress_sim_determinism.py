"""Similarity GEMM (snag_pairwise_dist): 30 repeated launches per shape must be bit-identical (race screen for the eight-wave tile in
grouped order), and 64 sampled rows are compared with an fp64 evaluation of |x - y|^2."""
import sys, torch
sys.path.insert(0, '.')
from snag_amd import evalsim as E
g = torch.Generator().manual_seed(1)
for (n, m, d) in ((10500, 10500, 1200), (8191, 9001, 300), (4097, 12000, 77)):
    x = torch.nn.functional.normalize(torch.randn(n, d, generator=g)).cuda()
    y = torch.nn.functional.normalize(torch.randn(m, d, generator=g)).cuda()
    ref = E.pairwise_distances(x, y).clone()
    bad = 0
    for _ in range(30):
        bad += int(not torch.equal(E.pairwise_distances(x, y), ref))
    # against fp64 on a sample of rows
    rows = torch.randint(0, n, (64,), generator=g)
    want = ((x[rows].double()[:, None, :] - y.double()[None, :, :]) ** 2).sum(-1) if m * 64 * d < 2e9 else None
    err = float((ref[rows].double() - want).abs().max()) if want is not None else float('nan')
    print(f"{n}x{m}x{d}: {bad} of 30 repeats differ; max |err| vs fp64 on 64 rows {err:.2e}")
