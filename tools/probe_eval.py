"""Config 4 (DBP15K ja_en eval shape): 10,500 x 10,500 x 1,200 similarity GEMM + CSLS(k=3) + ranks, timed with HIP events."""
import sys, time, torch
sys.path.insert(0, '.')
from snag_amd import evalsim as E
n, d = 10500, 1200
g = torch.Generator().manual_seed(4)
left = torch.randn(n, d, generator=g)
right = left + 10.0 * torch.randn(n, d, generator=g)
xl, xr = torch.nn.functional.normalize(left).cuda(), torch.nn.functional.normalize(right).cuda()
for _ in range(2): E.rank_metrics(xl, xr, csls_k=3)
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
e0, e1, e2 = ev(), ev(), ev()
e0.record()
for _ in range(5): dist = E.pairwise_distances(xl, xr)
e1.record()
for _ in range(5): res = E.rank_metrics(xl, xr, csls_k=3)
e2.record(); torch.cuda.synchronize()
t_gemm = e0.elapsed_time(e1) / 5; t_all = e1.elapsed_time(e2) / 5
fl = 2.0 * n * n * d
print(f"sim GEMM (+norms, dist epilogue): {t_gemm:.3f} ms = {fl / t_gemm / 1e9:.1f} TFLOP/s ({fl / t_gemm / 1e9 / 157.3 * 100:.1f}% of 157.3 f32 MFMA peak)")
print(f"rank_metrics end to end (GEMM + CSLS top-3 rows/cols + ranks both directions + summaries): {t_all:.3f} ms; hits@1/10/50 l2r = {res['hits_l2r']}")
print("(the comparison with the CPU oracle at this size lives in tests/test_eval_gpu.py::test_config4_shape_properties)")
