"""Config 4 (DBP15K ja_en eval shape): 10,500 x 10,500 x 1,200 similarity GEMM + CSLS(k=3) + ranks, timed with HIP events."""
import sys, time, torch
sys.path.insert(0, '.')
from snag_amd import evalsim as E
from oracle import ref_cpu as O
n, d = 10500, 1200
g = torch.Generator().manual_seed(4)
left = torch.randn(n, d, generator=g)
right = left + 10.0 * torch.randn(n, d, generator=g)
xl, xr = O.nrm(left).cuda(), O.nrm(right).cuda()
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
t0 = time.perf_counter()
torch.set_num_threads(16)
ro = O.rank_metrics(xl.cpu(), xr.cpu(), csls_k=3)
print(f"CPU oracle rank_metrics (16 threads, count-less-than form, not the reference's 21,000 torch.sort loop): {time.perf_counter() - t0:.2f} s; hits equal: {ro['hits_l2r'] == res['hits_l2r'] and ro['hits_r2l'] == res['hits_r2l']}; ranks equal: {bool((ro['rank_l2r'] == res['rank_l2r'].cpu()).all())}")

dl = (ro['rank_l2r'] - res['rank_l2r'].cpu().long()).abs(); dr = (ro['rank_r2l'] - res['rank_r2l'].cpu().long()).abs()
print(f"rank differences vs CPU oracle (MKL GEMM order): l2r {int((dl > 0).sum())}/{n} rows differ, max |d| {int(dl.max())}; r2l {int((dr > 0).sum())}/{n}, max {int(dr.max())}")
dd = (ro['dist'] - res['dist'].cpu()).abs().max()
print(f"max |dist_gpu - dist_cpu| = {float(dd):.2e}")
