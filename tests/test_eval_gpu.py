"""GPU parity: similarity eval (distance GEMM, CSLS, ranks, Hits@k, mutual NN) vs golden G4/G5 and the oracle."""
import numpy as np
import pytest
import torch

from helpers import T, assert_close, load
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("name", ["d64", "d1200", "d64_nocsls"])
def test_eval_golden(name):
    from snag_amd import evalsim as E
    z = load(f"g4_eval_{name}.npz")
    final = O.nrm(T(z["emb"])).to(DEV)
    left, right = T(z["left"], DEV), T(z["right"], DEV)
    k = int(z["csls_k"])
    xl, xr = final[left].contiguous(), final[right].contiguous()
    assert_close(E.pairwise_distances(xl, xr), z["dist"], atol=1e-5, what="dist")          # 1e-4 bar, fp32 GEMM order differs
    if k:
        assert_close(E.csls_sim(1 - T(z["dist"], DEV), k), z["csls"], atol=1e-5, what="csls")
    res = E.rank_metrics(xl, xr, csls_k=k)
    assert np.array_equal(res["rank_l2r"].cpu().numpy(), z["rank_l2r"])                     # ranking indices: bit-exact, both directions
    assert np.array_equal(res["rank_r2l"].cpu().numpy(), z["rank_r2l"])                     # (r2l = the reference's own per-column sorts, main.py:422-424)
    for d in ("l2r", "r2l"):
        assert np.array_equal(np.asarray(res[f"hits_{d}"]), z[f"hits_{d}"]), (res[f"hits_{d}"], z[f"hits_{d}"])
        assert res[f"mr_{d}"] == pytest.approx(float(z[f"mr_{d}"]), rel=1e-12)             # unrounded
        assert res[f"mrr_{d}"] == pytest.approx(float(z[f"mrr_{d}"]), rel=1e-12)
        assert f"{res[f'mr_{d}']:.3f}" == f"{float(z[f'mr_{d}_3dp']):.3f}" and f"{res[f'mrr_{d}']:.3f}" == f"{float(z[f'mrr_{d}_3dp']):.3f}"


def test_mutual_nearest_golden():
    from snag_amd import evalsim as E
    z = load("g5_il.npz")
    _, _, pairs = E.mutual_nearest(T(z["final"], DEV), z["left"].tolist(), z["right"].tolist())
    assert np.array_equal(np.array(pairs, dtype=np.int64), z["links"])


@pytest.mark.parametrize("n,m,d,k", [(257, 300, 96, 3), (64, 64, 20, 10), (1000, 33, 128, 16), (5, 700, 7, 1)])
def test_pieces_vs_oracle(n, m, d, k):
    from snag_amd import evalsim as E
    g = torch.Generator().manual_seed(n + m)
    x, y = O.nrm(torch.randn(n, d, generator=g)), O.nrm(torch.randn(m, d, generator=g))
    dist = E.pairwise_distances(x.to(DEV), y.to(DEV))
    assert_close(dist, O.pairwise_distances(x, y), atol=1e-5)
    if k <= min(n, m):
        sim = 1 - O.pairwise_distances(x, y)
        assert_close(E.csls_sim(sim.to(DEV), k), O.csls_sim(sim, k), atol=2e-6)
    assert_close(E.pairwise_distances(x.to(DEV)), O.pairwise_distances(x), atol=1e-5)


def test_config4_shape_properties():
    """DBP15K ja_en eval shape (10,500 x 10,500 x 1,200, csls_k = 3): size-independent properties."""
    from snag_amd import evalsim as E
    n, d = 10500, 1200
    g = torch.Generator().manual_seed(4)
    left = torch.randn(n, d, generator=g)
    right = left + 10.0 * torch.randn(n, d, generator=g)
    xl, xr = O.nrm(left).to(DEV), O.nrm(right).to(DEV)
    res = E.rank_metrics(xl, xr, csls_k=3)
    rl, rr = res["rank_l2r"].cpu(), res["rank_r2l"].cpu()
    assert int(rl.min()) >= 0 and int(rl.max()) < n and int(rr.min()) >= 0 and int(rr.max()) < n
    assert 0.2 < res["hits_l2r"][0] < 1.0 and res["hits_l2r"][0] <= res["hits_l2r"][1] <= res["hits_l2r"][2]
    # transposing the problem swaps the two directions: the same products, but a.b and b.a sum their split-bf16 product families in
    # mirrored order, so the two distance matrices agree to f32 rounding and a rank may move only across a proven near-tie
    res_t = E.rank_metrics(xr, xl, csls_k=3)
    d_t = (1 - O.csls_sim(1 - res_t["dist"].cpu(), 3)).t()
    d_0 = 1 - O.csls_sim(1 - res["dist"].cpu(), 3)
    tie = 2e-6
    assert 2 * float((d_t - d_0).abs().max()) <= tie
    for got, ref, dim in ((res_t["rank_l2r"].cpu(), rr, 0), (res_t["rank_r2l"].cpu(), rl, 1)):
        diff = (got.long() - ref.long()).abs()
        for i in diff.nonzero().flatten().tolist():
            line = d_0[i] if dim == 1 else d_0[:, i]
            assert int(diff[i]) <= int(((line - d_0[i, i]).abs() <= tie).sum()) - 1, (i, int(diff[i]))
    # a simultaneous permutation of the pairs permutes the ranks
    perm = torch.randperm(n, generator=g)
    res_p = E.rank_metrics(xl[perm.to(DEV)].contiguous(), xr[perm.to(DEV)].contiguous(), csls_k=3)
    assert torch.equal(res_p["rank_l2r"].cpu(), rl[perm])
    # spot-check 64 rows against the oracle's count-less-than on the same distance matrix
    dist = res["dist"]
    dd = 1 - O.csls_sim(1 - dist.cpu(), 3)
    rows = torch.arange(0, n, n // 64)[:64]
    ref = (dd[rows] < dd[rows, rows].view(-1, 1)).sum(1)
    assert torch.equal(rl[rows].long(), ref)
    # the whole problem against the oracle (its own CPU GEMM: a different fp32 summation order over the 1,200 products).
    # north_star: "bit-exact for ranking indices" -- so every rank that differs must be PROVEN a near-tie: with
    # delta = max |Dist'_gpu - Dist'_oracle| over the whole 10,500 x 10,500 matrix, candidate j can change sides of the true match i
    # only if |Dist'[i,j] - Dist'[i,i]| <= 2 delta; a row's (column's) rank may differ by at most the number of such candidates.
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    ro = O.rank_metrics(xl.cpu(), xr.cpu(), csls_k=3)
    d_o = ro["dist"]                                              # the oracle's CSLS-adjusted distances
    delta = float((dd - d_o).abs().max())
    eps = 5e-6                                                    # stated tie window; must cover 2 delta
    assert 2 * delta <= eps, f"GPU and oracle CSLS distances differ by {delta:.3e}: the tie window {eps:g} does not cover it"
    diag = d_o.diagonal()
    unexplained = 0
    for ref_rank, got, dim in ((ro["rank_l2r"], rl, 1), (ro["rank_r2l"], rr, 0)):
        diff = (ref_rank.long() - got.long()).abs()
        for i in diff.nonzero().flatten().tolist():
            line = d_o[i] if dim == 1 else d_o[:, i]
            near_ties = int(((line - diag[i]).abs() <= eps).sum()) - 1          # candidates within the window, the match itself excluded
            if int(diff[i]) > near_ties:
                unexplained += 1
    assert unexplained == 0, f"{unexplained} rank differences are not near-ties (window {eps:g}, delta {delta:.2e})"
    # Hits@k can then differ only through those proven near-tie rows
    for name, ref_rank, got in (("l2r", ro["rank_l2r"], rl), ("r2l", ro["rank_r2l"], rr)):
        n_diff = int((ref_rank.long() != got.long()).sum())
        for a, b in zip(ro[f"hits_{name}"], res[f"hits_{name}"]):
            assert abs(a - b) <= n_diff / n + 0.5e-4, (name, a, b, n_diff)
    # MR/MRR consistent with ranks
    assert res["mr_l2r"] == pytest.approx(float((rl.double() + 1).mean()), rel=1e-12)
    assert res["mrr_r2l"] == pytest.approx(float((1 / (rr.double() + 1)).mean()), rel=1e-12)


def test_eval_rejects_bad_input():
    from snag_amd import evalsim as E, _lib
    x = torch.zeros(4, 8, device=DEV)
    with pytest.raises(ValueError):
        E.pairwise_distances(x, torch.zeros(4, 9, device=DEV))
    with pytest.raises(_lib.SnagHipError):
        E.csls_sim(x, 17)                      # k > 16 unsupported, and > matrix like torch.topk
    with pytest.raises(ValueError):
        E.rank_metrics(x, torch.zeros(5, 8, device=DEV))


@pytest.mark.parametrize("N,D,B,idt", [(500, 128, 64, torch.int64), (15000, 128, 200, torch.int32), (333, 64, 17, torch.int32)])
def test_link_prediction_ranks_vs_oracle(N, D, B, idt):
    """Hits@1/10 of the MKGC metric: rank of the true tail against the full entity table, filtered setting; ranks bit-exact."""
    from snag_amd import evalsim as E
    g = torch.Generator().manual_seed(N + B)
    ent, rel = torch.randn(N, D, generator=g), torch.randn(9, D, generator=g) * 0.3
    h = torch.randint(0, N, (B,), generator=g).to(idt)
    r = torch.randint(0, 9, (B,), generator=g).to(idt)
    # plant structure so that Hits@10 is neither 0 nor 1: t = the entity closest to h + r for half of the triples
    q = O.nrm(ent[h.long()]) + O.nrm(rel[r.long()])
    near = ((q.unsqueeze(1) - O.nrm(ent).unsqueeze(0)) ** 2).sum(-1).argsort(dim=1)
    t = torch.where(torch.rand(B, generator=g) < 0.5, near[torch.arange(B), torch.randint(0, 20, (B,), generator=g)],
                    torch.randint(0, N, (B,), generator=g)).to(idt)
    filt = torch.rand(B, N, generator=g) < 0.01
    filt[torch.arange(B), t.long()] = False
    ref = O.rank_full_table(ent, rel, h, r, t, filt)
    res = E.link_prediction(ent.to(DEV), rel.to(DEV), h.to(DEV), r.to(DEV), t.to(DEV), filt.to(torch.uint8).to(DEV))
    got = res["rank"].cpu().long()
    # the two implementations sum the 128 squares in different orders: allow a flip only where the scores tie to 1e-6
    if not torch.equal(got, ref):
        sc = res["scores"].cpu()
        st = sc.gather(1, t.long().view(-1, 1))
        near_ties = ((sc - st).abs() < 2e-6).sum(1) - 1
        assert bool(((got - ref).abs() <= near_ties).all()), (got - ref).abs().max()
    ref_u = O.rank_full_table(ent, rel, h, r, t, None)
    res_u = E.link_prediction(ent.to(DEV), rel.to(DEV), h.to(DEV), r.to(DEV), t.to(DEV))
    assert bool((res_u["rank"].cpu().long() >= got).all())                  # filtering can only improve a rank
    assert int((res_u["rank"].cpu().long() - ref_u).abs().max()) <= 1
    assert 0.0 < res["hits"][1] < 1.0 and res["hits"][0] <= res["hits"][1]


def test_eval_past_2_31_matrix_elements():
    """48,000 x 48,000 distance matrix (2.3e9 elements): CSLS + ranks of sampled rows / columns equal a direct recomputation."""
    from snag_amd import evalsim as E
    n, d = 48000, 64
    g = torch.Generator().manual_seed(1)
    x = torch.nn.functional.normalize(torch.randn(n, d, generator=g)).to(DEV)
    y = torch.nn.functional.normalize(x.cpu() + 1.5 * torch.nn.functional.normalize(torch.randn(n, d, generator=g))).to(DEV)
    res = E.rank_metrics(x, y, csls_k=3)
    assert 0.0 < res["hits_l2r"][0] < 1.0
    sim = 1.0 - res["dist"]
    r1 = sim.topk(3, dim=1)[0].mean(1)
    r2 = sim.topk(3, dim=0)[0].mean(0)
    for i in (0, 777, 44739, 44740, n - 1):               # 44,739 * 48,000 is the last row below 2^31 elements
        drow = 1.0 - ((2.0 * sim[i] - r1[i]) - r2)
        assert int((drow < drow[i]).sum()) == int(res["rank_l2r"][i])
        dcol = 1.0 - ((2.0 * sim[:, i] - r1) - r2[i])
        assert int((dcol < dcol[i]).sum()) == int(res["rank_r2l"][i])
    rows = torch.tensor([0, 44740, n - 1], device=DEV)
    assert torch.equal(E.pairwise_distances(x[rows].contiguous(), y), res["dist"][rows])
    del res, sim
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,m,d", [(1, 1, 1), (65, 130, 33), (300, 257, 64), (500, 500, 1200)])
def test_cityblock_distance_and_ranks(n, m, d):
    """--distance 1 (main.py:388-391): sum_k |x - y| vs torch.cdist(p=1); ranks through the same count-less-than kernels."""
    from snag_amd import evalsim as E
    g = torch.Generator().manual_seed(n + m + d)
    x, y = torch.randn(n, d, generator=g), torch.randn(m, d, generator=g)
    got = E.cityblock_distances(x.to(DEV), y.to(DEV))
    ref = torch.cdist(x.double(), y.double(), p=1)
    assert_close(got, ref, atol=1e-5 * max(1.0, float(ref.max())))
    if n == m and n > 1:
        res = E.rank_metrics(x.to(DEV), (x + 0.7 * y).to(DEV), csls_k=0, distance=1)
        dd = torch.cdist(x.double(), (x + 0.7 * y).double(), p=1)
        rk = (dd < dd.diag().view(-1, 1)).sum(1)
        assert float((res["rank_l2r"].cpu().long() != rk).float().mean()) < 0.01          # fp32 vs fp64 near-ties only


def test_link_prediction_gemm_method_agrees_with_exact():
    """method="gemm": |q|^2 + |t|^2 - 2 q.t on the matrix cores; same ranks as the difference-form scorer up to near-ties."""
    from snag_amd import evalsim as E
    g = torch.Generator().manual_seed(12)
    N, R, D, B = 3000, 11, 128, 700
    ent = (torch.randn(N, D, generator=g) * 0.7).to(DEV)
    ent[5].zero_()                                            # a zero row: normalises to zero in both forms
    rel = (torch.randn(R, D, generator=g) * 0.3).to(DEV)
    h = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(DEV)
    r = torch.randint(0, R, (B,), generator=g, dtype=torch.int32).to(DEV)
    t = torch.randint(0, N, (B,), generator=g, dtype=torch.int32).to(DEV)
    filt = (torch.rand(B, N, generator=g) < 0.002).to(torch.uint8).to(DEV)
    a = E.link_prediction(ent, rel, h, r, t, filt, method="exact")
    b = E.link_prediction(ent[:, :].contiguous(), rel, h, r, t, filt, method="gemm")
    assert_close(b["scores"], a["scores"], atol=2e-5)
    diff = (a["rank"].long() - b["rank"].long()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.01
    for x, y in zip(a["hits"], b["hits"]):
        assert abs(x - y) <= 2.0 / B
    with pytest.raises(ValueError):
        E.link_prediction(ent, rel, h, r, t, method="fast")


def test_filter_mask_from_known_triples():
    """snag_build_filter_mask: the filtered protocol's mask built on the device from the sorted key table."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(2)
    N, R, T, B = 500, 7, 6000, 300
    kh, kr, kt = (torch.randint(0, hi, (T,), generator=g, dtype=torch.int32) for hi in (N, R, N))
    kh[:50], kr[:50] = 3, 2                                   # a long (h, r) run (more than one wave stride)
    kt[:50] = torch.randperm(N, generator=g)[:50].to(torch.int32)
    known = ops.triple_keys(kh.to(DEV), kr.to(DEV), kt.to(DEV), N, R)
    sel = torch.randint(0, T, (B,), generator=g)
    h, r, t = kh[sel].clone(), kr[sel].clone(), kt[sel].clone()
    h[0], r[0], t[0] = 3, 2, kt[0]
    h[1], r[1] = N - 1, R - 1                                 # possibly no known triple at all for this pair
    got = ops.build_filter_mask(h.to(DEV), r.to(DEV), t.to(DEV), N, R, known).cpu()
    ref = torch.zeros(B, N, dtype=torch.uint8)
    by_pair = {}
    for a, b_, c in zip(kh.tolist(), kr.tolist(), kt.tolist()):
        by_pair.setdefault((a, b_), set()).add(c)
    for i, (a, b_, c) in enumerate(zip(h.tolist(), r.tolist(), t.tolist())):
        for x in by_pair.get((a, b_), ()):
            if x != c:
                ref[i, x] = 1
    assert torch.equal(got, ref)
    assert int(got[0].sum()) >= 40


@pytest.mark.parametrize("n,k,from_dist", [(1003, 3, True), (1024, 5, False), (517, 10, True), (130, 16, False)])
def test_dwordx4_passes_equal_the_scalar_ones(n, k, from_dist):
    """The CSLS neighbourhood and rank passes read a matrix whose rows are 16-byte addressable four columns per lane; any other pitch
    takes the scalar kernels.  Same matrix in both layouts (n % 4 != 0 cases: the last lane's columns are ragged): identical r1 / r2
    (means of the same top-k values) and identical ranks."""
    import ctypes
    from snag_amd import _lib
    from snag_amd._lib import call, ptr, stream
    from snag_amd import evalsim as E
    g = torch.Generator().manual_seed(n * 7 + k)
    vals = torch.rand(n, n, generator=g)
    vals[torch.arange(n), torch.arange(n)] *= 0.2                      # planted diagonal, so the ranks are not all ~n/2
    bufs = []
    for pitch in ((n + 31) // 32 * 32, n + 1 if (n + 1) % 4 else n + 3):     # second pitch: rows not 16-byte addressable
        b = torch.zeros(n, pitch, device=DEV)
        b[:, :n] = vals.to(DEV)
        bufs.append(b[:, :n])
    assert bufs[0].stride(0) % 4 == 0 and bufs[1].stride(0) % 4 != 0
    res = []
    for mat in bufs:
        r1, r2 = E._neighbourhood(mat, k, from_dist)
        rl = torch.empty(n, device=DEV, dtype=torch.int32)
        rr = torch.empty(n, device=DEV, dtype=torch.int32)
        dist = mat if from_dist else None
        if dist is None:                                               # ranks take distances: 1 - sim in the same layout
            d = torch.zeros(n, mat.stride(0), device=DEV)
            d[:, :n] = 1.0 - mat
            dist = d[:, :n]
        call("snag_rank_diagonal", ptr(dist), dist.stride(0), n, n, ptr(r1), ptr(r2), ptr(rl), ptr(rr), stream())
        rl0 = torch.empty(n, device=DEV, dtype=torch.int32)
        rr0 = torch.empty(n, device=DEV, dtype=torch.int32)
        call("snag_rank_diagonal", ptr(dist), dist.stride(0), n, n, None, None, ptr(rl0), ptr(rr0), stream())      # and without CSLS
        # the row-sharded entries on the whole matrix as one shard: the same ranks
        rb = torch.empty(n, device=DEV, dtype=torch.int32)
        diag = torch.empty(n, device=DEV)
        cb = torch.empty(n, device=DEV, dtype=torch.int32)
        call("snag_rank_rows_block", ptr(dist), dist.stride(0), n, n, 0, ptr(r1), ptr(r2), ptr(rb), ptr(diag), stream())
        call("snag_rank_cols_block", ptr(dist), dist.stride(0), n, n, ptr(r1), ptr(r2), ptr(diag), ptr(cb), stream())
        assert torch.equal(rb, rl) and torch.equal(cb, rr)
        res.append((r1, r2, rl, rr, rl0, rr0))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # against the definition
    s = (1.0 - vals) if from_dist else vals
    assert_close(res[0][0], s.topk(k, dim=1)[0].mean(1), atol=1e-6)
    assert_close(res[0][1], s.topk(k, dim=0)[0].mean(0), atol=1e-6)
    dd = vals if from_dist else 1.0 - vals
    want = (dd < dd.diag().unsqueeze(1)).sum(1).to(torch.int32)
    assert torch.equal(res[0][4].cpu(), want)
    assert torch.equal(res[0][5].cpu(), (dd < dd.diag().unsqueeze(0)).sum(0).to(torch.int32))


@pytest.mark.parametrize("n,m", [(1003, 517), (512, 1024), (130, 131)])
def test_argmin_passes_in_both_layouts_with_ties(n, m):
    """snag_argmin_rows_cols on 16-byte addressable rows (dwordx4 kernels) and on an odd pitch (scalar kernels): the same arg-mins, and
    among equal minima the smallest index (numpy's first occurrence) -- values quantised to 1/64 so that ties are everywhere."""
    from snag_amd import _lib
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(n + 3 * m)
    vals = torch.floor(torch.rand(n, m, generator=g) * 64) / 64
    outs = []
    for pitch in ((m + 31) // 32 * 32, m + 1 if (m + 1) % 4 else m + 3):
        b = torch.zeros(n, pitch, device=DEV)
        b[:, :m] = vals.to(DEV)
        mat = b[:, :m]
        pl = torch.empty(n, device=DEV, dtype=torch.int32)
        pr = torch.empty(m, device=DEV, dtype=torch.int32)
        work = torch.empty(_lib.load().snag_eval_work_floats(n, m), device=DEV)
        call("snag_argmin_rows_cols", ptr(mat), mat.stride(0), n, m, ptr(pl), ptr(pr), ptr(work), stream())
        outs.append((pl.cpu().numpy(), pr.cpu().numpy()))
    v = vals.numpy()
    for pl, pr in outs:
        assert np.array_equal(pl, v.argmin(1)) and np.array_equal(pr, v.argmin(0))


def test_csls_neighbourhood_fuzz_rectangular_shapes_and_pitches():
    """snag_csls_neighbourhood on random rectangular shapes, both layouts (16-byte addressable rows or not), k up to 16, similarities and
    distances: r1 / r2 are the means of the k largest entries of every row / column (torch.topk), whichever kernels the layout picks."""
    from snag_amd import evalsim as E
    g = torch.Generator().manual_seed(77)
    for case in range(24):
        n = int(torch.randint(1, 700, (1,), generator=g))
        m = int(torch.randint(1, 700, (1,), generator=g))
        k = int(torch.randint(1, min(16, n, m) + 1, (1,), generator=g))
        from_dist = bool(case % 2)
        pitch = m + int(torch.randint(0, 9, (1,), generator=g))
        vals = torch.rand(n, m, generator=g)
        if case % 3 == 0:
            vals = torch.floor(vals * 16) / 16                          # ties
        buf = torch.zeros(n, pitch, device=DEV)
        buf[:, :m] = vals.to(DEV)
        r1, r2 = E._neighbourhood(buf[:, :m], k, from_dist)
        s = (1.0 - vals) if from_dist else vals
        assert_close(r1, s.topk(k, dim=1)[0].mean(1), atol=2e-6, what=f"rows, case {case}: {n}x{m} k={k} pitch={pitch}")
        assert_close(r2, s.topk(k, dim=0)[0].mean(0), atol=2e-6, what=f"columns, case {case}: {n}x{m} k={k} pitch={pitch}")


def test_rank_diagonal_fuzz_against_the_float32_definition():
    """snag_rank_diagonal with and without CSLS on random square matrices and pitches (ties included): the counts
    #{j : D'[i,j] < D'[i,i]} / #{i : D'[i,j] < D'[j,j]} with D' = 1 - ((2 (1 - D) - r1) - r2) evaluated in float32 in the kernels' operation
    order (main.py:392, utils.py:433-434) -- exact integers."""
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(123)
    for case in range(16):
        n = int(torch.randint(1, 600, (1,), generator=g))
        pitch = n + int(torch.randint(0, 9, (1,), generator=g))
        d = torch.rand(n, n, generator=g)
        if case % 2:
            d = torch.floor(d * 32) / 32
        r1, r2 = torch.rand(n, generator=g), torch.rand(n, generator=g)
        buf = torch.zeros(n, pitch, device=DEV)
        buf[:, :n] = d.to(DEV)
        mat = buf[:, :n]
        for csls in (False, True):
            rl = torch.empty(n, device=DEV, dtype=torch.int32)
            rr = torch.empty(n, device=DEV, dtype=torch.int32)
            a, b = (r1.to(DEV), r2.to(DEV)) if csls else (None, None)
            call("snag_rank_diagonal", ptr(mat), mat.stride(0), n, n, ptr(a), ptr(b), ptr(rl), ptr(rr), stream())
            dd = d
            if csls:
                s = 1.0 - d
                dd = 1.0 - ((2.0 * s - r1[:, None]) - r2[None, :])
            want_l = (dd < dd.diag()[:, None]).sum(1).to(torch.int32)
            want_r = (dd < dd.diag()[None, :]).sum(0).to(torch.int32)
            assert torch.equal(rl.cpu(), want_l) and torch.equal(rr.cpu(), want_r), f"case {case}: n={n} pitch={pitch} csls={csls}"


def test_rank_full_table_fuzz_with_filters():
    """snag_rank_full_table: rank[b] = #{n not filtered : score[b, n] < score[b, t_b]} on random sizes (entity counts that are and are not
    multiples of 4: dwordx4 rows / scalar rows), with and without a filter mask, ties included -- exact counts."""
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(9)
    for case in range(16):
        B = int(torch.randint(1, 70, (1,), generator=g))
        N = int(torch.randint(1, 2000, (1,), generator=g))
        if case % 2 == 0:
            N = (N + 3) // 4 * 4
        sc = torch.rand(B, N, generator=g)
        if case % 3 == 0:
            sc = torch.floor(sc * 64) / 64
        t = torch.randint(0, N, (B,), generator=g, dtype=torch.int32)
        filt = (torch.rand(B, N, generator=g) < 0.05).to(torch.uint8) if case % 4 else None
        rank = torch.empty(B, device=DEV, dtype=torch.int32)
        scd, td = sc.to(DEV), t.to(DEV)
        fd = filt.to(DEV) if filt is not None else None
        call("snag_rank_full_table", ptr(scd), B, N, ptr(td), 4, ptr(fd), ptr(rank), stream())
        ref = sc[torch.arange(B), t.long()][:, None]
        less = sc < ref
        if filt is not None:
            less &= filt == 0
        assert torch.equal(rank.cpu(), less.sum(1).to(torch.int32)), f"case {case}: B={B} N={N} filter={filt is not None}"
