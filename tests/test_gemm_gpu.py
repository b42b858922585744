"""GPU numerics: snag_gemm_f32 (fp32 via split-bf16 MFMA) vs a plain PyTorch fp32/fp64 reference of the same op."""
import math
import os

import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(a, b, a_kc, b_kc):
    A = a.double() if a_kc else a.double().t()
    B = b.double() if b_kc else b.double().t()
    return A @ B.t()


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (45000, 384, 128), (257, 300, 300), (1000, 128, 4096),
                                   (64, 128, 16), (5, 3, 7), (129, 130, 33), (300, 400, 1200), (96, 128, 1)])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_layouts(M, N, K, a_kc, b_kc):
    from snag_amd import ops
    if M * K > 20_000_000 and not (a_kc and b_kc):
        pytest.skip("large case only in the NT layout")
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((M, K) if a_kc else (K, M), generator=g)
    b = torch.randn((N, K) if b_kc else (K, N), generator=g)
    c = ops.gemm(a.to(DEV), b.to(DEV), a_kcontig=a_kc, b_kcontig=b_kc)
    ref = _ref(a, b, a_kc, b_kc)
    # fp32 accumulation error bound ~ K * eps * |a||b| ; compare against fp64
    assert_close(c, ref, atol=2e-6 * K + 1e-5, rtol=2e-5, what=f"gemm {M}x{N}x{K}")


def test_gemm_exact_integers_asymmetric():
    """Small-integer operands are exact in fp32: catches any row/col or k-mapping swap bit-for-bit."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-4, 5, (200, 70), generator=g).float()
    b = torch.randint(-4, 5, (150, 70), generator=g).float()
    c = ops.gemm(a.to(DEV), b.to(DEV))
    assert torch.equal(c.cpu(), a @ b.t())
    c2 = ops.gemm(a.t().contiguous().to(DEV), b.t().contiguous().to(DEV), a_kcontig=False, b_kcontig=False)
    assert torch.equal(c2.cpu(), a @ b.t())


def test_gemm_epilogue_chain():
    from snag_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N, K = 333, 128, 200
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, r1, r2 = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    keep = (torch.rand(M, N, generator=g) >= 0.1).to(torch.uint8)
    for act, fn in (("gelu", lambda x: 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))), ("relu", torch.relu), (None, lambda x: x)):
        c, pre = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV), alpha=0.5, act=act, want_pre_act=True,
                          keep_mask=keep.to(DEV), keep_p=0.1, add1=r1.to(DEV), add2=r2.to(DEV))
        u = 0.5 * (a.double() @ b.double().t()) + bias.double()
        ref = fn(u) * keep.double() / 0.9 + r1.double() + r2.double()
        assert_close(pre, u, atol=1e-5, what="pre_act")
        assert_close(c, ref, atol=2e-5, what=f"epilogue {act}")
    # accumulate + strided output (writes one modality slot of a stacked [M, 3, N] tensor)
    stacked = torch.zeros(M, 3, N, device=DEV)
    ops.gemm(a.to(DEV), b.to(DEV), out=stacked[:, 1, :])
    ops.gemm(a.to(DEV), b.to(DEV), out=stacked[:, 1, :], accumulate=True)
    assert_close(stacked[:, 1, :], 2 * (a.double() @ b.double().t()), atol=5e-5)
    assert float(stacked[:, 0, :].abs().max()) == 0 and float(stacked[:, 2, :].abs().max()) == 0


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (False, False)])
@pytest.mark.parametrize("splitk", [True, False])
def test_gemm_k_zero_applies_epilogue_only(a_kc, b_kc, splitk):
    """K = 0 is legal (gemm_check admits it): C = epilogue(0), so bias / add1 / accumulate still apply.  The dispatcher's split
    arithmetic once divided by zero here (host SIGFPE)."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(2)
    M, N = 70, 132
    a = torch.empty((M, 0) if a_kc else (0, M), device=DEV)
    b = torch.empty((N, 0) if b_kc else (0, N), device=DEV)
    bias, r1 = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    c = ops.gemm(a, b, a_kcontig=a_kc, b_kcontig=b_kc, bias=bias, add1=r1, splitk=splitk)
    assert torch.equal(c, bias[None, :] + r1)
    out = torch.full((M, N), 2.0, device=DEV)
    ops.gemm(a, b, a_kcontig=a_kc, b_kcontig=b_kc, out=out, accumulate=True, splitk=splitk)
    assert torch.equal(out, torch.full((M, N), 2.0, device=DEV))


def test_gemm_splitk_long_k_deterministic():
    """dW = dY^T X shape: tiny output, long reduction; split-K slabs are summed in a fixed order."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(5)
    Kred, Nout, Kin = 45000, 128, 128
    dy, x = torch.randn(Kred, Nout, generator=g), torch.randn(Kred, Kin, generator=g)
    dyd, xd = dy.to(DEV), x.to(DEV)
    c1 = ops.gemm(dyd, xd, a_kcontig=False, b_kcontig=False)
    c2 = ops.gemm(dyd, xd, a_kcontig=False, b_kcontig=False)
    assert torch.equal(c1, c2)
    assert_close(c1, dy.double().t() @ x.double(), atol=8e-3, rtol=1e-5)
    c3 = ops.gemm(dyd, xd, a_kcontig=False, b_kcontig=False, splitk=False)
    assert_close(c3, c1, atol=3e-2)      # one 45,000-long fp32 chain vs 86 short ones


@pytest.mark.parametrize("n,k,pad", [(7, 3, 0), (129, 16, 0), (300, 37, 3), (301, 300, 0), (1000, 300, 4), (64, 0, 0)])
def test_gram_is_symmetric_and_matches_gemm(n, k, pad):
    """snag_gram_f32: upper-triangle tiles + transposed stores.  S == S^T bit for bit (inside the diagonal tiles too: only their upper
    triangle is stored, and mirrored -- with split-bf16 products a.b and b.a differ in their last bits); the plain GEMM's values up to
    f32 summation order."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(n * 31 + k)
    zfull = torch.randn(n, k + pad, generator=g).to(DEV)
    z = zfull[:, :k] if k else torch.empty(n, 0, device=DEV)
    s = ops.gram(z, alpha=0.25)
    assert torch.equal(s, s.t().contiguous())
    if k:
        assert_close(s, 0.25 * (z.double().cpu() @ z.double().cpu().t()), atol=1e-5 * max(1.0, k ** 0.5))
        full = ops.gemm(z, z, alpha=0.25)
        scale = 0.25 * (z.abs() @ z.abs().t())                      # sum of the products' magnitudes: the scale of f32 accumulation error
        assert float(((s - full).abs() / scale.clamp_min(1e-30)).max()) <= 2e-6
    else:
        assert float(s.abs().max()) == 0.0


def _desc(a, b, out, a_kc, b_kc, ws=None, **kw):
    from snag_amd import _lib
    from snag_amd._lib import ptr
    M, K = (a.shape if a_kc else (a.shape[1], a.shape[0]))
    N = b.shape[0] if b_kc else b.shape[1]
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = M, N, K, int(a_kc), int(b_kc)
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.alpha = ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), 1.0
    for k, v in kw.items():
        setattr(d, k, v)
    if ws is not None:
        d.workspace, d.workspace_floats = ptr(ws), ws.numel()
    return d


@pytest.mark.parametrize("shape,a_kc,b_kc,split", [((15000, 128, 128), True, True, False), ((15000, 128, 128), True, False, False),
                                                   ((128, 128, 15000), False, False, True), ((333, 100, 77), True, True, False)])
def test_gemm_pair_equals_two_single_launches(shape, a_kc, b_kc, split):
    """snag_gemm_f32_pair: two same-shaped problems in one launch give the bits of two snag_gemm_f32 calls (with and without
    split-K, with different epilogues per problem); different shapes: see test_gemm_pair_of_different_shapes_is_one_grouped_launch."""
    import ctypes
    from snag_amd import _lib
    from snag_amd._lib import call, ptr, stream
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    mk = lambda r, c: torch.randn(r, c, generator=g).to(DEV)
    A = [mk(*((M, K) if a_kc else (K, M))) for _ in range(2)]
    B = [mk(*((N, K) if b_kc else (K, N))) for _ in range(2)]
    bias = mk(1, N).reshape(N)
    nws = _lib.load().snag_gemm_workspace_floats(M, N, K) if split else 0
    ref, got = [], []
    for single in (True, False):
        outs = [torch.zeros(M, N, device=DEV) for _ in range(2)]
        ws = [torch.empty(max(nws, 1), device=DEV) for _ in range(2)] if split else [None, None]
        d0 = _desc(A[0], B[0], outs[0], a_kc, b_kc, ws[0], bias=ptr(bias), act=1)
        d1 = _desc(A[1], B[1], outs[1], a_kc, b_kc, ws[1])
        if single:
            call("snag_gemm_f32", ctypes.byref(d0), stream())
            call("snag_gemm_f32", ctypes.byref(d1), stream())
            ref = outs
        else:
            call("snag_gemm_f32_pair", ctypes.byref(d0), ctypes.byref(d1), stream())
            got = outs
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    ra = (A[1].double() if a_kc else A[1].double().t()) @ (B[1].double().t() if b_kc else B[1].double())
    assert_close(got[1], ra, atol=2e-6 * K + 1e-5, rtol=2e-5)
    # different shapes: still correct (grouped launch, or two launches when one side splits K)
    o2, o3 = torch.zeros(M, N, device=DEV), torch.zeros(64, 128, device=DEV)
    a3, b3 = mk(64, 32), mk(128, 32)
    call("snag_gemm_f32_pair", ctypes.byref(_desc(A[1], B[1], o2, a_kc, b_kc)), ctypes.byref(_desc(a3, b3, o3, True, True)), stream())
    assert_close(o2, ra, atol=2e-6 * K + 1e-5, rtol=2e-5)
    assert_close(o3, a3.double() @ b3.double().t(), atol=1e-4)


@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("s0,s1", [((4500, 256, 128), (1500, 128, 128)), ((257, 130, 100), (64, 3, 37)), ((70, 300, 41), (900, 128, 256))])
def test_gemm_pair_of_different_shapes_is_one_grouped_launch(s0, s1, a_kc, b_kc):
    """snag_gemm_f32_pair with two DIFFERENT shapes of the same layouts and no K split: one grouped launch over a linear tile index
    (the key/value + query projections of the read-out layer).  Bits equal to two snag_gemm_f32 calls, epilogues per problem, ragged
    sizes and K on both sides."""
    import ctypes
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(sum(s0) + 7 * sum(s1))
    mk = lambda r, c: torch.randn(r, c, generator=g).to(DEV)
    ops_ = []
    for (M, N, K) in (s0, s1):
        ops_.append((mk(*((M, K) if a_kc else (K, M))), mk(*((N, K) if b_kc else (K, N))), mk(1, N).reshape(N), mk(M, N)))
    res = []
    for single in (True, False):
        outs = [torch.zeros(s[0], s[1], device=DEV) for s in (s0, s1)]
        d0 = _desc(ops_[0][0], ops_[0][1], outs[0], a_kc, b_kc, bias=ptr(ops_[0][2]), act=1)
        d1 = _desc(ops_[1][0], ops_[1][1], outs[1], a_kc, b_kc, add1=ptr(ops_[1][3]), ld1=ops_[1][3].stride(0))
        if single:
            call("snag_gemm_f32", ctypes.byref(d0), stream())
            call("snag_gemm_f32", ctypes.byref(d1), stream())
        else:
            call("snag_gemm_f32_pair", ctypes.byref(d0), ctypes.byref(d1), stream())
        res.append(outs)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    a, b, _, r1 = ops_[1]
    ref = (a.double() if a_kc else a.double().t()) @ (b.double().t() if b_kc else b.double()) + r1.double()
    assert_close(res[1][1], ref, atol=2e-6 * s1[2] + 1e-5, rtol=2e-5)


@pytest.mark.parametrize("a_kc,b_kc,shapes", [
    (True, False, [(15000, 128, 384), (15000, 128, 256), (15000, 128, 256)]),                        # dX of the read-out token + the two modalities' dmid
    (False, False, [(128, 128, 15000), (384, 128, 15000), (256, 128, 15000), (256, 128, 15000)]),     # dWo, dWqkv, dWf x 2: K splits, deferred reduces
    (True, True, [(15000, 256, 128), (15000, 256, 128), (15000, 256, 128), (15000, 128, 128)]),       # key/value of three tokens + query
    (True, True, [(257, 130, 100), (64, 3, 37), (1, 1, 1), (900, 128, 256), (70, 300, 41), (33, 64, 17)]),
    (False, True, [(300, 128, 5000), (64, 64, 16)]),
])
def test_gemm_group_equals_single_launches(a_kc, b_kc, shapes):
    """snag_gemm_f32_group: up to six problems of different shapes (same layouts) in one launch, each with its own epilogue, K split and
    workspace, give the bits of separate snag_gemm_f32 calls; deferred reduces are finished by snag_gemm_reduce_group."""
    import ctypes
    from snag_amd import _lib
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(sum(sum(x) for x in shapes))
    mk = lambda r, c: torch.randn(r, c, generator=g).to(DEV)
    ops_ = [(mk(*((M, K) if a_kc else (K, M))), mk(*((N, K) if b_kc else (K, N))), mk(1, N).reshape(N), mk(M, N)) for (M, N, K) in shapes]
    lib = _lib.load()
    res = []
    for single in (True, False):
        outs = [torch.zeros(M, N, device=DEV) for (M, N, K) in shapes]
        ws = [torch.empty(max(1, lib.snag_gemm_workspace_floats(M, N, K)), device=DEV) for (M, N, K) in shapes]
        descs = []
        for i, (a, b, bias, r) in enumerate(ops_):
            kw = [dict(bias=ptr(bias), act=1), dict(add1=ptr(r), ld1=r.stride(0)), dict(accumulate=1), dict()][i % 4]
            if i % 4 == 2:
                outs[i].copy_(r)
            descs.append(_desc(a, b, outs[i], a_kc, b_kc, ws[i], defer_reduce=int(i < 4 and not single), **kw))
        if single:
            for d in descs:
                call("snag_gemm_f32", ctypes.byref(d), stream())
        else:
            arr = (ctypes.c_void_p * len(descs))(*[ctypes.addressof(d) for d in descs])
            call("snag_gemm_f32_group", arr, len(descs), stream())
            call("snag_gemm_reduce_group", arr, min(4, len(descs)), stream())
        res.append(outs)
    for x, y in zip(*res):
        assert torch.equal(x, y)
    a, b, _, r = ops_[1]
    ref = (a.double() if a_kc else a.double().t()) @ (b.double().t() if b_kc else b.double()) + r.double()
    assert_close(res[1][1], ref, atol=2e-6 * shapes[1][2] + 1e-5, rtol=2e-5)


@pytest.mark.parametrize("s0,s1", [((15000, 128, 4096), (15000, 128, 768)), ((4096, 128, 15000), (768, 128, 15000)), ((300, 128, 5000), (64, 64, 16)),
                                   ((130, 70, 4000), (257, 100, 3000))])
def test_gemm_deferred_reduce_pair(s0, s1):
    """defer_reduce + snag_gemm_reduce_pair: two K-split problems of different shape leave their slabs in (distinct) workspaces and ONE
    launch reduces both and applies their epilogues -- the bits of two plain snag_gemm_f32 calls.  (64, 64, 16): no split on one
    side -- finished by its own call, skipped by the pair reduce; N = 70 / 100: the scalar (non-quad) reduce path."""
    import ctypes
    from snag_amd import _lib
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(sum(s0) + 3 * sum(s1))
    mk = lambda r, c: torch.randn(r, c, generator=g).to(DEV)
    probs = []
    for (M, N, K) in (s0, s1):
        probs.append((mk(K, M), mk(K, N), mk(1, N).reshape(N)))          # both operands row-contiguous (the projection layouts)
    res = []
    for deferred in (False, True):
        outs = [torch.zeros(s[0], s[1], device=DEV) for s in (s0, s1)]
        ws = [torch.empty(max(_lib.load().snag_gemm_workspace_floats(*s), 1) + 64, device=DEV) for s in (s0, s1)]
        d = [_desc(probs[i][0], probs[i][1], outs[i], False, False, ws[i], bias=ptr(probs[i][2]), act=1 if i == 0 else 0) for i in range(2)]
        for x in d:
            x.defer_reduce = int(deferred)
            call("snag_gemm_f32", ctypes.byref(x), stream())
        if deferred:
            call("snag_gemm_reduce_pair", ctypes.byref(d[0]), ctypes.byref(d[1]), stream())
        res.append(outs)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    a, b, bias = probs[1]
    assert_close(res[1][1], a.double().t() @ b.double() + bias.double(), atol=2e-6 * s1[2] + 1e-5, rtol=2e-5)


@pytest.mark.parametrize("pair_shape,s2,s3", [((128, 128, 15000), (4096, 128, 15000), (768, 128, 15000)), ((128, 128, 15000), (64, 64, 16), (130, 70, 4000)),
                                              ((100, 60, 3000), (300, 128, 5000), (257, 100, 3000))])
def test_gemm_deferred_reduce_group(pair_shape, s2, s3):
    """snag_gemm_reduce_group: a same-shaped snag_gemm_f32_pair (both members deferred) and two more deferred problems leave four sets of
    slabs behind, ONE launch finishes them all -- the bits of the plain calls, accumulate and epilogues per problem (the step's four
    projection weight gradients; the Mformer backward's dWo / dWq / dWkv use the same launch inside the library)."""
    import ctypes
    from snag_amd import _lib
    from snag_amd._lib import call, ptr, stream
    shapes = [pair_shape, pair_shape, s2, s3]
    g = torch.Generator().manual_seed(sum(pair_shape) + 3 * sum(s2) + 5 * sum(s3))
    mk = lambda r, c: torch.randn(r, c, generator=g).to(DEV)
    probs = [(mk(K, M), mk(K, N), mk(1, N).reshape(N), mk(M, N)) for (M, N, K) in shapes]
    res = []
    for deferred in (False, True):
        outs = [probs[i][3].clone() for i in range(4)]          # accumulate = 1 on problem 1 and 3: C += the product
        ws = [torch.empty(max(_lib.load().snag_gemm_workspace_floats(*s), 1) + 64, device=DEV) for s in shapes]
        d = [_desc(probs[i][0], probs[i][1], outs[i], False, False, ws[i], bias=ptr(probs[i][2]) if i % 2 == 0 else None, act=1 if i == 2 else 0,
                   accumulate=i % 2, defer_reduce=int(deferred)) for i in range(4)]
        call("snag_gemm_f32_pair", ctypes.byref(d[0]), ctypes.byref(d[1]), stream())
        call("snag_gemm_f32", ctypes.byref(d[2]), stream())
        call("snag_gemm_f32", ctypes.byref(d[3]), stream())
        if deferred:
            group = (ctypes.POINTER(_lib.GemmDesc) * 4)(*[ctypes.pointer(x) for x in d])
            call("snag_gemm_reduce_group", group, 4, stream())
        res.append(outs)
    for i in range(4):
        assert torch.equal(res[0][i], res[1][i]), i
    a, b, _, c0 = probs[3]
    assert_close(res[1][3], a.double().t() @ b.double() + c0.double(), atol=2e-6 * s3[2] + 1e-5, rtol=2e-5)
    # more than four problems / a shared workspace are refused
    with pytest.raises(Exception):
        call("snag_gemm_reduce_group", group, 5, stream())


@pytest.mark.parametrize("M,N,K,dist", [(1500, 128, 4096, "randn"), (1500, 128, 4096, "uniform"), (1024, 256, 128, "randn"), (300, 128, 15000, "randn"),
                                        (1500, 128, 4096, "wide")])
def test_gemm_error_against_fp64_is_f32_grade(M, N, K, dist):
    """The split-bf16 product form (six bf16 MFMA products of exactly split operands) against an fp64 product: the error, relative to
    sum_k |a_k b_k| (the scale f32 accumulation itself errs on), stays at what an f32 fmaf chain of that length gives, and the
    rounding of the result alone (2^-24) bounds it from below.  'wide': operand magnitudes spread over 12 decades."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(M + K)
    if dist == "randn":
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    elif dist == "uniform":
        a, b = torch.rand(M, K, generator=g), torch.rand(N, K, generator=g)
    else:
        a = torch.randn(M, K, generator=g) * torch.pow(10.0, torch.rand(M, K, generator=g) * 12 - 6)
        b = torch.randn(N, K, generator=g) * torch.pow(10.0, torch.rand(N, K, generator=g) * 12 - 6)
    c = ops.gemm(a.to(DEV), b.to(DEV)).double().cpu()
    c64 = a.double() @ b.double().t()
    scale = a.double().abs() @ b.double().abs().t()
    err = (c - c64).abs() / scale
    assert float(err.max()) <= 1e-6 and float(err.pow(2).mean().sqrt()) <= 1.5e-7
    # and never worse than a sequential f32 accumulation of the same products would be allowed to be: K * 2^-24
    assert float(err.max()) <= K * 2.0 ** -24


@pytest.mark.parametrize("M,N,K", [(200, 128, 1000), (4096, 128, 1500), (64, 64, 8), (130, 70, 41), (512, 128, 15000)])
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_never_reads_past_its_operands(M, N, K, a_kc, b_kc):
    """K not a multiple of the 16-deep k-tile, operands embedded in larger buffers whose surroundings are poisoned (1e30, alternating
    signs): any read past row K / column K of an operand -- the staged loaders over-read by design and must discard it -- shows up
    as a huge error.  Split-K shapes (small output, long K) included: the last split's range ends ragged."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + 2 * a_kc + b_kc)

    def embedded(rows, cols):           # a [rows, cols] view with a 5-row / 4-column poisoned margin on every side
        big = torch.full((rows + 10, cols + 8), 1e30)
        big[::2] *= -1
        big[5:5 + rows, 4:4 + cols] = torch.randn(rows, cols, generator=g)
        big = big.to(DEV)
        return big, big[5:5 + rows, 4:4 + cols]
    _, a = embedded(*((M, K) if a_kc else (K, M)))
    _, b = embedded(*((N, K) if b_kc else (K, N)))
    c = ops.gemm(a, b, a_kcontig=a_kc, b_kcontig=b_kc)
    ref = _ref(a.cpu(), b.cpu(), a_kc, b_kc)
    assert_close(c, ref, atol=2e-6 * K + 1e-5, rtol=2e-5, what=f"gemm {M}x{N}x{K} inside poisoned buffers")


@pytest.mark.parametrize("sa,sb,a_kc,b_kc", [((4096, 6000), (4096, 128), False, False),      # eight-wave ping-pong tile (N <= 128, long K)
                                             ((3000, 2000), (2000, 300), True, False),       # four-wave row-private tile (several column tiles)
                                             ((9000, 128), (256, 128), True, True),          # 64 x 128 pre-split tile
                                             ((1501, 4099), (1501, 130), False, False)])     # ragged everything, K split
def test_gemm_repeated_launches_are_bit_identical(sa, sb, a_kc, b_kc):
    """Race screen for the staged k-loops (LDS images written and read across barriers, two groups of waves half an iteration apart in
    the ping-pong form): 40 launches of one problem give 40 bit-identical results (tools/stress_gemm_determinism.py runs the long form)."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(sa[0] + sb[1])
    a, b = torch.randn(*sa, generator=g).to(DEV), torch.randn(*sb, generator=g).to(DEV)
    ref = ops.gemm(a, b, a_kcontig=a_kc, b_kcontig=b_kc).clone()
    for _ in range(40):
        assert torch.equal(ops.gemm(a, b, a_kcontig=a_kc, b_kcontig=b_kc), ref)


@pytest.mark.parametrize("M,N,K,split", [(1500, 128, 128, False), (333, 100, 77, False), (200, 128, 5000, True), (64, 70, 900, True)])
def test_gemm_mirror_output(M, N, K, split):
    """desc.mirror: a second copy of the FINAL C values (after accumulate), rows scaled where mirror_rowmask[m] != 0 -- the way the
    entity-table gradient leaves the Mformer backward's last GEMM.  Through the plain epilogue and through the split-K reduce,
    dwordx4 and scalar (N = 70 / 100) paths."""
    import ctypes
    from snag_amd import _lib
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, K, generator=g).to(DEV), torch.randn(N, K, generator=g).to(DEV)
    c0 = torch.randn(M, N, generator=g).to(DEV)
    rowmask = (torch.rand(M, generator=g) < 0.4).to(torch.uint8).to(DEV)
    out, mirror = c0.clone(), torch.full((M, N + 4), 7.0, device=DEV)
    ws = torch.empty(max(_lib.load().snag_gemm_workspace_floats(M, N, K), 1) + 64, device=DEV) if split else None
    d = _desc(a, b, out, True, True, ws, accumulate=1, mirror=ptr(mirror), ldmirror=mirror.stride(0), mirror_rowmask=ptr(rowmask), mirror_scale=0.65)
    call("snag_gemm_f32", ctypes.byref(d), stream())
    ref = torch.zeros(M, N, device=DEV)
    call("snag_gemm_f32", ctypes.byref(_desc(a, b, ref, True, True, ws)), stream())
    assert torch.equal(out, c0 + ref)
    scale = torch.where(rowmask.bool(), torch.tensor(0.65, device=DEV), torch.tensor(1.0, device=DEV)).view(-1, 1)
    assert torch.equal(mirror[:, :N], out * scale) and float((mirror[:, N:] - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K,split", [(4096, 128, 15000, True), (300, 70, 41, False), (128, 128, 5000, True), (1000, 128, 128, False)])
def test_gemm_transposed_output(M, N, K, split):
    """c_transposed: value (m, n) lands at C[n, m] -- the bits of the plain call, transposed; bias / activation / accumulate still follow (m, n)."""
    from snag_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(K, M, generator=g).to(DEV)
    b = torch.randn(K, N, generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    plain = ops.gemm(a, b, a_kcontig=False, b_kcontig=False, bias=bias, act="relu", splitk=split)
    tr = ops.gemm(a, b, a_kcontig=False, b_kcontig=False, bias=bias, act="relu", splitk=split, c_transposed=True)
    assert tuple(tr.shape) == (N, M) and torch.equal(tr.t(), plain)
    acc = torch.ones(N, M, device=DEV)
    ops.gemm(a, b, a_kcontig=False, b_kcontig=False, splitk=split, c_transposed=True, out=acc, accumulate=True)
    assert_close(acc.t(), 1.0 + ops.gemm(a, b, a_kcontig=False, b_kcontig=False, splitk=split), atol=1e-5, rtol=1e-6)


def test_gemm_non_finite_and_extreme_operands_are_documented_behaviour():
    """The split-bf16 form (a = h + m + l, six products) vs an fp32 addmm on operands at the edges of fp32 (include/snag_hip.h, snag_gemm_f32):
    * finite values up to ~3e38 and down to the smallest normals multiply like fp32 (the split is exact, accumulation is fp32);
    * an infinite operand: the split gives (inf, NaN, NaN) -- a - bf16(a) is inf - inf --, so the affected outputs are NaN where fp32 gives
      +-inf: DOCUMENTED divergence (feature tables and weights are finite; a non-finite input still poisons its outputs visibly);
    * a NaN operand stays NaN in its row / column, as in fp32;
    * denormal operands: their lower planes flush, the product is exact to the top 8 bits of the denormal (absolute error < 2^-133 |b|)."""
    from snag_amd import ops
    M, N, K = 64, 128, 64
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    a[0, 0] = 3.0e38                                    # near FLT_MAX, times values below one: representable, no overflow in the split
    b[:, 0] = b[:, 0].clamp(-0.9, 0.9)
    a[1, :] = 0.0
    a[1, 3], b[5, 3] = 1.1754944e-38, 1.0               # the smallest normal
    a[2, 7] = float("inf")
    a[3, 9] = float("nan")
    a[4, :] = 0.0
    a[4, 2], b[9, 2] = 1.0e-40, 1.0                     # a denormal
    c = ops.gemm(a.to(DEV), b.to(DEV)).cpu()
    ref = a.double() @ b.double().t()
    fin = torch.ones(M, dtype=torch.bool)
    fin[[0, 2, 3]] = False                              # (row 0 carries the 3e38 value, rows 2 / 3 the inf / NaN)
    assert_close(c[fin], ref[fin], atol=1e-5, rtol=2e-5)
    assert torch.isfinite(c[0]).all() and torch.allclose(c[0].double(), ref[0], rtol=1e-5, atol=1e31)      # huge but finite, as in fp32
    assert c[1, 5] == 1.1754944e-38                                      # smallest normal: exact
    assert torch.isnan(c[2]).all() or torch.isinf(c[2]).any()            # inf operand: NaN (documented), never a silently finite value
    assert not torch.isfinite(c[2]).any()
    assert torch.isnan(c[3]).all()                                       # NaN operand: NaN, as in fp32
    assert abs(float(c[4, 9]) - 1.0e-40) < 9.2e-41                        # denormal: what bf16's own denormal range holds (absolute error < 2^-133 |b|)



@pytest.mark.parametrize("shapes,splits", [([(4096, 128, 15000), (768, 128, 15000)], 13), ([(2048, 128, 3000), (1100, 100, 3000), (300, 64, 3000)], 5)])
def test_gemm_group_of_long_k_problems_with_a_requested_split(shapes, splits):
    """snag_gemm_f32_group on the eight-wave 256 x 128 tile: long-K weight-gradient shapes (A and B row-contiguous, C stored transposed)
    with snag_gemm_desc.splits set by the caller -- one launch, deferred reduces -- give the bits of single launches with the same request,
    and the problem whose row count alone would pick another tile rides along."""
    import ctypes
    from snag_amd import _lib
    from snag_amd._lib import call, stream
    g = torch.Generator().manual_seed(sum(sum(x) for x in shapes) + splits)
    mk = lambda r, c: torch.randn(r, c, generator=g).to(DEV)
    ops_ = [(mk(K, M), mk(K, N)) for (M, N, K) in shapes]
    res = []
    for single in (True, False):
        outs = [torch.zeros(N, M, device=DEV) for (M, N, K) in shapes]           # transposed outputs
        ws = [torch.empty(splits * M * N + 64, device=DEV) for (M, N, K) in shapes]
        descs = [_desc(a, b, o, False, False, w, c_transposed=1, splits=splits, defer_reduce=int(not single)) for (a, b), o, w in zip(ops_, outs, ws)]
        for d, (M, N, K) in zip(descs, shapes):
            d.ldc = M
        if single:
            for d in descs:
                call("snag_gemm_f32", ctypes.byref(d), stream())
        else:
            arr = (ctypes.c_void_p * len(descs))(*[ctypes.addressof(d) for d in descs])
            call("snag_gemm_f32_group", arr, len(descs), stream())
            call("snag_gemm_reduce_group", arr, len(descs), stream())
        res.append(outs)
    for x, y in zip(*res):
        assert torch.equal(x, y)
    a, b = ops_[1]
    assert_close(res[1][1], (a.double().t() @ b.double()).t(), atol=2e-6 * shapes[1][2] + 1e-5, rtol=2e-5)
