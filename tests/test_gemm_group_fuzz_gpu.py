"""GPU: seeded random sweep of snag_gemm_f32_group -- 2..8 problems of random shapes (round 5: the A operand's layout drawn per problem for the
64 x 128-tile kinds), random epilogues, with and
without a K split, requested splits (snag_gemm_desc.splits), deferred and immediate reduces -- against the SAME descriptors launched one
by one: the grouped launch must give the same bits."""
import ctypes
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(seed):
    from snag_amd import _lib
    from snag_amd._lib import call, ptr, stream
    lib = _lib.load()
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    mk = lambda r, c: torch.randn(r, c, generator=g).to(DEV)
    for _ in range(4):
        n = rng.randint(2, 8)
        akc0, bkc = rng.random() < 0.5, rng.random() < 0.5
        kind = rng.choice(["short", "short", "long", "tall", "mixed"])
        probs = []
        for i in range(n):
            # "mixed": the read-out layer's backward -- weight-gradient shapes (A row-contiguous, long K) and input-gradient shapes (A K-contiguous,
            # short K) in ONE group; "short" / "long" draw the A layout per problem half of the time too
            akc = akc0 if (kind == "tall" or rng.random() < 0.5) else (rng.random() < 0.5)
            if kind == "mixed":
                akc = i % 2 == 1
                bkc = False
            if kind == "short":          # the Mformer-sized products: many rows, K = 16 .. 400
                M, N, K = rng.choice([1, 63, 700, 4097, 9000]), rng.choice([1, 64, 100, 128, 130, 384]), rng.choice([1, 16, 100, 128, 256, 384])
            elif kind == "long":         # weight-gradient shapes: small output, long K -> K splits
                M, N, K = rng.choice([64, 128, 300, 384]), rng.choice([64, 128]), rng.choice([3000, 5001, 15000])
            elif kind == "mixed":
                M, N, K = (rng.choice([128, 256, 384]), 128, rng.choice([3000, 15000])) if not akc else (rng.choice([700, 4097, 15000]), 128, rng.choice([128, 256, 384]))
            else:                        # projection shapes: the eight-wave tile (first problem) and riders with a requested split
                M, N, K = (rng.choice([2048, 4000]) if i == 0 else rng.choice([300, 768, 1100, 2048])), rng.choice([64, 100, 128]), rng.choice([600, 1024, 3000])
            a, b = mk(*((M, K) if akc else (K, M))), mk(*((N, K) if bkc else (K, N)))
            kw = {}
            side = []
            if rng.random() < 0.4:
                bias = mk(1, N).reshape(N)
                side.append(bias)
                kw["bias"] = ptr(bias)
            if rng.random() < 0.4:
                kw["act"] = rng.choice([1, 2])
            if rng.random() < 0.3:
                r = mk(M, N)
                side.append(r)
                kw["add1"], kw["ld1"] = ptr(r), r.stride(0)
            if rng.random() < 0.2:
                kw["accumulate"] = 1
            if kind != "short" and rng.random() < 0.7:
                kw["splits"] = rng.randint(1, 9)
            if kind == "tall":
                kw.setdefault("splits", rng.randint(2, 6))
            defer = rng.random() < 0.5
            probs.append((M, N, K, a, b, kw, side, defer, akc))
        res = []
        for single in (True, False):
            outs = [torch.full((M, N), 0.25, device=DEV) for (M, N, K, *_rest) in probs]
            ws = [torch.empty(max(64, lib.snag_gemm_workspace_floats(M, N, K), (kw.get("splits", 0) + 1) * M * N), device=DEV) for (M, N, K, a, b, kw, s, d, _k) in probs]
            descs = []
            for (M, N, K, a, b, kw, side, defer, akc), o, w in zip(probs, outs, ws):
                d = _lib.GemmDesc()
                d.M, d.N, d.K, d.a_kcontig, d.b_kcontig = M, N, K, int(akc), int(bkc)
                d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.alpha = ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(o), o.stride(0), 1.0
                d.workspace, d.workspace_floats = ptr(w), w.numel()
                for k_, v_ in kw.items():
                    setattr(d, k_, v_)
                d.defer_reduce = int(defer and not single)
                descs.append(d)
            if single:
                for d in descs:
                    call("snag_gemm_f32", ctypes.byref(d), stream())
            else:
                arr = (ctypes.c_void_p * n)(*[ctypes.addressof(d) for d in descs])
                call("snag_gemm_f32_group", arr, n, stream())
                deferred = [d for d in descs if d.defer_reduce]
                for i in range(0, len(deferred), 4):
                    part = deferred[i:i + 4]
                    arr2 = (ctypes.c_void_p * len(part))(*[ctypes.addressof(d) for d in part])
                    call("snag_gemm_reduce_group", arr2, len(part), stream())
            res.append(outs)
        for i, (x, y) in enumerate(zip(*res)):
            assert torch.equal(x, y), f"seed {seed}: problem {i} of {[(p[0], p[1], p[2], p[5], p[8]) for p in probs]} (b_kc={bkc}) differs"
        M, N, K, a, b, kw, side, defer, akc = probs[0]
        if not kw.get("act") and not kw.get("accumulate"):
            ref = (a.double() if akc else a.double().t()) @ (b.double().t() if bkc else b.double())
            if "bias" in kw:
                ref = ref + side[0].double()
            if "add1" in kw:
                ref = ref + side[-1].double()
            assert (res[1][0].double() - ref).abs().max().item() <= 2e-6 * K + 2e-5 * ref.abs().max().item() + 1e-5


@pytest.mark.parametrize("seed", range(16))
def test_gemm_group_random_cases(seed):
    _run(seed)
