"""GPU parity, bit-exact (integer work): snag_sample_negatives and snag_sample_negatives_distinct through the C ABI vs the CPU
restatement of the same Philox4x32-10 stream and accept / redraw rule (oracle/sampler.py)."""
import numpy as np
import pytest
import torch

from oracle import sampler as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _kg(seed, B, N, R):
    rng = np.random.default_rng(seed)
    h, r, t = rng.integers(0, N, B), rng.integers(0, R, B), rng.integers(0, N, B)
    extra = rng.integers(0, N, (6 * B, 3))
    keys = np.concatenate([(h * R + r) * N + t, (extra[:, 0] * R + extra[:, 1] % R) * N + extra[:, 2]]).astype(np.int64)
    return h, r, t, np.unique(keys)


def _dev(h, r, t):
    return [torch.from_numpy(x.astype(np.int32)).to(DEV) for x in (h, r, t)]


@pytest.mark.parametrize("N,R,B,K,mode,filt,tick", [
    (15000, 558, 1024, 32, 0, True, 7), (60, 4, 50, 32, 0, True, None), (60, 4, 50, 5, 1, True, 3), (300, 7, 33, 130, 1, False, 0),
    (40, 3, 9, 1, 0, True, 1), ((1 << 20) + 3, 11, 17, 64, 0, True, 123456789),
    # >= 8,192 known triples: the workgroup's coarse key index in LDS narrows the search (dense key space: many rejections)
    (15000, 558, 4096, 32, 0, True, 5), (200, 5, 3000, 8, 1, True, 2), (120, 3, 4000, 16, 0, True, None)])
def test_with_replacement_sampler_equals_oracle(N, R, B, K, mode, filt, tick):
    from snag_amd import ops
    h, r, t, known = _kg(N + K, B, N, R)
    hd, rd, td = _dev(h, r, t)
    kd = torch.from_numpy(known).to(DEV) if filt else None
    tk = torch.tensor([tick], device=DEV, dtype=torch.int64) if tick is not None else None
    seed = 0x9e3779b97f4a7c15 ^ (N * 1315423911)
    got = ops.sample_negatives(hd, rd, td, K, N, R, seed, known_sorted=kd, head_or_tail=bool(mode), tick=tk)
    ref_h, ref_t = S.sample_negatives(seed, tick, h, r, t, K, N, R, mode, known if filt else None)
    if mode:
        assert np.array_equal(got[0].cpu().numpy(), ref_h) and np.array_equal(got[1].cpu().numpy(), ref_t)
    else:
        assert np.array_equal(got.cpu().numpy(), ref_t)


@pytest.mark.parametrize("N,R,B,K,mode,filt,sub,tick", [
    (15000, 558, 257, 32, 0, True, None, 5), (500, 4, 40, 100, 0, True, 300, None), (500, 4, 40, 100, 1, True, None, 2),
    (50, 3, 21, 40, 0, True, None, 0), (64, 2, 6, 64, 0, False, 64, 9), (3000, 9, 11, 1000, 0, True, 2000, 1), (70, 3, 5, 1, 1, True, 10, 4)])
def test_distinct_sampler_equals_oracle(N, R, B, K, mode, filt, sub, tick):
    """(50, .., K=40): nearly the whole list per positive -- rounds run out and the unfiltered last round kicks in; (64, K=64): a
    full permutation of the list; K=1000: sixteen blocks of slots sharing one accepted set."""
    from snag_amd import ops
    h, r, t, known = _kg(N * 3 + K, B, N, R)
    hd, rd, td = _dev(h, r, t)
    kd = torch.from_numpy(known).to(DEV) if filt else None
    tk = torch.tensor([tick], device=DEV, dtype=torch.int64) if tick is not None else None
    ent_list = np.random.default_rng(N).permutation(N)[:sub].astype(np.int32) if sub else None
    ld = torch.from_numpy(ent_list).to(DEV) if sub else None
    got = ops.sample_negatives(hd, rd, td, K, N, R, 77, known_sorted=kd, head_or_tail=bool(mode), tick=tk, distinct=True, ent_list=ld)
    ref_h, ref_t = S.sample_negatives_distinct(77, tick, h, r, t, K, N, R, mode, known if filt else None, ent_list)
    if mode:
        assert np.array_equal(got[0].cpu().numpy(), ref_h) and np.array_equal(got[1].cpu().numpy(), ref_t)
    else:
        assert np.array_equal(got.cpu().numpy(), ref_t)
        if sub:
            assert np.isin(ref_t, ent_list).all()


def test_distinct_sampler_refuses_impossible_requests():
    from snag_amd import ops, _lib
    i = torch.zeros(4, device=DEV, dtype=torch.int32)
    with pytest.raises(_lib.SnagHipError):
        ops.sample_negatives(i, i, i, 20, 10, 2, 1, distinct=True)            # 20 distinct out of 10 (random.sample raises too)
    with pytest.raises(_lib.SnagHipError):
        ops.sample_negatives(i, i, i, 2000, 5000, 2, 1, distinct=True)        # more slots than the kernel's per-positive LDS list
