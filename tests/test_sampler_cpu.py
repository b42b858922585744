"""CPU: the sampler oracle (oracle/sampler.py).  Its Philox4x32-10 is pinned to the published generator's known-answer vectors
(Random123 kat_vectors, philox4x32 10 rounds); the accept / redraw rule is checked through the properties the in-tree sampler
guarantees (generate_neg_triples_fast, model/MSNEA_tools.py:59-90)."""
import numpy as np

from oracle import sampler as S


def test_philox_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, "408f276d 41c83b0e a20bc7c6 6d5451fd"),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for ctr, key, want in kat:
        out = S.philox_rounds(*[np.array([c], dtype=np.uint64) for c in ctr], *key)
        assert " ".join(f"{int(v):08x}" for v in out[0]) == want
    # the kernels' counter / key packing: (ctr_lo, ctr_hi, 0x5a17c0de, 0x9e3779b9) under key (seed_lo, seed_hi)
    seed, ctr = 0x1234567890abcdef, np.array([5, (7 << 32) | 9], dtype=np.uint64)
    a = S.philox4x32_10(seed, ctr)
    b = S.philox_rounds(ctr & np.uint64(0xffffffff), ctr >> np.uint64(32), [0x5a17c0de] * 2, [0x9e3779b9] * 2, seed & 0xffffffff, seed >> 32)
    assert np.array_equal(a, b) and a.dtype == np.uint32 and a.shape == (2, 4)


def _kg(seed, B, N, R):
    rng = np.random.default_rng(seed)
    h, r, t = rng.integers(0, N, B), rng.integers(0, R, B), rng.integers(0, N, B)
    extra = rng.integers(0, N, (4 * B, 3))
    keys = np.concatenate([(h * R + r) * N + t, (extra[:, 0] * R + extra[:, 1] % R) * N + extra[:, 2]]).astype(np.int64)
    return h, r, t, np.unique(keys)


def test_with_replacement_rule():
    N, R, B, K = 60, 4, 50, 32
    h, r, t, known = _kg(1, B, N, R)
    nh, nt = S.sample_negatives(11, 3, h, r, t, K, N, R, 0, known)
    assert np.array_equal(nh, np.repeat(h[:, None], K, 1)) and nt.min() >= 0 and nt.max() < N
    hit = np.isin(((h[:, None] * R + r[:, None]) * N + nt).astype(np.int64), known)
    assert hit.mean() < 0.002                               # a known triple survives only ten redraws in a row
    # head-or-tail: one coin per positive, the kept side is the positive's own
    nh2, nt2 = S.sample_negatives(11, 3, h, r, t, K, N, R, 1, known)
    heads = (nt2 == t[:, None]).all(1)
    assert 0.2 < heads.mean() < 0.8 and ((nh2 == h[:, None]).all(1) | heads).all()
    # the tick selects a fresh stream; no filter = first draw kept
    assert not np.array_equal(S.sample_negatives(11, 4, h, r, t, K, N, R, 0, known)[1], nt)
    first = S.sample_negatives(11, 3, h, r, t, K, N, R, 0, None)[1]
    assert (first == nt).mean() > 0.9


def test_distinct_rule():
    N, R, B, K = 500, 4, 40, 100                            # K > 64: two blocks of slots share the accepted set
    h, r, t, known = _kg(2, B, N, R)
    ent_list = np.random.default_rng(5).permutation(N)[:300].astype(np.int32)      # the KG's own entity list: a subset of the ids
    nh, nt = S.sample_negatives_distinct(7, None, h, r, t, K, N, R, 0, known, ent_list)
    assert np.isin(nt, ent_list).all()
    assert all(len(set(row.tolist())) == K for row in nt)   # random.sample: no repeats inside one positive
    assert not np.isin(((h[:, None] * R + r[:, None]) * N + nt).astype(np.int64), known).any()
    nh1, nt1 = S.sample_negatives_distinct(7, None, h, r, t, K, N, R, 1, known, None)
    for b in range(B):
        side = nh1[b] if (nt1[b] == t[b]).all() else nt1[b]
        assert len(set(side.tolist())) == K
