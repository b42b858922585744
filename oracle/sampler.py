"""CPU restatement of the on-device negative samplers -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

What it follows: ``generate_neg_triples_fast`` (/root/reference/SNAG_MMEA/model/MSNEA_tools.py:59-90): per positive up to
``max_try = 10`` tries (:59,:66), candidates that form a known triple are dropped (:80), the last try is kept whatever it is
(:76-78); ``random.sample`` over the KG's own entity list gives distinct candidates (:64-65,:69,:72).

PARITY UNPINNED against the reference itself: its candidates come from python's ``random`` / ``np.random`` global streams, which
a counter-based device generator cannot reproduce.  What IS pinned, bit for bit, is the device kernels against this file: the
same Philox4x32-10 stream (snag_amd/csrc/philox.hip.h) and the same accept / redraw rule (misc.hip sample_negatives_kernel,
sampler.hip sample_distinct_kernel), so that every candidate the GPU draws is checkable.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
TICK_MUL = 0x9E3779B97F4A7C15
MASK64 = (1 << 64) - 1


def philox_rounds(c0, c1, c2, c3, k0: int, k1: int):
    """The ten Philox4x32 rounds on counter words c0..c3 (uint64 arrays holding 32-bit values) with key (k0, k1).
    Known-answer vectors of the published generator (Random123 kat_vectors) are checked in tests/test_sampler_cpu.py."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c1, c3, c0, c2 = p1 & M32, p0 & M32, n0, n2
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def philox4x32_10(seed: int, ctr):
    """snag_amd/csrc/philox.hip.h: counter words (ctr_lo, ctr_hi, 0x5a17c0de, 0x9e3779b9), key (seed_lo, seed_hi).
    -> uint32 [len(ctr), 4].  seed: python int (64 bit), ctr: array of uint64 counters."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    return philox_rounds(ctr & M32, ctr >> np.uint64(32), np.full_like(ctr, 0x5a17c0de), np.full_like(ctr, 0x9e3779b9),
                         seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def _effective_seed(seed, tick):
    return (int(seed) + (int(tick) * TICK_MUL if tick is not None else 0)) & MASK64


def _corrupt_head(seed, B, mode):
    if mode != 1:
        return np.zeros(B, dtype=bool)
    return (philox4x32_10(seed ^ 0xC0FFEE, np.arange(B, dtype=np.uint64))[:, 0] & 1) != 0


def _key(hh, rr, tt, n_ent, n_rel):
    return (np.int64(hh) * n_rel + np.int64(rr)) * n_ent + np.int64(tt)


def sample_negatives(seed, tick, h, r, t, K, n_ent, n_rel, mode=0, known_sorted=None):
    """misc.hip sample_negatives_kernel: slot (b, k) redraws up to 10 times while (h_b, r_b, cand) [or (cand, r_b, t_b)] is known;
    WITH replacement, candidates 0..n_ent-1.  -> (neg_h [B,K], neg_t [B,K]) int32."""
    h, r, t = (np.asarray(x, dtype=np.int64) for x in (h, r, t))
    B = len(h)
    seed = _effective_seed(seed, tick)
    ch = _corrupt_head(seed, B, mode)
    known = set(np.asarray(known_sorted).tolist()) if known_sorted is not None else None
    neg_h = np.empty((B, K), dtype=np.int32)
    neg_t = np.empty((B, K), dtype=np.int32)
    i = np.arange(B * K, dtype=np.uint64)
    draws = np.stack([philox4x32_10(seed, i * np.uint64(16) + np.uint64(a))[:, 0] for a in range(10)], axis=1)     # [B*K, 10]
    cands = ((draws.astype(np.uint64) * np.uint64(n_ent)) >> np.uint64(32)).astype(np.int64)
    for b in range(B):
        for k in range(K):
            cand = 0
            for a in range(10):
                cand = int(cands[b * K + k, a])
                if known is None or a == 9:
                    break
                key = _key(cand, r[b], t[b], n_ent, n_rel) if ch[b] else _key(h[b], r[b], cand, n_ent, n_rel)
                if int(key) not in known:
                    break
            neg_h[b, k] = cand if ch[b] else h[b]
            neg_t[b, k] = t[b] if ch[b] else cand
    return neg_h, neg_t


def sample_negatives_distinct(seed, tick, h, r, t, K, n_ent, n_rel, mode=0, known_sorted=None, ent_list=None):
    """sampler.hip sample_distinct_kernel: the negatives of one positive are distinct and come from ``ent_list``
    (random.sample over the KG's entity list, MSNEA_tools.py:64-72); rounds = tries; slots are handled in blocks of 64."""
    h, r, t = (np.asarray(x, dtype=np.int64) for x in (h, r, t))
    B = len(h)
    seed = _effective_seed(seed, tick)
    ch = _corrupt_head(seed, B, mode)
    known = set(np.asarray(known_sorted).tolist()) if known_sorted is not None else None
    lst = np.asarray(ent_list, dtype=np.int64) if ent_list is not None else None
    n_list = len(lst) if lst is not None else n_ent
    neg_h = np.empty((B, K), dtype=np.int32)
    neg_t = np.empty((B, K), dtype=np.int32)
    for b in range(B):
        vals = np.full(K, -1, dtype=np.int64)
        for c0 in range(0, K, 64):
            slots = np.arange(c0, min(K, c0 + 64))
            pending = np.ones(len(slots), dtype=bool)
            for a in range(10):
                if not pending.any():
                    break
                u = philox4x32_10(seed, (np.uint64(b * K) + slots.astype(np.uint64)) * np.uint64(16) + np.uint64(a))[:, 0]
                j = ((u.astype(np.uint64) * np.uint64(n_list)) >> np.uint64(32)).astype(np.int64)
                cand = lst[j] if lst is not None else j
                ok = pending.copy()
                if a != 9:
                    accepted_before = set(vals[vals >= 0].tolist())
                    taken_this_round = set()
                    for li in range(len(slots)):                          # ascending slot = ascending lane: the lowest slot keeps a value
                        if not ok[li]:
                            continue
                        c = int(cand[li])
                        if known is not None:
                            key = _key(c, r[b], t[b], n_ent, n_rel) if ch[b] else _key(h[b], r[b], c, n_ent, n_rel)
                            if int(key) in known:
                                ok[li] = False
                                continue
                        if c in accepted_before or c in taken_this_round:
                            ok[li] = False
                            continue
                        taken_this_round.add(c)
                vals[slots[ok]] = cand[ok]
                pending &= ~ok
        neg_h[b] = np.where(ch[b], vals, h[b])
        neg_t[b] = np.where(ch[b], t[b], vals)
    return neg_h, neg_t
