"""Similarity evaluation on the GPU behind the reference's function names.

``pairwise_distances(x, y=None)`` and ``csls_sim(sim_mat, k)`` keep the signatures of
src/utils.py:202 and :417 (imported by name into main.py:22); ``rank_metrics`` fuses what
``Runner._test`` (main.py:376-436) does after ``joint_emb_generat``: distance -> CSLS -> rank of the true
match per row and per column -> Hits@k / MR / MRR, without the 2 x 10,500 ``torch.sort`` + ``.item()``
round trips; ``mutual_nearest`` is the distance part of ``SNAG.Iter_new_links`` (model/SNAG.py:192-208).
"""
import torch

from . import _lib, ops
from ._lib import call, ptr, stream

f32 = torch.float32


def _sqnorm(x):
    out = torch.empty(x.shape[0], device=x.device, dtype=f32)
    call("snag_row_sqnorm", ptr(x), x.shape[0], x.shape[1], x.stride(0), ptr(out), stream())
    return out


def pairwise_distances(x, y=None, as_similarity=False, _padded=False):
    """dist[i,j] = clamp(|x_i|^2 + |y_j|^2 - 2 x_i.y_j, 0)  (src/utils.py:202-218).

    The public form returns a contiguous [n, m] tensor like the reference; the internal callers (rank_metrics, mutual_nearest) ask
    for a row pitch padded to whole 128-byte lines (``_padded``: a strided view), which the GEMM writes ~10 % faster when m * 4 is
    not a multiple of 128 (10,500 columns are not)."""
    ops._rows(x, "x")
    if y is None:
        y = x
    ops._rows(y, "y")
    if x.shape[1] != y.shape[1]:
        raise ValueError(f"pairwise_distances: feature sizes differ ({x.shape[1]} vs {y.shape[1]})")
    n, m, d = x.shape[0], y.shape[0], x.shape[1]
    out = torch.empty(n, (m + 31) // 32 * 32, device=x.device, dtype=f32)[:, :m] if _padded else torch.empty(n, m, device=x.device, dtype=f32)
    xn = _sqnorm(x)
    yn = xn if y is x else _sqnorm(y)
    call("snag_pairwise_dist", ptr(x), x.stride(0), n, ptr(y), y.stride(0), m, d, ptr(xn), ptr(yn), ptr(out), out.stride(0),
         int(as_similarity), stream())
    return out


def cityblock_distances(x, y=None, _padded=False):
    """dist[i,j] = sum_k |x_i[k] - y_j[k]|: what ``--distance 1`` computes with scipy's cdist on the CPU (main.py:388-391)."""
    ops._rows(x, "x")
    if y is None:
        y = x
    ops._rows(y, "y")
    if x.shape[1] != y.shape[1]:
        raise ValueError(f"cityblock_distances: feature sizes differ ({x.shape[1]} vs {y.shape[1]})")
    n, m, d = x.shape[0], y.shape[0], x.shape[1]
    out = torch.empty(n, (m + 31) // 32 * 32, device=x.device, dtype=f32)[:, :m] if _padded else torch.empty(n, m, device=x.device, dtype=f32)
    call("snag_cityblock_dist", ptr(x), x.stride(0), n, ptr(y), y.stride(0), m, d, ptr(out), out.stride(0), stream())
    return out


def _neighbourhood(mat, k, from_dist):
    n, m = mat.shape
    r1 = torch.empty(n, device=mat.device, dtype=f32)
    r2 = torch.empty(m, device=mat.device, dtype=f32)
    work = torch.empty(_lib.load().snag_eval_work_floats(n, m), device=mat.device, dtype=f32)
    call("snag_csls_neighbourhood", ptr(mat), mat.stride(0), n, m, int(k), int(from_dist), ptr(r1), ptr(r2), ptr(work), stream())
    return r1, r2


def csls_sim(sim_mat, k):
    """csls[i,j] = 2 S[i,j] - mean(topk_j S[i,:]) - mean(topk_i S[:,j])  (src/utils.py:417-435)."""
    ops._rows(sim_mat, "sim_mat")
    n, m = sim_mat.shape
    r1, r2 = _neighbourhood(sim_mat, k, False)
    out = torch.empty(n, m, device=sim_mat.device, dtype=f32)
    call("snag_csls_apply", ptr(sim_mat), sim_mat.stride(0), n, m, ptr(r1), ptr(r2), ptr(out), stream())
    return out


def rank_metrics(emb_left, emb_right, csls_k=0, top_k=(1, 10, 50), distance=2):
    """Runner._test after the gather/normalise: ranks, Hits@k (rounded to 4 dp), MR, MRR in both directions.

    emb_left[i] and emb_right[i] are the i-th aligned test pair (``final_emb[test_left]``, ``final_emb[test_right]``,
    main.py:386).  csls_k = 0 disables CSLS (``--csls`` off); ``distance`` is ``--distance`` (2 = squared Euclidean, 1 = cityblock).
    """
    if distance not in (1, 2):
        raise ValueError("distance must be 1 (cityblock) or 2 (squared Euclidean), as --distance")
    dist = pairwise_distances(emb_left, emb_right, _padded=True) if distance == 2 else cityblock_distances(emb_left, emb_right, _padded=True)
    n, m = dist.shape
    if n != m:
        raise ValueError("rank_metrics: left and right must hold the same number of aligned rows")
    if n == 0:          # (the reference divides by len(test_left), main.py:435: Hits over nothing are undefined)
        raise ValueError("rank_metrics: no aligned test pairs")
    dev = dist.device
    r1 = r2 = None
    if csls_k and csls_k > 0:
        r1, r2 = _neighbourhood(dist, csls_k, True)
    rl = torch.empty(n, device=dev, dtype=torch.int32)
    rr = torch.empty(m, device=dev, dtype=torch.int32)
    call("snag_rank_diagonal", ptr(dist), dist.stride(0), n, m, ptr(r1), ptr(r2), ptr(rl), ptr(rr), stream())
    tk = torch.tensor(list(top_k), device=dev, dtype=torch.int32)
    out = {"rank_l2r": rl, "rank_r2l": rr, "dist": dist}
    nt = len(top_k) + 2
    summ = torch.empty(2 * nt, device=dev, dtype=torch.float64)              # both directions: one launch, one copy to the host
    call("snag_rank_summary_pair", ptr(rl), ptr(rr), n, ptr(tk), len(top_k), ptr(summ), stream())
    both = summ.cpu().tolist()
    for d, name in enumerate(("l2r", "r2l")):
        s = both[d * nt:(d + 1) * nt]
        out[f"hits_{name}"] = [round(c / n, 4) for c in s[:len(top_k)]]          # main.py:435-436
        out[f"mr_{name}"] = s[len(top_k)] / n
        out[f"mrr_{name}"] = s[len(top_k) + 1] / n
    return out


def rank_metrics_sharded(emb_left, emb_right, csls_k=0, top_k=(1, 10, 50), rank=None, world=None, group=None):
    """``rank_metrics`` with the rows of the left side sharded over the ranks of a process group (SURVEY.md par. 8e, config 4): every
    rank holds both embedding tables (they are the model's output), computes the distance block of ITS n/G left rows against all right
    rows -- 1/G of the similarity GEMM -- and three small exchanges finish the job: an all-gather of the per-column top-k candidates
    ([G, k, m] floats) for the CSLS column neighbourhoods, an all-gather of the adjusted diagonal (m floats) and an all-reduce of the
    per-column counts (m integers).  Ranks and summaries equal ``rank_metrics`` on one rank exactly.  Squared-Euclidean distance only.

    Returns the same dict on every rank (without ``dist``: no rank holds the whole matrix)."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    n, m = emb_left.shape[0], emb_right.shape[0]
    if n != m:
        raise ValueError("rank_metrics_sharded: left and right must hold the same number of aligned rows")
    dev = emb_left.device
    per = -(-n // world)
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    host = dist.get_backend(group) == "gloo"

    def gather(t):                  # [per, ...] on every rank -> [world * per, ...]
        t = t.contiguous()
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device="cpu" if host else dev)
        dist.all_gather_into_tensor(out, t.cpu() if host else t, group=group)
        return out.to(dev)

    nl = hi - lo
    k = int(csls_k) if csls_k and csls_k > 0 else 0
    if k:
        # decided from (n, world, k), which every rank knows: all ranks raise together, before any collective (a rank that raised
        # alone would leave the others waiting in the all-gather)
        short = [g for g in range(world) if 0 < min(n, (g + 1) * per) - min(n, g * per) < k]
        if short:
            raise ValueError(f"rank_metrics_sharded: every non-empty shard needs at least csls_k = {k} rows; with n = {n} over {world} ranks "
                             f"rank(s) {short} would hold fewer")
    r1 = r2 = None
    rl_loc = torch.zeros(per, device=dev, dtype=torch.int32)
    diag_loc = torch.zeros(per, device=dev, dtype=f32)
    cand = torch.full((max(k, 1), m), float("-inf"), device=dev, dtype=f32)
    block = None
    if nl > 0:
        block = pairwise_distances(emb_left[lo:hi].contiguous(), emb_right, _padded=True)
        if k:
            r1, _ = _neighbourhood(block, k, True)              # row neighbourhoods: complete (a row sees every column); the local r2 is not used
            work = torch.empty(_lib.load().snag_eval_work_floats(nl, m), device=dev, dtype=f32)
            call("snag_col_topk", ptr(block), block.stride(0), nl, m, k, 1, ptr(cand), ptr(work), stream())
    if k:
        allc = gather(cand)                                     # [world * k, m]: world chunks of k descending candidates
        r2 = torch.empty(m, device=dev, dtype=f32)
        call("snag_col_topk_mean", ptr(allc), world, m, k, ptr(r2), stream())
    if nl > 0:
        call("snag_rank_rows_block", ptr(block), block.stride(0), nl, m, lo, ptr(r1), ptr(r2), ptr(rl_loc), ptr(diag_loc), stream())
    diag = gather(diag_loc)[:n].contiguous()
    cnt = torch.zeros(m, device=dev, dtype=torch.int32)
    if nl > 0:
        call("snag_rank_cols_block", ptr(block), block.stride(0), nl, m, ptr(r1), ptr(r2), ptr(diag), ptr(cnt), stream())
    if host:
        c = cnt.cpu()
        dist.all_reduce(c, group=group)
        rr = c.to(dev)
    else:
        dist.all_reduce(cnt, group=group)
        rr = cnt
    rl = gather(rl_loc)[:n].contiguous()
    tk = torch.tensor(list(top_k), device=dev, dtype=torch.int32)
    out = {"rank_l2r": rl, "rank_r2l": rr}
    for name, rk in (("l2r", rl), ("r2l", rr)):
        summ = torch.empty(len(top_k) + 2, device=dev, dtype=torch.float64)
        call("snag_rank_summary", ptr(rk), n, ptr(tk), len(top_k), ptr(summ), stream())
        sv = summ.cpu().tolist()
        out[f"hits_{name}"] = [round(c / n, 4) for c in sv[:len(top_k)]]
        out[f"mr_{name}"] = sv[len(top_k)] / n
        out[f"mrr_{name}"] = sv[len(top_k) + 1] / n
    return out


def mutual_nearest(final_emb, left, right, chunk=None):
    """Row / column arg-min of the left x right distance matrix and the mutual pairs (model/SNAG.py:195-204).

    ``left`` / ``right`` are index tensors or lists into ``final_emb``.  The reference builds the matrix in 1000-row
    chunks only to bound memory; one launch covers it here.  Returns (preds_l, preds_r, pairs) with ``pairs`` the
    (left_id, right_id) list of mutual nearest neighbours.
    """
    dev = final_emb.device
    li = torch.as_tensor(left, device=dev, dtype=torch.int64)
    ri = torch.as_tensor(right, device=dev, dtype=torch.int64)
    dist = pairwise_distances(final_emb[li].contiguous(), final_emb[ri].contiguous(), _padded=True)
    n, m = dist.shape
    pl = torch.empty(n, device=dev, dtype=torch.int32)
    pr = torch.empty(m, device=dev, dtype=torch.int32)
    work = torch.empty(_lib.load().snag_eval_work_floats(n, m), device=dev, dtype=f32)
    call("snag_argmin_rows_cols", ptr(dist), dist.stride(0), n, m, ptr(pl), ptr(pr), ptr(work), stream())
    plc, prc = pl.cpu().tolist(), pr.cpu().tolist()
    lic, ric = li.cpu().tolist(), ri.cpu().tolist()
    pairs = [(lic[i], ric[p]) for i, p in enumerate(plc) if prc[p] == i]
    return pl, pr, pairs


def link_prediction(ent, rel, h, r, t, filter_mask=None, top_k=(1, 10), method="exact", known_sorted=None):
    """MKGC tail prediction: rank of the true tail among ALL entities under s(h,r,x) (lower = better), optional filter.

    PARITY UNPINNED (no MKGC eval code in the reference tree): the score is the in-tree TransE-L2^2 of
    model/MSNEA.py:110-121, the rank is count-less-than as in ``rank_metrics``.  Returns ranks [B] (int32) and Hits@k/MR/MRR.

    ``method="exact"`` scores every candidate with the training scorer's difference form sum_k (q_k - t_k)^2 (bit-identical to the
    scores the model was trained on; 29 G scores/s).  ``method="gemm"`` expands it to |q|^2 + |t|^2 - 2 q.t and runs the B x N x D
    product on the matrix cores (fp32 in, fp32 accumulate, split-bf16 products) through the similarity-eval GEMM (about 8x faster at 16k queries x 15k entities); the
    scores differ from the exact ones by fp32 cancellation error (~1e-6), so ranks can move by one place on near-ties.
    ``known_sorted`` (the sampler's sorted key table, ``ops.triple_keys``) builds the filtered protocol's mask on the device when no
    ``filter_mask`` is given.
    """
    ops._rows(ent, "ent"), ops._chk(rel, "rel", ndim=2)
    B, N, dev = h.shape[0], ent.shape[0], ent.device
    if B == 0:
        raise ValueError("link_prediction: no query triples")
    ib = 4 if h.dtype == torch.int32 else 8
    if method == "exact":
        pos = torch.empty(B, device=dev, dtype=f32)
        scores = torch.empty(B, N, device=dev, dtype=f32)
        call("snag_score_margin_fwd", ptr(ent), ent.stride(0), N, ptr(rel), rel.shape[0], ent.shape[1], ptr(h), ptr(r), ptr(t), None, ib,
             B, N, 0.0, ptr(pos), ptr(scores), None, stream())
    elif method == "gemm":
        d = ent.shape[1]
        st = stream()

        def rows(table, idx, n_idx):          # L2-normalised gather (F.normalize then index), the scorer's own row treatment
            z = torch.empty(n_idx, d, device=dev, dtype=f32)
            den = torch.empty(n_idx, device=dev, dtype=f32)
            call("snag_gather_normalize_fwd", ptr(table), table.stride(0), table.shape[0], d, ptr(idx), 4 if idx.dtype == torch.int32 else 8, n_idx, 1,
                 ptr(z), ptr(den), st)
            return z
        q = rows(ent, h, B) + rows(rel, r, B)
        cand = rows(ent, torch.arange(N, device=dev, dtype=torch.int32), N)
        scores = pairwise_distances(q, cand, _padded=True)
    else:
        raise ValueError("method must be 'exact' or 'gemm'")
    if filter_mask is None and known_sorted is not None:
        filter_mask = ops.build_filter_mask(h.to(torch.int32), r.to(torch.int32), t.to(torch.int32), N, rel.shape[0], known_sorted)
    if filter_mask is not None:
        ops._chk(filter_mask, "filter_mask", dtype=torch.uint8, ndim=2)
        if tuple(filter_mask.shape) != (B, N):
            raise ValueError("filter_mask must be [B, n_ent]")
    if scores.stride(0) != N:                  # snag_rank_full_table reads dense rows
        scores = scores.contiguous()
    rank = torch.empty(B, device=dev, dtype=torch.int32)
    call("snag_rank_full_table", ptr(scores), B, N, ptr(t), ib, ptr(filter_mask), ptr(rank), stream())
    tk = torch.tensor(list(top_k), device=dev, dtype=torch.int32)
    summ = torch.empty(len(top_k) + 2, device=dev, dtype=torch.float64)
    call("snag_rank_summary", ptr(rank), B, ptr(tk), len(top_k), ptr(summ), stream())
    s = summ.cpu().tolist()
    return {"rank": rank, "scores": scores, "hits": [round(c / B, 4) for c in s[:len(top_k)]], "mr": s[len(top_k)] / B,
            "mrr": s[len(top_k) + 1] / B}
