"""Negative-sampled margin scoring behind the reference's operator interface.

Reference: ``MSNEA.kge`` / ``r_rep`` (model/MSNEA.py:108-132,151-152) and the README-level MKGC
knobs ``MARGIN`` / ``NEG_NUM`` (README.md:71-79).  All arithmetic runs in libsnag_hip.so.
"""
import torch

from . import ops


class _MarginScoreFn(torch.autograd.Function):
    """(ent, rel) -> (pos [B], neg [B,K]); gradients via snag_score_margin_bwd (explicit mode)."""

    @staticmethod
    def forward(ctx, ent, rel, h, r, t, neg_t):
        pos, neg, _ = ops.score_margin_fwd(ent, rel, h, r, t, neg_t, 0.0, want_loss=False)
        ctx.save_for_backward(ent, rel, h, r, t, neg_t)
        ctx.has_neg = neg_t is not None
        return pos, neg

    @staticmethod
    def backward(ctx, dpos, dneg):
        ent, rel, h, r, t, neg_t = ctx.saved_tensors
        dpos = dpos.contiguous()
        dneg = dneg.contiguous() if neg_t is not None else None
        d_ent, d_rel = ops.score_margin_bwd(ent, rel, h, r, t, neg_t, 0.0, 1.0, dpos=dpos, dneg=dneg)
        return d_ent, d_rel, None, None, None, None


class _MarginLossFn(torch.autograd.Function):
    """(ent, rel) -> scalar hinge loss over 1 positive + K negative tails.

    When a gradient will be asked for, forward AND backward run as ONE launch (snag_score_margin_fwd_bwd: every table row is
    gathered once) and ``backward`` only scales the stored gradients by the upstream scalar -- on the device, no host
    synchronisation.  Without grad (evaluation) only the forward kernel runs."""

    @staticmethod
    def forward(ctx, ent, rel, h, r, t, neg_t, margin, want_grad):
        ctx.fused = bool(want_grad and neg_t is not None)
        if ctx.fused:
            loss, pos, neg, _, d_ent, d_rel = ops.score_margin_fwd_bwd(ent, rel, h, r, t, neg_t, margin)
            ctx.save_for_backward(d_ent, d_rel)
        else:
            pos, neg, part = ops.score_margin_fwd(ent, rel, h, r, t, neg_t, margin, want_loss=True)
            loss = ops.reduce_sum(part)
        ctx.mark_non_differentiable(pos, neg)
        return loss, pos, neg

    @staticmethod
    def backward(ctx, dloss, _dp, _dn):
        if not ctx.fused:
            raise RuntimeError("margin_loss: no input required a gradient in the forward")
        d_ent, d_rel = ctx.saved_tensors
        return d_ent * dloss, d_rel * dloss, None, None, None, None, None, None


class _SelfAdvLossFn(torch.autograd.Function):
    """(ent, rel) -> scalar self-adversarial loss; the epilogue kernel emits d loss / d scores, the scorer's explicit backward does the rest."""

    @staticmethod
    def forward(ctx, ent, rel, h, r, t, neg_t, margin, adv_temp):
        from ._lib import call, ptr, stream
        pos, neg, _ = ops.score_margin_fwd(ent, rel, h, r, t, neg_t, 0.0, want_loss=False)
        B, K = neg.shape
        part = torch.empty(B, device=ent.device, dtype=torch.float32)
        dpos, dneg = torch.empty_like(pos), torch.empty_like(neg)
        call("snag_selfadv_loss", ptr(pos), ptr(neg), B, K, float(margin), float(adv_temp), ptr(part), ptr(dpos), ptr(dneg), stream())
        ctx.save_for_backward(ent, rel, h, r, t, neg_t, dpos, dneg)
        ctx.mark_non_differentiable(pos, neg)
        return ops.reduce_sum(part), pos, neg

    @staticmethod
    def backward(ctx, dloss, _dp, _dn):
        ent, rel, h, r, t, neg_t, dpos, dneg = ctx.saved_tensors
        d_ent, d_rel = ops.score_margin_bwd(ent, rel, h, r, t, neg_t, 0.0, 1.0, dpos=dpos, dneg=dneg)
        return d_ent * dloss, d_rel * dloss, None, None, None, None, None, None      # upstream scalar applied on the device: no host sync


def self_adversarial_loss(ent, rel, h, r, t, neg_t, margin, adv_temp=1.0):
    """sum_b [-log sigmoid(margin - s_pos) - sum_k softmax_k(-adv_temp s_neg)_k log sigmoid(s_neg_k - margin)] -> (loss, pos, neg).

    The optional epilogue of SURVEY.md par. 8 a7 (``--adv_temp``, config.py:54); parity unpinned, see include/snag_hip.h.
    """
    return _SelfAdvLossFn.apply(ent, rel, h, r, t, neg_t, float(margin), float(adv_temp))


def score(ent, rel, h, r, t, neg_t=None):
    """s(h,r,t) and s(h,r,neg_t[b,k]) (TransE, squared L2 on L2-normalised rows).  Differentiable."""
    return _MarginScoreFn.apply(ent, rel, h, r, t, neg_t)


def margin_loss(ent, rel, h, r, t, neg_t, margin):
    """sum_b sum_k relu(margin + s(h,r,t) - s(h,r,neg_t[b,k])) -> (loss, pos, neg).

    The MKGC shape of BASELINE.json (NEG_NUM negative tails per positive).
    """
    want_grad = torch.is_grad_enabled() and (ent.requires_grad or rel.requires_grad)
    return _MarginLossFn.apply(ent, rel, h, r, t, neg_t, float(margin), want_grad)


def kge_relation_loss(ent, rel, p_h, p_r, p_t, n_h, n_r, n_t, margin):
    """In-tree pairing: sum_i relu(margin + s(pos_i) - s(neg_i)), arbitrary (head or tail) corruption.

    Same arguments as the first six of ``MSNEA.kge`` (model/MSNEA.py:108); term 1 (:120-122) only.
    Returns (loss, pos_score, neg_score).
    """
    pos, _ = score(ent, rel, p_h, p_r, p_t)
    neg, _ = score(ent, rel, n_h, n_r, n_t)
    loss = torch.relu(margin + pos - neg).sum()
    return loss, pos, neg
