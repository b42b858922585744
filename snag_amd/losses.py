"""MMEA training objective on the GPU (SURVEY.md par. 8f "next" #1).

``icl_loss`` keeps the reference's constructor and call signature (model/SNAG_loss.py:31-41,58): bidirectional in-batch
InfoNCE over aligned pairs with an optional per-entity weight.  Gather + normalise, the masked row log-sum-exp and the
logits-to-gradient transform are HIP kernels; the 2B x 2B x d products (one forward, one backward) are split-bf16 MFMA GEMMs (fp32 in, fp32 accumulate).
``CustomMultiLossLayer`` (:12-29) is six scalars and stays plain torch.
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import call, ptr, stream

f32 = torch.float32


class _IclFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, w, left, right, tau, alpha, normalize):
        ops._rows(emb, "emb")
        dev = emb.device
        B, d = left.shape[0], emb.shape[1]
        ib = 4 if left.dtype == torch.int32 else 8
        st = stream()
        Z = torch.empty(2 * B, d, device=dev, dtype=f32)               # stacked views [z1; z2]
        den = torch.empty(2 * B, device=dev, dtype=f32)
        for k, idx in enumerate((left, right)):
            call("snag_gather_normalize_fwd", ptr(emb), emb.stride(0), emb.shape[0], d, ptr(idx), ib, B, int(normalize),
                 ptr(Z[k * B:]), ptr(den[k * B:]), st)
        # S = Z Z^T / tau holds logits_aa, _ab, _ba, _bb (:99-109) as its four blocks; the diagonal is masked in the row kernel
        # (row pitch padded to whole 128-byte lines: with 2B = 7000 a 28,000-byte pitch makes every tile's row segments straddle
        # lines shared with the neighbouring tile, which costs the GEMM ~20 %)
        pitch = (2 * B + 31) // 32 * 32
        S = ops.gram(Z, alpha=1.0 / tau, out=torch.empty(2 * B, pitch, device=dev, dtype=f32)[:, :2 * B])
        lse = torch.empty(2 * B, device=dev, dtype=f32)
        rl = torch.empty(2 * B, device=dev, dtype=f32)
        call("snag_infonce_rows", ptr(S), S.stride(0), B, ptr(lse), ptr(rl), st)
        per_row = (alpha * rl[:B] + (1.0 - alpha) * rl[B:]) / B          # [B] glue; the weighting below is differentiable in w
        ctx.save_for_backward(emb, w, left, right, Z, den, S, lse, per_row)
        ctx.cfg = (tau, alpha, normalize, ib)
        return ops.reduce_sum((per_row * w).contiguous() if w is not None else per_row)

    @staticmethod
    def backward(ctx, g):
        emb, w, left, right, Z, den, S, lse, per_row = ctx.saved_tensors
        tau, alpha, normalize, ib = ctx.cfg
        B, d = left.shape[0], Z.shape[1]
        st = stream()
        gdev = g.detach().to(f32).reshape(1).contiguous()              # upstream gradient stays on the device: no host sync
        # logits -> (G + G^T), G = d loss / d logits, in place (the saved logits are consumed); then dZ = (G + G^T) Z / tau
        call("snag_infonce_grad", ptr(S), S.stride(0), B, ptr(lse), ptr(w), alpha / B, (1.0 - alpha) / B, ptr(gdev), st)
        dZ = ops.gemm(S, Z, b_kcontig=False, alpha=1.0 / tau)
        d_emb = torch.zeros_like(emb)
        for k, idx in enumerate((left, right)):
            call("snag_gather_normalize_bwd", ptr(dZ[k * B:]), ptr(Z[k * B:]), ptr(den[k * B:]), d, ptr(idx), ib, B, int(normalize),
                 ptr(d_emb), d_emb.stride(0), st)
        d_w = per_row * g if (w is not None and ctx.needs_input_grad[1]) else None
        return d_emb, d_w, None, None, None, None, None


class icl_loss(nn.Module):
    """Drop-in for model/SNAG_loss.py:31 (``icl_loss(tau, ab_weight, n_view, intra_weight, inversion, neg_cross_kg)``)."""

    def __init__(self, tau=0.05, ab_weight=0.5, n_view=2, intra_weight=1.0, inversion=False, neg_cross_kg=False):
        super().__init__()
        if n_view != 2 or inversion:
            raise NotImplementedError("only the configuration SNAG instantiates (n_view=2, inversion=False) is built")
        self.tau, self.weight, self.n_view = tau, ab_weight, n_view
        self.intra_weight, self.inversion, self.neg_cross_kg = intra_weight, inversion, neg_cross_kg

    def forward(self, emb, train_links, neg_l=None, neg_r=None, weight_norm=None, norm=True):
        if neg_l is not None or neg_r is not None:
            raise NotImplementedError("explicit negatives (neg_l / neg_r) are never passed by SNAG (model/SNAG.py:106-162)")
        links = torch.as_tensor(train_links, device=emb.device)
        if links.dtype not in (torch.int32, torch.int64):
            links = links.long()
        if links.dim() != 2 or links.shape[0] == 0 or links.shape[1] < 2:
            raise ValueError("icl_loss: train_links must be [B >= 1, 2] (the reference's mean over an empty batch is NaN)")
        left, right = links[:, 0].contiguous(), links[:, 1].contiguous()
        w = None
        if weight_norm is not None:                                    # :66-70 min over the pair, differentiable
            w = torch.minimum(weight_norm[left.long()], weight_norm[right.long()]).contiguous()
        return _IclFn.apply(emb, w, left, right, float(self.tau), float(self.weight), bool(norm))


class CustomMultiLossLayer(nn.Module):
    """sum_i exp(-s_i) L_i + s_i with learnable log-variances s (model/SNAG_loss.py:12-29)."""

    def __init__(self, loss_num):
        super().__init__()
        self.loss_num = loss_num
        self.log_vars = nn.Parameter(torch.zeros(loss_num), requires_grad=True)

    def forward(self, loss_list):
        assert len(loss_list) <= self.loss_num
        precision = torch.exp(-self.log_vars)
        loss = 0
        for i in range(len(loss_list)):
            loss = loss + precision[i] * loss_list[i] + self.log_vars[i]
        return loss
