"""MMEA training objective on the GPU (SURVEY.md par. 8f "next" #1).

``icl_loss`` keeps the reference's constructor and call signature (model/SNAG_loss.py:31-41,58): bidirectional in-batch
InfoNCE over aligned pairs with an optional per-entity weight.  Gather + normalise, the masked row log-sum-exp and the
logits-to-gradient transform are HIP kernels; the B x B x d products are exact-f32 MFMA GEMMs.
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
        z = [torch.empty(B, d, device=dev, dtype=f32) for _ in range(2)]
        den = [torch.empty(B, device=dev, dtype=f32) for _ in range(2)]
        for k, idx in enumerate((left, right)):
            call("snag_gather_normalize_fwd", ptr(emb), emb.stride(0), emb.shape[0], d, ptr(idx), ib, B, int(normalize), ptr(z[k]), ptr(den[k]), st)
        z1, z2 = z
        inv_tau = 1.0 / tau
        La = torch.empty(B, 2 * B, device=dev, dtype=f32)
        Lb = torch.empty(B, 2 * B, device=dev, dtype=f32)
        ops.gemm(z1, z2, alpha=inv_tau, out=La[:, :B])          # logits_ab  (:108)
        ops.gemm(z1, z1, alpha=inv_tau, out=La[:, B:])          # logits_aa  (:99), diagonal masked in the row kernel
        ops.gemm(z2, z1, alpha=inv_tau, out=Lb[:, :B])          # logits_ba  (:109)
        ops.gemm(z2, z2, alpha=inv_tau, out=Lb[:, B:])          # logits_bb  (:102)
        lse = [torch.empty(B, device=dev, dtype=f32) for _ in range(2)]
        rl = [torch.empty(B, device=dev, dtype=f32) for _ in range(2)]
        for k, L in enumerate((La, Lb)):
            call("snag_infonce_rows", ptr(L), L.stride(0), B, ptr(lse[k]), ptr(rl[k]), st)
        per_row = (alpha * rl[0] + (1.0 - alpha) * rl[1]) / B          # [B] glue; the weighting below is differentiable in w
        ctx.save_for_backward(emb, w, left, right, z1, z2, den[0], den[1], La, Lb, lse[0], lse[1], per_row)
        ctx.cfg = (tau, alpha, normalize, ib)
        return ops.reduce_sum((per_row * w).contiguous() if w is not None else per_row)

    @staticmethod
    def backward(ctx, g):
        emb, w, left, right, z1, z2, den1, den2, La, Lb, lse1, lse2, per_row = ctx.saved_tensors
        tau, alpha, normalize, ib = ctx.cfg
        B, d = z1.shape
        st = stream()
        gs = float(g)
        # logits -> d loss / d logits, in place (the saved logits are consumed)
        call("snag_infonce_grad", ptr(La), La.stride(0), B, ptr(lse1), ptr(w), gs * alpha / B, st)
        call("snag_infonce_grad", ptr(Lb), Lb.stride(0), B, ptr(lse2), ptr(w), gs * (1.0 - alpha) / B, st)
        it = 1.0 / tau
        dz1 = torch.empty_like(z1)
        dz2 = torch.empty_like(z2)
        Ga1, Ga2, Gb1, Gb2 = La[:, :B], La[:, B:], Lb[:, :B], Lb[:, B:]
        # ab = z1 z2^T: dz1 += Ga1 z2 ; dz2 += Ga1^T z1        aa = z1 z1^T: dz1 += (Ga2 + Ga2^T) z1
        ops.gemm(Ga1, z2, b_kcontig=False, alpha=it, out=dz1)
        ops.gemm(Ga2, z1, b_kcontig=False, alpha=it, out=dz1, accumulate=True)
        ops.gemm(Ga2, z1, a_kcontig=False, b_kcontig=False, alpha=it, out=dz1, accumulate=True)
        ops.gemm(Gb1, z2, a_kcontig=False, b_kcontig=False, alpha=it, out=dz1, accumulate=True)      # ba = z2 z1^T: dz1 += Gb1^T z2
        ops.gemm(Gb1, z1, b_kcontig=False, alpha=it, out=dz2)                                        #                 dz2 += Gb1 z1
        ops.gemm(Ga1, z1, a_kcontig=False, b_kcontig=False, alpha=it, out=dz2, accumulate=True)
        ops.gemm(Gb2, z2, b_kcontig=False, alpha=it, out=dz2, accumulate=True)
        ops.gemm(Gb2, z2, a_kcontig=False, b_kcontig=False, alpha=it, out=dz2, accumulate=True)
        d_emb = torch.zeros_like(emb)
        for dz, z, den, idx in ((dz1, z1, den1, left), (dz2, z2, den2, right)):
            call("snag_gather_normalize_bwd", ptr(dz), ptr(z), ptr(den), d, ptr(idx), ib, B, int(normalize), ptr(d_emb), d_emb.stride(0), st)
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
