"""Per-modality projections on the matrix cores (fp32 in, fp32 accumulate, split-bf16 products).

Reference: ``img_fc / rel_fc / att_fc / name_fc / char_fc = nn.Linear(F_m, D)`` (model/SNAG_tools.py:92-96, used at
:133-152) -- README ``num_proj = 1`` -- and ``ProjectionHead`` (model/layers.py:135-148: Linear(no bias) -> ReLU ->
dropout -> Linear(no bias)) -- the ``num_proj = 2`` analogue.  Same parameter names / state_dict keys; forward and
backward are HIP GEMMs with fused bias / ReLU / dropout epilogues.
"""
import math

import torch
import torch.nn as nn

from . import ops


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return ops.gemm(x, w, bias=b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.gemm(dy, w, b_kcontig=False) if ctx.needs_input_grad[0] else None          # dY W
        dw = ops.gemm(dy, x, a_kcontig=False, b_kcontig=False)                               # dY^T X
        db = ops.colsum(dy) if ctx.has_bias else None
        return dx, dw, db


def linear(x, weight, bias=None):
    """x @ weight.T + bias (nn.Linear) through snag_gemm_f32, differentiable."""
    return _LinearFn.apply(x.contiguous(), weight.contiguous(), bias)


class Linear(nn.Linear):
    """nn.Linear whose forward/backward run in libsnag_hip.so (same parameters, same init, same state_dict)."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class _ProjectionHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, w2, keep, p):
        h = ops.gemm(x, w1, act="relu", keep_mask=keep, keep_p=p)          # relu then inverted dropout, fused
        ctx.save_for_backward(x, w1, w2, h, keep)
        ctx.p = p
        return ops.gemm(h, w2)

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, h, keep = ctx.saved_tensors
        dy = dy.contiguous()
        dw2 = ops.gemm(dy, h, a_kcontig=False, b_kcontig=False)
        # d(pre-relu) = (dY W2) * keep/(1-p) * [h > 0]  (h > 0 iff relu passed and the unit was kept)
        dh = ops.gemm(dy, w2, b_kcontig=False, act="relu_grad", aux=h, keep_mask=keep, keep_p=ctx.p)
        dw1 = ops.gemm(dh, x, a_kcontig=False, b_kcontig=False)
        dx = ops.gemm(dh, w1, b_kcontig=False) if ctx.needs_input_grad[0] else None
        return dx, dw1, dw2, None, None


class ProjectionHead(nn.Module):
    """Drop-in for model/layers.py:135-148 (constructor ``ProjectionHead(in_dim, hidden_dim, out_dim, dropout)``)."""

    def __init__(self, in_dim, hidden_dim, out_dim, dropout):
        super().__init__()
        self.l1 = nn.Linear(in_dim, hidden_dim, bias=False)
        self.l2 = nn.Linear(hidden_dim, out_dim, bias=False)
        self.dropout = dropout

    def forward(self, x, keep_mask=None):
        if x is None:                                   # reference returns None for a missing modality (:142)
            return None
        keep = keep_mask
        if keep is None and self.training and self.dropout > 0:
            keep = ops.dropout_keep_mask((x.shape[0], self.l1.out_features), self.dropout,
                                         int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), device=x.device)
        return _ProjectionHeadFn.apply(x.contiguous(), self.l1.weight, self.l2.weight, keep, float(self.dropout))
