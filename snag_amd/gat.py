"""Sparse GAT structural encoder (diag variant) on the GPU -- SURVEY.md par. 8(f) "next" #2.

Drop-in for ``GAT(n_units, n_heads, dropout, attn_dropout, instance_normalization, diag)`` (model/Tool_model.py:61-94) built from
``MultiHeadGraphAttention`` layers (model/layers.py:35-100): same constructor, same ``forward(x, adj)`` with ``adj`` a torch sparse COO
tensor (only ``adj._indices()`` is used, as in the reference), same state_dict keys ``layer_stack.{i}.w`` / ``.a_src_dst``.
Only what SNAG instantiates is built: ``diag=True``, no instance normalisation, dropout = attn_dropout = 0 (run_snag.sh sets neither).
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import call, ptr, stream

f32 = torch.float32


class MultiHeadGraphAttention(nn.Module):
    """Parameter container with the reference's shapes and initialisers (model/layers.py:41-66, init=nn.init.ones_, diag)."""

    def __init__(self, n_head, f_in, f_out, attn_dropout, diag=True, init=None, bias=False):
        super().__init__()
        if not diag or bias:
            raise NotImplementedError("only the diag, bias-free layer SNAG builds (Tool_model.py:72) is implemented")
        self.n_head, self.f_in, self.f_out, self.diag, self.attn_dropout = n_head, f_in, f_out, diag, attn_dropout
        self.w = nn.Parameter(torch.empty(n_head, 1, f_out))
        self.a_src_dst = nn.Parameter(torch.empty(n_head, f_out * 2, 1))
        if init is not None:
            init(self.w)
            stdv = 1.0 / math.sqrt(self.a_src_dst.size(1))
            nn.init.uniform_(self.a_src_dst, -stdv, stdv)
        else:
            nn.init.xavier_uniform_(self.w)
            nn.init.xavier_uniform_(self.a_src_dst)


def _group(key, other, n_nodes):
    order = torch.sort(key, stable=True)[1]
    col = other[order].to(torch.int32).contiguous()
    counts = torch.bincount(key, minlength=n_nodes)
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=key.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32).contiguous(), col


def build_csr(adj_or_edges, n_nodes):
    """COO index list [2, E] (or a sparse COO tensor) -> (rowptr [N+1], col [E], t_rowptr [N+1], t_col [E]) int32: the edges grouped by
    source (col = destinations) and, for the backward, by destination (t_col = sources); both stable in the stored edge order."""
    edges = adj_or_edges._indices() if adj_or_edges.layout != torch.strided else adj_or_edges
    src, dst = edges[0].long(), edges[1].long()
    return _group(src, dst, n_nodes) + _group(dst, src, n_nodes)


class _GatLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, a, rowptr, col, t_rowptr, t_col, apply_elu):
        ops._chk(x, "x", ndim=2)
        N, f = x.shape
        H = w.shape[0]
        dev = x.device
        w2 = w.detach().reshape(H, f).contiguous()
        a2 = a.detach().reshape(H, 2 * f).contiguous()
        s_src = torch.empty(N, H, device=dev, dtype=f32)
        s_dst = torch.empty(N, H, device=dev, dtype=f32)
        agg = torch.empty(N, H, f, device=dev, dtype=f32)
        rsum = torch.empty(N, H, device=dev, dtype=f32)
        pre = torch.empty(N, f, device=dev, dtype=f32) if apply_elu else None
        out = torch.empty(N, f, device=dev, dtype=f32)
        call("snag_gat_layer_fwd", ptr(x), N, f, H, ptr(w2), ptr(a2), ptr(rowptr), ptr(col), int(apply_elu), ptr(s_src), ptr(s_dst), ptr(agg),
             ptr(rsum), ptr(pre), ptr(out), stream())
        ctx.save_for_backward(x, w2, a2, rowptr, col, t_rowptr, t_col, s_src, s_dst, agg, rsum, pre)
        ctx.apply_elu = apply_elu
        ctx.shapes = (tuple(w.shape), tuple(a.shape))
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, w2, a2, rowptr, col, t_rowptr, t_col, s_src, s_dst, agg, rsum, pre = ctx.saved_tensors
        N, f = x.shape
        H = w2.shape[0]
        dev = x.device
        d_x = torch.empty_like(x)
        d_w = torch.empty(H, f, device=dev, dtype=f32)
        d_a = torch.empty(H, 2 * f, device=dev, dtype=f32)
        work = torch.empty(_lib.load().snag_gat_work_floats(N, f, H), device=dev, dtype=f32)
        call("snag_gat_layer_bwd", ptr(x), N, f, H, ptr(w2), ptr(a2), ptr(rowptr), ptr(col), ptr(t_rowptr), ptr(t_col), ptr(s_src), ptr(s_dst), ptr(agg), ptr(rsum),
             ptr(pre), int(ctx.apply_elu), ptr(d_out.contiguous()), ptr(d_x), ptr(d_w), ptr(d_a), 0, ptr(work), stream())
        return d_x, d_w.view(ctx.shapes[0]), d_a.view(ctx.shapes[1]), None, None, None, None, None


class GAT(nn.Module):
    """HIP drop-in for model/Tool_model.py:61-94."""

    def __init__(self, n_units, n_heads, dropout, attn_dropout, instance_normalization, diag):
        super().__init__()
        if instance_normalization or not diag:
            raise NotImplementedError("SNAG builds GAT(..., instance_normalization=False, diag=True) (model/SNAG_tools.py:101-104)")
        if dropout != 0.0 or attn_dropout != 0.0:
            raise NotImplementedError("dropout / attn_dropout > 0 are not built (run_snag.sh leaves both at their 0.0 defaults)")
        self.num_layer = len(n_units) - 1
        self.dropout, self.inst_norm, self.diag = dropout, instance_normalization, diag
        self.layer_stack = nn.ModuleList()
        for i in range(self.num_layer):
            f_in = n_units[i] * n_heads[i - 1] if i else n_units[i]
            self.layer_stack.append(MultiHeadGraphAttention(n_heads[i], f_in, n_units[i + 1], attn_dropout, diag, nn.init.ones_, False))
        self._csr_key, self._csr = None, None

    def _graph(self, adj, n):
        key = (id(adj), n)
        if self._csr_key != key:                     # the adjacency is fixed for a run (kgs["adj"], model/SNAG.py:25)
            self._csr, self._csr_key = build_csr(adj, n), key
        return self._csr

    def forward(self, x, adj):
        _lib.require_gpu()
        rowptr, col, t_rowptr, t_col = self._graph(adj, x.shape[0])
        x = x.contiguous()
        for i, layer in enumerate(self.layer_stack):
            if layer.w.shape[2] != x.shape[1]:
                raise ValueError(f"diag GAT layer {i} expects width {layer.w.shape[2]}, got {x.shape[1]}")
            x = _GatLayerFn.apply(x, layer.w, layer.a_src_dst, rowptr, col, t_rowptr, t_col, i + 1 < self.num_layer)
        return x
