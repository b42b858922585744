"""Sparse GAT structural encoder (diag variant) on the GPU -- SURVEY.md par. 8(f) "next" #2.

Drop-in for ``GAT(n_units, n_heads, dropout, attn_dropout, instance_normalization, diag)`` (model/Tool_model.py:61-94) built from
``MultiHeadGraphAttention`` layers (model/layers.py:35-100): same constructor, same ``forward(x, adj)`` with ``adj`` a torch sparse COO
tensor (only ``adj._indices()`` is used, as in the reference), same state_dict keys ``layer_stack.{i}.w`` / ``.a_src_dst``.
``diag=True`` as SNAG instantiates it; round 5: the training options ``dropout`` (F.dropout on the input of every layer but the last,
Tool_model.py:78-79) and ``attn_dropout`` (F.dropout on the edge weights after the row sums, layers.py:83), with masks drawn on the device
(Philox, seeded from torch's generator) or passed explicitly (``forward(x, adj, drop_masks=...)``: parity tests, golden G16).
``instance_normalization=True`` is refused: the reference's ``nn.InstanceNorm1d(n_units[0])`` itself raises on [N, f] node features.
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
    return rowptr.to(torch.int32).contiguous(), col, order.to(torch.int32).contiguous()


def build_csr(adj_or_edges, n_nodes):
    """COO index list [2, E] (or a sparse COO tensor) -> (rowptr [N+1], col [E], t_rowptr [N+1], t_col [E], eid [E], t_eid [E]) int32: the edges
    grouped by source (col = destinations) and, for the backward, by destination (t_col = sources); both stable in the stored edge order;
    eid / t_eid = the stored edge id of every position of the two orders (what an edge keep mask in stored order is read through)."""
    edges = adj_or_edges._indices() if adj_or_edges.layout != torch.strided else adj_or_edges
    src, dst = edges[0].long(), edges[1].long()
    rowptr, col, eid = _group(src, dst, n_nodes)
    t_rowptr, t_col, t_eid = _group(dst, src, n_nodes)
    return rowptr, col, t_rowptr, t_col, eid, t_eid


class _GatLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, a, rowptr, col, t_rowptr, t_col, apply_elu, edge_keep=None, eid=None, t_eid=None, edge_p=0.0):
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
        E = col.shape[0]
        if edge_keep is not None and (edge_keep.dtype != torch.uint8 or tuple(edge_keep.shape) != (H, E) or not edge_keep.is_cuda):
            raise ValueError(f"edge keep mask must be a uint8 CUDA tensor [{H}, {E}] in the stored edge order")
        ek = edge_keep.contiguous() if edge_keep is not None else None
        scale = 1.0 / (1.0 - edge_p) if ek is not None else 1.0
        call("snag_gat_layer_fwd", ptr(x), N, f, H, ptr(w2), ptr(a2), ptr(rowptr), ptr(col), int(apply_elu), ptr(s_src), ptr(s_dst), ptr(agg),
             ptr(rsum), ptr(pre), ptr(out), ptr(ek), ptr(eid) if ek is not None else None, E, scale, stream())
        ctx.save_for_backward(x, w2, a2, rowptr, col, t_rowptr, t_col, s_src, s_dst, agg, rsum, pre, ek, eid, t_eid)
        ctx.apply_elu, ctx.edge_scale = apply_elu, scale
        ctx.shapes = (tuple(w.shape), tuple(a.shape))
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, w2, a2, rowptr, col, t_rowptr, t_col, s_src, s_dst, agg, rsum, pre, ek, eid, t_eid = ctx.saved_tensors
        N, f = x.shape
        H = w2.shape[0]
        dev = x.device
        d_x = torch.empty_like(x)
        d_w = torch.empty(H, f, device=dev, dtype=f32)
        d_a = torch.empty(H, 2 * f, device=dev, dtype=f32)
        work = torch.empty(_lib.load().snag_gat_work_floats(N, f, H), device=dev, dtype=f32)
        call("snag_gat_layer_bwd", ptr(x), N, f, H, ptr(w2), ptr(a2), ptr(rowptr), ptr(col), ptr(t_rowptr), ptr(t_col), ptr(s_src), ptr(s_dst), ptr(agg), ptr(rsum),
             ptr(pre), int(ctx.apply_elu), ptr(d_out.contiguous()), ptr(d_x), ptr(d_w), ptr(d_a), 0, ptr(work), ptr(ek),
             ptr(eid) if ek is not None else None, ptr(t_eid) if ek is not None else None, col.shape[0], ctx.edge_scale, stream())
        return d_x, d_w.view(ctx.shapes[0]), d_a.view(ctx.shapes[1]), None, None, None, None, None, None, None, None, None


class _KeepScaleFn(torch.autograd.Function):
    """F.dropout with an explicit keep mask (the input dropout of the non-final layers, model/Tool_model.py:78-79)."""

    @staticmethod
    def forward(ctx, x, keep, p):
        ops._chk(x, "x", ndim=2)
        if keep.dtype != torch.uint8 or keep.shape != x.shape or not keep.is_cuda:
            raise ValueError("input keep mask must be a uint8 CUDA tensor of the input's shape")
        keep = keep.contiguous()
        out = torch.empty_like(x)
        ctx.scale = 1.0 / (1.0 - p)
        call("snag_keep_scale", ptr(x), x.numel(), ptr(keep), ctx.scale, ptr(out), stream())
        ctx.save_for_backward(keep)
        return out

    @staticmethod
    def backward(ctx, g):
        (keep,) = ctx.saved_tensors
        g = g.contiguous()
        d = torch.empty_like(g)
        call("snag_keep_scale", ptr(g), g.numel(), ptr(keep), ctx.scale, ptr(d), stream())
        return d, None, None


class GAT(nn.Module):
    """HIP drop-in for model/Tool_model.py:61-94."""

    def __init__(self, n_units, n_heads, dropout, attn_dropout, instance_normalization, diag):
        super().__init__()
        if not diag:
            raise NotImplementedError("SNAG builds GAT(..., diag=True) (model/SNAG_tools.py:101-104)")
        if instance_normalization:
            raise NotImplementedError("instance_normalization: the reference's nn.InstanceNorm1d(n_units[0]) raises ValueError on [N, f] node features "
                                      "(model/Tool_model.py:66-68,76-77): there is no behaviour to reproduce")
        if not (0.0 <= dropout < 1.0 and 0.0 <= attn_dropout < 1.0):
            raise ValueError("dropout rates must lie in [0, 1)")
        self.num_layer = len(n_units) - 1
        self.dropout, self.attn_dropout, self.inst_norm, self.diag = float(dropout), float(attn_dropout), instance_normalization, diag
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

    def forward(self, x, adj, drop_masks=None):
        """``drop_masks`` (training, explicit masks instead of drawn ones): {"in": [uint8 [N, f] or None per layer], "edge": [uint8 [H, E] in the
        stored edge order or None per layer]}."""
        _lib.require_gpu()
        rowptr, col, t_rowptr, t_col, eid, t_eid = self._graph(adj, x.shape[0])
        x = x.contiguous()
        E = col.shape[0]
        for i, layer in enumerate(self.layer_stack):
            if layer.w.shape[2] != x.shape[1]:
                raise ValueError(f"diag GAT layer {i} expects width {layer.w.shape[2]}, got {x.shape[1]}")
            in_keep = edge_keep = None
            if self.training:
                if drop_masks is not None:
                    in_keep, edge_keep = drop_masks["in"][i], drop_masks["edge"][i]
                else:           # one Philox launch per mask, seeded from torch's CPU generator (torch.manual_seed makes a run reproducible)
                    if self.dropout > 0 and i + 1 < self.num_layer:
                        in_keep = ops.dropout_keep_mask(tuple(x.shape), self.dropout, int(torch.randint(0, 2 ** 62, (1,)).item()), device=x.device)
                    if self.attn_dropout > 0:
                        edge_keep = ops.dropout_keep_mask((layer.n_head, E), self.attn_dropout, int(torch.randint(0, 2 ** 62, (1,)).item()), device=x.device)
            if in_keep is not None:
                x = _KeepScaleFn.apply(x, in_keep, self.dropout)
            x = _GatLayerFn.apply(x, layer.w, layer.a_src_dst, rowptr, col, t_rowptr, t_col, i + 1 < self.num_layer, edge_keep, eid, t_eid, self.attn_dropout)
        return x
