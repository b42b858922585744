"""GCN structural encoder on the GPU: the ``structure_encoder == "gcn"`` option of MultiModalEncoder (model/SNAG_tools.py:98-100).

Drop-in for ``GCN(nfeat, nhid, nout, dropout)`` (model/Tool_model.py:98-110) and ``GraphConvolution`` (model/layers.py:102-130): same
constructors, same initialisers, same state_dict keys (``gc1.weight [in, out]``, ``gc1.bias``, ``gc2.*``), same ``forward(x, adj)`` with
``adj`` a torch sparse COO tensor whose values ARE used (torch.spmm).  X W is an split-bf16 MFMA GEMM (fp32 in, fp32 accumulate), A (X W) a CSR gather kernel
(snag_spmm_csr); the backward runs the same kernel on the destination-grouped edge list, so nothing is scattered with atomics.
Only dropout = 0 is built (run_snag.sh leaves --dropout at 0.0).
"""
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import call, ptr, stream

f32 = torch.float32


def build_csr_values(adj, n_nodes):
    """Sparse COO adjacency -> ((rowptr, col, val), (t_rowptr, t_col, t_val)): edges grouped by row for the forward and by column
    for the backward, values carried along; stable in the stored order (duplicates are kept and add up)."""
    idx, val = adj._indices(), adj._values().to(f32)
    out = []
    for key, other in ((idx[0].long(), idx[1].long()), (idx[1].long(), idx[0].long())):
        order = torch.sort(key, stable=True)[1]
        counts = torch.bincount(key, minlength=n_nodes)
        rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=key.device)
        rowptr[1:] = torch.cumsum(counts, 0)
        out.append((rowptr.to(torch.int32).contiguous(), other[order].to(torch.int32).contiguous(), val[order].contiguous()))
    return tuple(out)


def _spmm(csr, x, bias=None):
    rowptr, col, val = csr
    out = torch.empty_like(x)
    call("snag_spmm_csr", ptr(rowptr), ptr(col), ptr(val), ptr(x), x.shape[0], x.shape[1], ptr(bias), ptr(out), stream())
    return out


class _GraphConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, fwd_csr, bwd_csr):
        ops._chk(x, "x", ndim=2)
        support = ops.gemm(x, weight, b_kcontig=False)                 # input @ weight, weight stored [in, out]   (:125)
        out = _spmm(fwd_csr, support, bias)                            # spmm(adj, support) + bias               (:126-128)
        ctx.save_for_backward(x, weight)
        ctx.bwd_csr, ctx.has_bias = bwd_csr, bias is not None
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, weight = ctx.saved_tensors
        d_out = d_out.contiguous()
        d_support = _spmm(ctx.bwd_csr, d_out)                          # A^T d_out
        d_x = ops.gemm(d_support, weight)                              # d_support @ weight^T
        d_w = ops.gemm(x, d_support, a_kcontig=False, b_kcontig=False)  # x^T @ d_support
        d_b = ops.colsum(d_out) if ctx.has_bias else None
        return d_x, d_w, d_b, None, None


class GraphConvolution(nn.Module):
    """model/layers.py:102-130."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, csr_pair):
        fwd_csr, bwd_csr = csr_pair
        return _GraphConvFn.apply(input.contiguous(), self.weight, self.bias, fwd_csr, bwd_csr)


class GCN(nn.Module):
    """HIP drop-in for model/Tool_model.py:98-110: relu(gc1(x, adj)) -> dropout -> gc2(., adj)."""

    def __init__(self, nfeat, nhid, nout, dropout):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 is not built (run_snag.sh leaves --dropout at its 0.0 default)")
        self.gc1 = GraphConvolution(nfeat, nhid)
        self.gc2 = GraphConvolution(nhid, nout)
        self.dropout = dropout
        self._csr_key, self._csr = None, None

    def _graph(self, adj, n):
        key = (id(adj), n)
        if self._csr_key != key:                     # the adjacency is fixed for a run (kgs["adj"], model/SNAG.py:25)
            self._csr, self._csr_key = build_csr_values(adj, n), key
        return self._csr

    def forward(self, x, adj):
        _lib.require_gpu()
        csr = self._graph(adj, x.shape[0])
        x = torch.relu(self.gc1(x, csr))
        return self.gc2(x, csr)
