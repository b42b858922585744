"""Thin, checked Python wrappers over the C ABI: tensors in, tensors out, no arithmetic here.

Shape/dtype/contiguity errors are raised in Python before the call, mirroring the reference's
assert/exception convention (SURVEY.md par. 8b "Errors").
"""
import torch

from . import _lib
from ._lib import call, ptr, stream

f32 = torch.float32


def _chk(t, name, dtype=f32, ndim=None):
    if not torch.is_tensor(t):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise _lib.SnagHipError(f"{name}: tensor is on {t.device}; snag_amd kernels run on the GPU only (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dims, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _idx(t, name, ndim):
    if not torch.is_tensor(t) or t.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"{name}: expected an int32/int64 tensor")
    return _chk(t, name, dtype=None, ndim=ndim)


def _idx_group(h, r, t, neg_t):
    _idx(h, "h", 1), _idx(r, "r", 1), _idx(t, "t", 1)
    B = h.shape[0]
    if r.shape[0] != B or t.shape[0] != B:
        raise ValueError(f"h, r, t must have equal length, got {B}, {r.shape[0]}, {t.shape[0]}")
    dt = h.dtype
    if r.dtype != dt or t.dtype != dt:
        raise TypeError("h, r, t must share one index dtype")
    K = 0
    if neg_t is not None:
        _idx(neg_t, "neg_t", 2)
        if neg_t.shape[0] != B or neg_t.dtype != dt:
            raise ValueError(f"neg_t must be [{B}, K] of dtype {dt}, got {tuple(neg_t.shape)} {neg_t.dtype}")
        K = neg_t.shape[1]
    return B, K, (4 if dt == torch.int32 else 8)


def _rows(t, name):
    """2-D fp32 CUDA tensor with unit inner stride (row stride free: lets a [N, M, D] slot be used in place)."""
    if not (torch.is_tensor(t) and t.dim() == 2 and t.dtype == f32 and t.stride(1) == 1):
        raise ValueError(f"{name}: expected a 2-D fp32 tensor with unit inner stride")
    if not t.is_cuda:
        raise _lib.SnagHipError(f"{name}: tensor is on {t.device}; snag_amd kernels run on the GPU only (no CPU fallback)")
    return t


def _quads(entry, **tensors):
    """The entry moves rows as float4: every base pointer 16-byte aligned, every row stride a multiple of four floats (what the C ABI
    itself refuses with SNAG_ERR_UNSUPPORTED; checked here for the readable message)."""
    for name, t in tensors.items():
        if t.numel() and (t.data_ptr() % 16 or (t.dim() == 2 and t.stride(0) % 4)):
            raise ValueError(f"{entry}: {name} must be 16-byte aligned with a row stride that is a multiple of 4 floats "
                             f"(data_ptr % 16 = {t.data_ptr() % 16}, strides {tuple(t.stride())})")


def score_margin_fwd(ent, rel, h, r, t, neg_t, margin, want_loss=True):
    """-> pos [B], neg [B,K], loss_part [B] (or None).  C ABI: snag_score_margin_fwd."""
    _rows(ent, "ent"), _chk(rel, "rel", ndim=2)
    if ent.shape[1] != rel.shape[1]:
        raise ValueError(f"ent dim {ent.shape[1]} != rel dim {rel.shape[1]}")
    B, K, ib = _idx_group(h, r, t, neg_t)
    pos = torch.empty(B, device=ent.device, dtype=f32)
    neg = torch.empty(B, K, device=ent.device, dtype=f32)
    loss_part = torch.empty(B, device=ent.device, dtype=f32) if want_loss else None
    call("snag_score_margin_fwd", ptr(ent), ent.stride(0), ent.shape[0], ptr(rel), rel.shape[0], ent.shape[1],
         ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(margin), ptr(pos), ptr(neg), ptr(loss_part), stream())
    return pos, neg, loss_part


def score_margin_fwd_bf16(ent, rel, h, r, t, neg_t, margin, want_loss=True):
    """bf16 tables -> fp32 scores.  C ABI: snag_score_margin_fwd_bf16."""
    for x, nm in ((ent, "ent"), (rel, "rel")):
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1):
            raise ValueError(f"{nm}: expected a 2-D bfloat16 CUDA tensor with unit inner stride")
    if not rel.is_contiguous():
        raise ValueError("rel must be contiguous")
    B, K, ib = _idx_group(h, r, t, neg_t)
    pos = torch.empty(B, device=ent.device, dtype=f32)
    neg = torch.empty(B, K, device=ent.device, dtype=f32)
    loss_part = torch.empty(B, device=ent.device, dtype=f32) if want_loss else None
    call("snag_score_margin_fwd_bf16", ptr(ent), ent.stride(0), ent.shape[0], ptr(rel), rel.shape[0], ent.shape[1],
         ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(margin), ptr(pos), ptr(neg), ptr(loss_part), stream())
    return pos, neg, loss_part


def score_margin_bwd(ent, rel, h, r, t, neg_t, margin, grad_scale=1.0, dpos=None, dneg=None, d_ent=None, d_rel=None):
    """Accumulates into d_ent/d_rel (allocated zeroed when None).  C ABI: snag_score_margin_bwd."""
    _rows(ent, "ent"), _chk(rel, "rel", ndim=2)
    B, K, ib = _idx_group(h, r, t, neg_t)
    if dpos is not None:
        _chk(dpos, "dpos", ndim=1)
        if dpos.shape[0] != B:
            raise ValueError("dpos must be [B]")
    if dneg is not None:
        _chk(dneg, "dneg", ndim=2)
        if tuple(dneg.shape) != (B, K):
            raise ValueError("dneg must be [B, K]")
    if d_ent is None:
        d_ent = torch.zeros(ent.shape, device=ent.device, dtype=f32)
    if d_rel is None:
        d_rel = torch.zeros_like(rel)
    _rows(d_ent, "d_ent"), _chk(d_rel, "d_rel", ndim=2)
    if d_ent.shape != ent.shape or d_rel.shape != rel.shape:
        raise ValueError("gradient buffers must match the table shapes")
    call("snag_score_margin_bwd", ptr(ent), ent.stride(0), ent.shape[0], ptr(rel), rel.shape[0], ent.shape[1],
         ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(margin), float(grad_scale),
         ptr(dpos), ptr(dneg), ptr(d_ent), d_ent.stride(0), ptr(d_rel), stream())
    return d_ent, d_rel


_TICKETS = {}


def _ticket(device):
    """SNAG_SCORE_TICKET_WORDS zeroed uint32 per device for the fused scorer's last-block reduction (the kernel leaves it zero again)."""
    key = (device.type, device.index, stream())          # per stream: two streams' launches must not share the ticket words
    if key not in _TICKETS:
        _TICKETS[key] = torch.zeros(_lib.SCORE_TICKET_WORDS, device=device, dtype=torch.int32)
    return _TICKETS[key]


def score_margin_fwd_bwd(ent, rel, h, r, t, neg_t, margin, grad_scale=1.0, d_ent=None, d_rel=None, want_loss=True):
    """Hinge scorer, forward and backward in one launch -> (loss 0-dim or None, pos [B], neg [B,K], loss_part [B], d_ent, d_rel).
    Accumulates into d_ent / d_rel (allocated zeroed when None).  C ABI: snag_score_margin_fwd_bwd."""
    _rows(ent, "ent"), _chk(rel, "rel", ndim=2)
    if ent.shape[1] != rel.shape[1]:
        raise ValueError(f"ent dim {ent.shape[1]} != rel dim {rel.shape[1]}")
    if neg_t is None:
        raise ValueError("score_margin_fwd_bwd needs explicit negative tails")
    B, K, ib = _idx_group(h, r, t, neg_t)
    if d_ent is None:
        d_ent = torch.zeros(ent.shape, device=ent.device, dtype=f32)
    if d_rel is None:
        d_rel = torch.zeros_like(rel)
    _rows(d_ent, "d_ent"), _chk(d_rel, "d_rel", ndim=2)
    if d_ent.shape != ent.shape or d_rel.shape != rel.shape:
        raise ValueError("gradient buffers must match the table shapes")
    pos = torch.empty(B, device=ent.device, dtype=f32)
    neg = torch.empty(B, K, device=ent.device, dtype=f32)
    loss_part = torch.empty(B, device=ent.device, dtype=f32)
    loss = torch.empty((), device=ent.device, dtype=f32) if want_loss else None
    call("snag_score_margin_fwd_bwd", ptr(ent), ent.stride(0), ent.shape[0], ptr(rel), rel.shape[0], ent.shape[1],
         ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(margin), float(grad_scale), ptr(pos), ptr(neg), ptr(loss_part),
         ptr(loss), ptr(_ticket(ent.device)) if want_loss else None, ptr(d_ent), d_ent.stride(0), ptr(d_rel), stream())
    return loss, pos, neg, loss_part, d_ent, d_rel


_SEG_WORK = {}           # at most _SEG_WORK_MAX workspaces, least recently used dropped (ragged batch sizes would otherwise keep one each)
_SEG_WORK_MAX = 4


def score_margin_fwd_bwd_seg(ent, rel, h, r, t, neg_t, margin, grad_scale=1.0, d_ent=None, d_rel=None, write_untouched=True):
    """Hinge scorer, forward and backward WITHOUT float atomics (per-triple pass + per-destination pass, bit-reproducible)
    -> (loss, pos, neg, loss_part, d_ent, d_rel); d_ent / d_rel are overwritten.  C ABI: snag_score_margin_fwd_bwd_seg."""
    _rows(ent, "ent"), _chk(rel, "rel", ndim=2)
    if ent.shape[1] != rel.shape[1]:
        raise ValueError(f"ent dim {ent.shape[1]} != rel dim {rel.shape[1]}")
    if neg_t is None:
        raise ValueError("score_margin_fwd_bwd_seg needs explicit negative tails")
    B, K, ib = _idx_group(h, r, t, neg_t)
    if d_ent is None:
        d_ent = torch.empty(ent.shape, device=ent.device, dtype=f32) if write_untouched else torch.zeros(ent.shape, device=ent.device, dtype=f32)
    if d_rel is None:
        d_rel = torch.empty_like(rel) if write_untouched else torch.zeros_like(rel)
    _rows(d_ent, "d_ent"), _chk(d_rel, "d_rel", ndim=2)
    if d_ent.shape != ent.shape or d_rel.shape != rel.shape:
        raise ValueError("gradient buffers must match the table shapes")
    N, R, D = ent.shape[0], rel.shape[0], ent.shape[1]
    nbytes = _lib.load().snag_score_seg_work_bytes(N, R, B, K, D)
    # The workspace and the ticket words belong to ONE stream at a time (the launches hand their header phase and counters to each
    # other in stream order): they are keyed by the stream as well, so two streams never share them.
    key = (ent.device.type, ent.device.index, stream(), N, R, B, K, D)
    if key not in _SEG_WORK:                     # zeroed ONCE; every launch leaves it ready for the next
        while len(_SEG_WORK) >= _SEG_WORK_MAX:
            _SEG_WORK.pop(next(iter(_SEG_WORK)))
        _SEG_WORK[key] = torch.zeros((nbytes + 255) // 256 * 256, device=ent.device, dtype=torch.uint8)
    work = _SEG_WORK[key] = _SEG_WORK.pop(key)   # (re-inserted: most recently used last)
    pos = torch.empty(B, device=ent.device, dtype=f32)
    neg = torch.empty(B, K, device=ent.device, dtype=f32)
    loss_part = torch.empty(B, device=ent.device, dtype=f32)
    loss = torch.empty((), device=ent.device, dtype=f32)
    call("snag_score_margin_fwd_bwd_seg", ptr(ent), ent.stride(0), N, ptr(rel), R, D, ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(margin),
         float(grad_scale), ptr(pos), ptr(neg), ptr(loss_part), ptr(loss), ptr(_ticket(ent.device)), ptr(d_ent), d_ent.stride(0), ptr(d_rel),
         int(write_untouched), ptr(work), work.numel(), stream())
    return loss, pos, neg, loss_part, d_ent, d_rel


def score_margin_fwd_bwd_seg_ln(t_pre, gamma, beta, rel, h, r, t, neg_t, margin, grad_scale=1.0, keep=None, keep_p=0.0, eps=1e-12):
    """The segmented hinge scorer on the table ent = LayerNorm(t_pre) (formed here with torch, as the Mformer's read-out rows are by the
    fusion forward), its entity gradient carried through the LayerNorm inside the per-destination pass:
    -> the 10-tuple (loss, pos, neg, dt [N, D], dtm or None, dgamma, dbeta, dcol [D] (column sums of dtm-or-dt), d_rel, ent [N, D] (the
    LayerNorm output the scorer gathered)).
    C ABI: snag_score_margin_fwd_bwd_seg_ln (the partial slab is reduced here with torch, in slab order)."""
    _rows(t_pre, "t_pre"), _chk(rel, "rel", ndim=2)
    N, D = t_pre.shape
    R = rel.shape[0]
    mean = t_pre.mean(dim=1, keepdim=True)
    var = ((t_pre - mean) ** 2).mean(dim=1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    ent = ((t_pre - mean) * rstd * gamma + beta).contiguous()
    stats = torch.cat([mean, rstd], dim=1).contiguous()
    B, K, ib = _idx_group(h, r, t, neg_t)
    nbytes = _lib.load().snag_score_seg_work_bytes(N, R, B, K, D)
    work = torch.zeros((nbytes + 255) // 256 * 256, device=ent.device, dtype=torch.uint8)
    pos, neg = torch.empty(B, device=ent.device, dtype=f32), torch.empty(B, K, device=ent.device, dtype=f32)
    loss_part, loss = torch.empty(B, device=ent.device, dtype=f32), torch.empty((), device=ent.device, dtype=f32)
    d_rel = torch.empty_like(rel)
    nblk = (N + 15) // 16
    dt = torch.empty(N, D, device=ent.device, dtype=f32)
    dtm = torch.empty(N, D, device=ent.device, dtype=f32) if keep is not None else None
    part = torch.zeros(nblk, 3, D, device=ent.device, dtype=f32)
    tl = _lib.LnBwdTail()
    tl.t, tl.stats, tl.gamma = ptr(t_pre), ptr(stats), ptr(gamma)
    if keep is not None:
        _chk(keep, "keep", dtype=torch.uint8, ndim=2)
        tl.keep, tl.keep_ld, tl.keep_scale, tl.dtm = ptr(keep), keep.stride(0), 1.0 / (1.0 - keep_p), ptr(dtm)
    tl.dt, tl.part = ptr(dt), ptr(part)
    import ctypes
    call("snag_score_margin_fwd_bwd_seg_ln", ptr(ent), N, ptr(rel), R, D, ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(margin), float(grad_scale),
         ptr(pos), ptr(neg), ptr(loss_part), ptr(loss), ptr(d_rel), ptr(work), work.numel(), ctypes.byref(tl), stream())
    sums = part.sum(dim=0)
    return loss, pos, neg, dt, dtm, sums[0], sums[1], sums[2], d_rel, ent


_TOUCHED_WORK = {}


def score_margin_fwd_bwd_touched(ent, rel, h, r, t, neg_t, margin, grad_scale=1.0, d_ent=None, d_rel=None):
    """The segmented hinge scorer for tables far larger than a batch's footprint: pass B walks the compact list of destinations that
    received an entry instead of every counter.  Only touched rows of d_ent are written -- OVERWRITTEN, not accumulated into (allocated zeroed
    when None; the caller keeps the untouched rows zero) -- and EVERY row of d_rel is overwritten (zeros for relations the batch does not
    name): a d_rel holding gradients of an earlier call loses them (since SNAG_ABI_VERSION 2; the earlier form accumulated).  fp32 or
    bfloat16 tables.  -> (loss, pos, neg, loss_part, d_ent, d_rel).  C ABI: snag_score_margin_fwd_bwd_touched(_bf16)."""
    bf16 = ent.dtype == torch.bfloat16
    for x, nm in ((ent, "ent"), (rel, "rel")):
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == (torch.bfloat16 if bf16 else f32) and x.dim() == 2 and x.stride(1) == 1):
            raise ValueError(f"{nm}: expected a 2-D {'bfloat16' if bf16 else 'fp32'} CUDA tensor with unit inner stride")
    if not rel.is_contiguous() or ent.shape[1] != rel.shape[1]:
        raise ValueError("rel must be contiguous and as wide as ent")
    if neg_t is None:
        raise ValueError("score_margin_fwd_bwd_touched needs explicit negative tails")
    B, K, ib = _idx_group(h, r, t, neg_t)
    N, R, D = ent.shape[0], rel.shape[0], ent.shape[1]
    if d_ent is None:
        d_ent = torch.zeros(N, D, device=ent.device, dtype=f32)
    if d_rel is None:
        d_rel = torch.zeros(R, D, device=ent.device, dtype=f32)
    _rows(d_ent, "d_ent"), _chk(d_rel, "d_rel", ndim=2)
    if tuple(d_ent.shape) != (N, D) or tuple(d_rel.shape) != (R, D):
        raise ValueError("gradient buffers must match the table shapes")
    nbytes = _lib.load().snag_score_touched_work_bytes(N, R, B, K, D)
    key = (ent.device.type, ent.device.index, stream(), N, R, B, K, D)
    if key not in _TOUCHED_WORK:                 # zeroed ONCE; every launch leaves it ready for the next
        while len(_TOUCHED_WORK) >= _SEG_WORK_MAX:
            _TOUCHED_WORK.pop(next(iter(_TOUCHED_WORK)))
        _TOUCHED_WORK[key] = torch.zeros((nbytes + 255) // 256 * 256, device=ent.device, dtype=torch.uint8)
    work = _TOUCHED_WORK[key] = _TOUCHED_WORK.pop(key)
    pos = torch.empty(B, device=ent.device, dtype=f32)
    neg = torch.empty(B, K, device=ent.device, dtype=f32)
    loss_part = torch.empty(B, device=ent.device, dtype=f32)
    loss = torch.empty((), device=ent.device, dtype=f32)
    call("snag_score_margin_fwd_bwd_touched_bf16" if bf16 else "snag_score_margin_fwd_bwd_touched", ptr(ent), ent.stride(0), N, ptr(rel), R, D, ptr(h), ptr(r),
         ptr(t), ptr(neg_t), ib, B, K, float(margin), float(grad_scale), ptr(pos), ptr(neg), ptr(loss_part), ptr(loss), ptr(d_ent), d_ent.stride(0),
         ptr(d_rel), ptr(work), work.numel(), stream())
    return loss, pos, neg, loss_part, d_ent, d_rel


def reduce_sum(x):
    """Deterministic sum of a float tensor -> 0-dim tensor.  C ABI: snag_reduce_sum."""
    _chk(x, "x")
    out = torch.empty((), device=x.device, dtype=f32)
    call("snag_reduce_sum", ptr(x), x.numel(), ptr(out), stream())
    return out


ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "gelu_grad": 3, "relu_grad": 4}


def gram(z, alpha=1.0, out=None):
    """S = alpha * Z Z^T, symmetric bit for bit, upper-triangle tiles only (C ABI: snag_gram_f32)."""
    if not (torch.is_tensor(z) and z.is_cuda and z.dtype == f32 and z.dim() == 2 and z.stride(1) == 1):
        raise ValueError("gram: z must be a 2-D fp32 CUDA tensor with unit inner stride")
    n, k = z.shape
    if out is None:
        out = torch.empty(n, n, device=z.device, dtype=f32)
    if tuple(out.shape) != (n, n) or out.stride(1) != 1 or out.dtype != f32:
        raise ValueError("gram: out has the wrong shape")
    call("snag_gram_f32", ptr(z), z.stride(0), n, k, float(alpha), ptr(out), out.stride(0), stream())
    return out


def gemm(a, b, *, a_kcontig=True, b_kcontig=True, bias=None, alpha=1.0, act=None, want_pre_act=False,
         keep_mask=None, keep_p=0.0, add1=None, add2=None, out=None, accumulate=False, splitk=True, aux=None, c_transposed=False):
    """C = add1 + add2 + keep * act(alpha * A.B + bias)   (C ABI: snag_gemm_f32; fp32 via split-bf16 MFMA).

    a: [M,K] if a_kcontig else [K,M];  b: [N,K] if b_kcontig else [K,N]  (2-D, last dim contiguous,
    row stride free).  Returns C [M,N] (and the pre-activation when want_pre_act).
    """
    for t, nm in ((a, "a"), (b, "b")):
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == f32 and t.dim() == 2 and t.stride(1) == 1):
            raise ValueError(f"gemm: {nm} must be a 2-D fp32 CUDA tensor with unit inner stride")
    M, K = a.shape if a_kcontig else (a.shape[1], a.shape[0])
    N, Kb = (b.shape if b_kcontig else (b.shape[1], b.shape[0]))
    if K != Kb:
        raise ValueError(f"gemm: inner sizes differ ({K} vs {Kb})")
    oshape = (N, M) if c_transposed else (M, N)
    if out is None:
        out = torch.empty(oshape, device=a.device, dtype=f32)
    if tuple(out.shape) != oshape or out.stride(1) != 1:
        raise ValueError("gemm: out has the wrong shape")
    d = _lib.GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.a_kcontig, d.b_kcontig = int(a_kcontig), int(b_kcontig)
    d.c_transposed = int(c_transposed)
    d.A, d.lda = ptr(a), a.stride(0)
    d.B, d.ldb = ptr(b), b.stride(0)
    d.C, d.ldc = ptr(out), out.stride(0)
    d.bias = ptr(_chk(bias, "bias", ndim=1)) if bias is not None else None
    d.alpha = float(alpha)
    d.act = ACT[act]
    pre = None
    if want_pre_act:
        pre = torch.empty_strided((M, N), (out.stride(0), 1), device=a.device, dtype=f32)
        d.pre_act = ptr(pre)
    if keep_mask is not None:
        _chk(keep_mask, "keep_mask", dtype=torch.uint8, ndim=2)
        if tuple(keep_mask.shape) != (M, N):
            raise ValueError("gemm: keep_mask must be [M,N]")
        d.keep_mask, d.keep_scale = ptr(keep_mask), 1.0 / (1.0 - keep_p)
    for nm, t in (("add1", add1), ("add2", add2)):
        if t is not None:
            if tuple(t.shape) != (M, N) or t.stride(1) != 1 or t.dtype != f32:
                raise ValueError(f"gemm: {nm} must be [M,N] fp32")
            setattr(d, nm, ptr(t))
            setattr(d, "ld" + nm[-1], t.stride(0))
    d.accumulate = int(accumulate)
    if aux is not None:
        if tuple(aux.shape) != (M, N) or aux.stride(1) != 1 or aux.dtype != f32:
            raise ValueError("gemm: aux must be [M,N] fp32")
        d.aux, d.ldaux = ptr(aux), aux.stride(0)
    ws = None
    if splitk:
        nws = _lib.load().snag_gemm_workspace_floats(M, N, K)
        if nws:
            ws = torch.empty(nws, device=a.device, dtype=f32)
            d.workspace, d.workspace_floats = ptr(ws), nws
    import ctypes
    call("snag_gemm_f32", ctypes.byref(d), stream())
    return (out, pre) if want_pre_act else out


# ---- a6 noise / statistics ------------------------------------------------------------------------
def colstats(x, exclude=None, n_valid=None, want_std=True):
    """Per-column mean and unbiased std over rows with exclude == 0.  C ABI: snag_colstats."""
    _chk(x, "x", ndim=2)
    R, C = x.shape
    if exclude is not None:
        _chk(exclude, "exclude", dtype=torch.uint8, ndim=1)
        if n_valid is None:
            n_valid = R - int(exclude.sum().item())
    elif n_valid is None:
        n_valid = R
    mean = torch.empty(C, device=x.device, dtype=f32)
    std = torch.empty(C, device=x.device, dtype=f32) if want_std else None
    work = torch.empty(_lib.load().snag_colstats_work_floats(C), device=x.device, dtype=f32)
    call("snag_colstats", ptr(x), R, C, ptr(exclude), int(n_valid), ptr(mean), ptr(std), ptr(work), stream())
    return mean, std


def noise_apply(x, mean, std, row_mask, mask_ratio, z=None, sel_index=None, seed=0, out=None):
    """Row-level Gaussian noise injection.  C ABI: snag_noise_apply."""
    _chk(x, "x", ndim=2), _chk(mean, "mean", ndim=1), _chk(std, "std", ndim=1)
    _chk(row_mask, "row_mask", dtype=torch.uint8, ndim=1)
    N, F = x.shape
    if row_mask.shape[0] != N or mean.shape[0] != F or std.shape[0] != F:
        raise ValueError("noise_apply: shape mismatch")
    if z is not None:
        _chk(z, "z", ndim=2)
        _chk(sel_index, "sel_index", dtype=torch.int32, ndim=1)
    if out is None:
        out = torch.empty_like(x)
    call("snag_noise_apply", ptr(x), N, F, ptr(mean), ptr(std), ptr(row_mask), ptr(sel_index), ptr(z),
         int(seed), float(mask_ratio), ptr(out), stream())
    return out


def dropout_keep_mask(shape, p, seed, device="cuda", tick=None, out=None):
    """uint8 keep mask (u >= p) from Philox(seed [+ device tick]).  C ABI: snag_dropout_keep_mask."""
    if out is None:
        out = torch.empty(shape, device=device, dtype=torch.uint8)
    call("snag_dropout_keep_mask", int(seed), ptr(tick), out.numel(), float(p), ptr(out), stream())
    return out


def gaussian_fill(mean, std, N, seed):
    """[N, F] tensor mean + std * z.  C ABI: snag_gaussian_fill."""
    _chk(mean, "mean", ndim=1), _chk(std, "std", ndim=1)
    out = torch.empty(N, mean.shape[0], device=mean.device, dtype=f32)
    call("snag_gaussian_fill", int(seed), N, mean.shape[0], ptr(mean), ptr(std), ptr(out), stream())
    return out


def colsum(x, out=None, accumulate=False):
    """out[c] = sum_r x[r, c] for a 2-D tensor with free row stride.  C ABI: snag_colsum."""
    _rows(x, "x")
    R, C = x.shape
    if out is None:
        out = torch.empty(C, device=x.device, dtype=f32)
    work = torch.empty(256 * C, device=x.device, dtype=f32)
    call("snag_colsum", ptr(x), R, C, x.stride(0), ptr(out), int(accumulate), ptr(work), stream())
    return out


def pack_tokens(embs, g_slot=-1, noise=None, noise_mask=None, mask_ratio=0.0, out=None):
    """torch.stack(embs, dim=1) (+ entity-noise blend in slot g_slot).  C ABI: snag_pack_tokens."""
    import ctypes
    M = len(embs)
    for i, e in enumerate(embs):
        _chk(e, f"embs[{i}]", ndim=2)
    N, D = embs[0].shape
    if out is None:
        out = torch.empty(N, M, D, device=embs[0].device, dtype=f32)
    arr = (ctypes.c_void_p * M)(*[e.data_ptr() for e in embs])
    if noise is not None:
        _chk(noise, "noise", ndim=2), _chk(noise_mask, "noise_mask", dtype=torch.uint8, ndim=1)
    call("snag_pack_tokens", arr, M, N, D, g_slot % M if M else 0, ptr(noise), ptr(noise_mask), float(mask_ratio) * 0.5,
         ptr(out), stream())
    return out


def atten_weight_fwd(x, a, out=None):
    """README joint_way "atten_weight": x [rows, M, D] tokens, a [D] -> (out [rows, D] = sum_m softmax_m(relu(x_m . a)) x_m, aw [rows, 2 M] = the
    weights and pre-activations the backward reads).  C ABI: snag_atten_weight_fwd (arithmetic: model/EVA_tools.py:190-232)."""
    _chk(x, "x", ndim=3), _chk(a, "a", ndim=1)
    rows, M, D = x.shape
    if a.shape[0] != D:
        raise ValueError(f"a has {a.shape[0]} entries, the tokens are {D} wide")
    if out is None:
        out = torch.empty(rows, D, device=x.device, dtype=f32)
    _rows(out, "out")
    if D % 4 == 0:            # (otherwise the C ABI refuses the shape itself)
        _quads("atten_weight_fwd", x=x, a=a, out=out)
    aw = torch.empty(rows, 2 * M, device=x.device, dtype=f32)
    call("snag_atten_weight_fwd", ptr(x), rows, M, D, ptr(a), ptr(out), out.stride(0), ptr(aw), stream())
    return out, aw


def atten_weight_bwd(x, a, aw, d_out, d_x=None, d_a=None, accumulate_da=False):
    """-> (d_x [rows, M, D], d_a [D]).  C ABI: snag_atten_weight_bwd."""
    _chk(x, "x", ndim=3), _chk(a, "a", ndim=1), _chk(aw, "aw", ndim=2), _rows(d_out, "d_out")
    rows, M, D = x.shape
    if tuple(aw.shape) != (rows, 2 * M) or tuple(d_out.shape) != (rows, D):
        raise ValueError(f"aw must be [{rows}, {2 * M}] and d_out [{rows}, {D}], got {tuple(aw.shape)}, {tuple(d_out.shape)}")
    d_x = torch.empty_like(x) if d_x is None else _chk(d_x, "d_x", ndim=3)
    d_a = torch.zeros(D, device=x.device, dtype=f32) if d_a is None else _chk(d_a, "d_a", ndim=1)
    if D % 4 == 0:
        _quads("atten_weight_bwd", x=x, a=a, d_out=d_out, d_x=d_x)
    work = torch.empty(max(1, _lib.load().snag_atten_weight_work_floats(D)), device=x.device, dtype=f32)
    call("snag_atten_weight_bwd", ptr(x), rows, M, D, ptr(a), ptr(aw), ptr(d_out), d_out.stride(0), ptr(d_x), ptr(d_a), int(accumulate_da), ptr(work), stream())
    return d_x, d_a


def adaptive_avg_pool_rows(x, out_hw, view_hw=None, out=None):
    """README POOL=1 (README.md:68,79): every row of x [N, F], viewed as ``view_hw`` = (H, W) with H W = F (default (1, F): the 1-D pool), is
    adaptively average-pooled to ``out_hw`` = (oh, ow) -> [N, oh * ow].  C ABI: snag_adaptive_avg_pool_rows."""
    _rows(x, "x")
    N, F = x.shape
    H, W = (1, F) if view_hw is None else (int(view_hw[0]), int(view_hw[1]))
    oh, ow = (1, int(out_hw)) if isinstance(out_hw, int) else (int(out_hw[0]), int(out_hw[1]))
    if H * W != F:
        raise ValueError(f"view {H} x {W} does not cover rows of {F} features")
    if out is None:
        out = torch.empty(N, oh * ow, device=x.device, dtype=f32)
    _rows(out, "out")
    call("snag_adaptive_avg_pool_rows", ptr(x), N, x.stride(0), H, W, oh, ow, ptr(out), out.stride(0), stream())
    return out


def triple_keys(h, r, t, n_ent, n_rel):
    """Sorted int64 keys (h * n_rel + r) * n_ent + t of a triple set, for the sampler's known-triple filter."""
    k = (h.long() * n_rel + r.long()) * n_ent + t.long()
    return torch.sort(k)[0].contiguous()


def build_filter_mask(h, r, t, n_ent, n_rel, known_sorted):
    """uint8 [B, n_ent]: 1 where (h_b, r_b, x) is a known triple and x != t_b (the filtered ranking protocol).  C ABI: snag_build_filter_mask."""
    for x, nm in ((h, "h"), (r, "r"), (t, "t")):
        _chk(x, nm, dtype=torch.int32, ndim=1)
    _chk(known_sorted, "known_sorted", dtype=torch.int64, ndim=1)
    B = h.shape[0]
    mask = torch.empty(B, n_ent, device=h.device, dtype=torch.uint8)
    call("snag_build_filter_mask", ptr(h), ptr(r), ptr(t), B, int(n_ent), int(n_rel), ptr(known_sorted), known_sorted.numel(), ptr(mask), stream())
    return mask


def sample_negatives(h, r, t, n_neg, n_ent, n_rel, seed, known_sorted=None, head_or_tail=False, tick=None, out=None, distinct=False,
                     ent_list=None):
    """neg_t [B, n_neg] (and neg_h when head_or_tail) drawn on the device.  C ABI: snag_sample_negatives, or -- ``distinct=True``:
    the negatives of a positive are distinct and come from ``ent_list`` (the in-tree sampler's random.sample over the KG's own entity
    list, model/MSNEA_tools.py:59-90) -- snag_sample_negatives_distinct."""
    if distinct or ent_list is not None:
        for x, nm in ((h, "h"), (r, "r"), (t, "t")):
            _chk(x, nm, dtype=torch.int32, ndim=1)
        B = h.shape[0]
        neg_t = out if out is not None else torch.empty(B, n_neg, device=h.device, dtype=torch.int32)
        neg_h = torch.empty(B, n_neg, device=h.device, dtype=torch.int32) if head_or_tail else None
        if known_sorted is not None:
            _chk(known_sorted, "known_sorted", dtype=torch.int64, ndim=1)
        if ent_list is not None:
            _chk(ent_list, "ent_list", dtype=torch.int32, ndim=1)
        call("snag_sample_negatives_distinct", int(seed), ptr(tick), ptr(h), ptr(r), ptr(t), B, int(n_neg), int(n_ent), int(n_rel), int(head_or_tail),
             ptr(known_sorted), known_sorted.numel() if known_sorted is not None else 0, ptr(ent_list),
             ent_list.numel() if ent_list is not None else 0, ptr(neg_h), ptr(neg_t), stream())
        return (neg_h, neg_t) if head_or_tail else neg_t
    for x, nm in ((h, "h"), (r, "r"), (t, "t")):
        _chk(x, nm, dtype=torch.int32, ndim=1)
    B = h.shape[0]
    neg_t = out if out is not None else torch.empty(B, n_neg, device=h.device, dtype=torch.int32)
    neg_h = torch.empty(B, n_neg, device=h.device, dtype=torch.int32) if head_or_tail else None
    if known_sorted is not None:
        _chk(known_sorted, "known_sorted", dtype=torch.int64, ndim=1)
    call("snag_sample_negatives", int(seed), ptr(tick), ptr(h), ptr(r), ptr(t), B, int(n_neg), int(n_ent), int(n_rel), int(head_or_tail),
         ptr(known_sorted), known_sorted.numel() if known_sorted is not None else 0, ptr(neg_h), ptr(neg_t), stream())
    return (neg_h, neg_t) if head_or_tail else neg_t
