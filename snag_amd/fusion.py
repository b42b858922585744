"""Drop-in ``MformerFusion`` running on libsnag_hip.so.

Mirrors the reference operator (model/SNAG_tools.py:23-51): same constructor
``MformerFusion(args, modal_num, with_weight=1)`` reading ``args.num_hidden_layers``,
``.num_attention_heads``, ``.hidden_size``, ``.intermediate_size``, ``.use_intermediate``; same
``forward(embs) -> (joint_emb, joint_emb_fz, hidden_states, weight_norm)``; same state_dict keys
(SURVEY.md par. 8b), so reference checkpoints load with ``load_state_dict``.  The nn.Linear /
nn.LayerNorm children are parameter containers only: all arithmetic is in the HIP library, forward
and backward, and there is no CPU path.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import call, ptr, stream

f32 = torch.float32
JOINT_WAYS = ("Mformer_hd_mean", "Mformer_hd_graph", "Mformer_weight", "learnable_weight")


# --- parameter containers with the reference's module tree (names = state_dict keys) -------------
class _SelfAttention(nn.Module):      # BertSelfAttention, model/SNAG_tools.py:158-167
    def __init__(self, D):
        super().__init__()
        self.query = nn.Linear(D, D)
        self.key = nn.Linear(D, D)
        self.value = nn.Linear(D, D)


class _SelfOutput(nn.Module):         # BertSelfOutput / BertOutput, :211-216, :255-260
    def __init__(self, fan_in, D):
        super().__init__()
        self.dense = nn.Linear(fan_in, D)
        self.LayerNorm = nn.LayerNorm(D, eps=1e-12)


class _Attention(nn.Module):          # BertAttention, :224-228
    def __init__(self, D):
        super().__init__()
        self.self = _SelfAttention(D)
        self.output = _SelfOutput(D, D)


class _Intermediate(nn.Module):       # BertIntermediate, :244-248
    def __init__(self, D, I):
        super().__init__()
        self.dense = nn.Linear(D, I)


class _Layer(nn.Module):              # BertLayer, :268-277 (``output`` exists even when unused)
    def __init__(self, D, I, use_intermediate):
        super().__init__()
        self.attention = _Attention(D)
        if use_intermediate:
            self.intermediate = _Intermediate(D, I)
        self.output = _SelfOutput(I, D)


def _layer_params(layer, ffn):
    """Flat, fixed-order parameter list of one layer (the order of the autograd Function's inputs)."""
    a = layer.attention
    ps = [a.self.query.weight, a.self.key.weight, a.self.value.weight,
          a.self.query.bias, a.self.key.bias, a.self.value.bias,
          a.output.dense.weight, a.output.dense.bias, a.output.LayerNorm.weight, a.output.LayerNorm.bias]
    if ffn:
        ps += [layer.intermediate.dense.weight, layer.intermediate.dense.bias,
               layer.output.dense.weight, layer.output.dense.bias,
               layer.output.LayerNorm.weight, layer.output.LayerNorm.bias]
    return ps


def _stacked(ts):
    """[query, key, value] tensors -> one contiguous [3D, ...] tensor; free when they already are adjacent views."""
    t0 = ts[0]
    step = t0.numel() * t0.element_size()
    if all(t.is_contiguous() for t in ts) and all(ts[i].data_ptr() == t0.data_ptr() + i * step for i in range(3)):
        try:
            return torch.as_strided(t0, (3 * t0.shape[0],) + tuple(t0.shape[1:]), t0.stride())
        except RuntimeError:
            pass
    return torch.cat([t.detach() for t in ts], dim=0)


class _Cfg:
    __slots__ = ("N", "M", "D", "H", "I", "L", "ffn", "drop_p")


def _fill_desc(cfg, layer_tensors, masks, weight_raw, x0, outs, save, work):
    """Build snag_fusion_desc (+ keeps every referenced tensor alive in the returned holder list)."""
    hold = []
    LayerArr = _lib.FusionLayer * cfg.L
    arr = LayerArr()
    for l in range(cfg.L):
        t = layer_tensors[l]
        names = ["wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b"] + (["w1", "b1", "w2", "b2", "ln2_g", "ln2_b"] if cfg.ffn else [])
        for nm, ten in zip(names, t):
            setattr(arr[l], nm, ptr(ten))
        m = masks[l] if masks is not None else {}
        for nm, key in (("keep_probs", "probs"), ("keep_attn_out", "attn_out"), ("keep_ffn_out", "ffn_out")):
            if m.get(key) is not None:
                setattr(arr[l], nm, ptr(m[key]))
        hold.append(t)
    d = _lib.FusionDesc()
    d.N, d.M, d.D, d.H, d.I, d.L = cfg.N, cfg.M, cfg.D, cfg.H, cfg.I, cfg.L
    d.use_intermediate = int(cfg.ffn)
    d.drop_p = float(cfg.drop_p)
    d.layers = ctypes.cast(arr, ctypes.POINTER(_lib.FusionLayer))
    d.weight_raw = ptr(weight_raw)
    d.x0 = ptr(x0)
    if outs is not None:
        d.joint, d.joint_fz, d.hidden, d.weight_norm = (ptr(o) for o in outs)
    d.save, d.save_floats = ptr(save), save.numel()
    if work is not None:
        d.work, d.work_floats = ptr(work), work.numel()
    hold += [arr, masks, weight_raw, x0, outs, save, work]
    return d, hold


def _check_masks(masks, cfg, device):
    if masks is None:
        return None
    if len(masks) != cfg.L:
        raise ValueError(f"drop_masks: expected {cfg.L} per-layer dicts, got {len(masks)}")
    out = []
    shapes = {"probs": (cfg.N, cfg.H, cfg.M, cfg.M), "attn_out": (cfg.N, cfg.M, cfg.D), "ffn_out": (cfg.N, cfg.M, cfg.D)}
    for m in masks:
        mm = {}
        for k, shp in shapes.items():
            t = m.get(k)
            if t is None:
                continue
            if t.dtype != torch.uint8 or tuple(t.shape) != shp or not t.is_cuda:
                raise ValueError(f"drop_masks[{k}] must be a uint8 CUDA tensor of shape {shp}")
            mm[k] = t.contiguous()
        out.append(mm)
    return out


class _FusionFn(torch.autograd.Function):
    """x0 [N,M,D] + parameters -> (joint, joint_fz, hidden, weight_norm) via snag_fusion_fwd / snag_fusion_bwd."""

    @staticmethod
    def forward(ctx, cfg, masks, x0, weight_raw, *params):
        ops._chk(x0, "x0", ndim=3)
        per = 16 if cfg.ffn else 10
        dev = x0.device
        layer_tensors = []
        for l in range(cfg.L):
            p = [q.detach() for q in params[l * per:(l + 1) * per]]
            t = [_stacked(p[0:3]), _stacked(p[3:6])] + [q.contiguous() for q in p[6:]]
            layer_tensors.append(t)
        N, M, D = cfg.N, cfg.M, cfg.D
        outs = (torch.empty(N, M * D, device=dev, dtype=f32), torch.empty(N, M * D, device=dev, dtype=f32),
                torch.empty(N, M, D, device=dev, dtype=f32), torch.empty(N, M, device=dev, dtype=f32))
        lib = _lib.load()
        save = torch.empty(lib.snag_fusion_save_floats(N, M, D, cfg.H, cfg.I, cfg.L, int(cfg.ffn)), device=dev, dtype=f32)
        d, hold = _fill_desc(cfg, layer_tensors, masks, weight_raw.detach(), x0.detach(), outs, save, None)
        call("snag_fusion_fwd", ctypes.byref(d), stream())
        ctx.cfg, ctx.masks, ctx.layer_tensors, ctx.save = cfg, masks, layer_tensors, save
        ctx.save_for_backward(x0, weight_raw)
        ctx.param_shapes = [tuple(p.shape) for p in params]
        return outs

    @staticmethod
    def backward(ctx, d_joint, d_joint_fz, d_hidden, d_wnorm):
        cfg = ctx.cfg
        x0, weight_raw = ctx.saved_tensors
        dev = x0.device
        N, M, D, I = cfg.N, cfg.M, cfg.D, cfg.I
        lib = _lib.load()
        work = torch.empty(lib.snag_fusion_work_floats(N, M, D, cfg.H, I, cfg.L, int(cfg.ffn)), device=dev, dtype=f32)
        d, hold = _fill_desc(cfg, ctx.layer_tensors, ctx.masks, weight_raw.detach(), x0.detach(), None, ctx.save, work)
        b = _lib.FusionBwdDesc()
        b.f = d
        ups = []
        for nm, g in (("d_joint", d_joint), ("d_joint_fz", d_joint_fz), ("d_hidden", d_hidden), ("d_weight_norm", d_wnorm)):
            if g is not None:
                g = g.contiguous()
                ups.append(g)
                setattr(b, nm, ptr(g))
        d_x0 = torch.empty_like(x0)
        b.d_x0 = ptr(d_x0)
        GradArr = _lib.FusionLayerGrad * cfg.L
        garr = GradArr()
        grads = []
        for l in range(cfg.L):
            shapes = [(3 * D, D), (3 * D,), (D, D), (D,), (D,), (D,)]
            names = ["wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b"]
            if cfg.ffn:
                shapes += [(I, D), (I,), (D, I), (D,), (D,), (D,)]
                names += ["w1", "b1", "w2", "b2", "ln2_g", "ln2_b"]
            gl = [torch.empty(s, device=dev, dtype=f32) for s in shapes]
            for nm, t in zip(names, gl):
                setattr(garr[l], nm, ptr(t))
            grads.append(gl)
        b.grads = ctypes.cast(garr, ctypes.POINTER(_lib.FusionLayerGrad))
        d_raw = torch.empty(6, device=dev, dtype=f32)
        b.d_weight_raw = ptr(d_raw)
        b.accumulate = 0
        call("snag_fusion_bwd", ctypes.byref(b), stream())
        out = []
        for gl in grads:
            wq, bq = gl[0], gl[1]
            out += [wq[0:D], wq[D:2 * D], wq[2 * D:3 * D], bq[0:D], bq[D:2 * D], bq[2 * D:3 * D]] + gl[2:]
        return (None, None, d_x0, d_raw, *out)


class MformerFusion(nn.Module):
    """HIP drop-in for the reference ``MformerFusion`` (model/SNAG_tools.py:23-51).

    Extra, optional surface (not in the reference): ``joint_way`` selects the MKGC read-out
    returned by :meth:`readout` (README.md:51); ``forward(embs, drop_masks=...)`` accepts explicit
    inverted-dropout keep masks (parity tests) instead of drawing them.
    """

    DROP_P = 0.1     # nn.Dropout(0.1) hard-coded at model/SNAG_tools.py:169,216,260

    def __init__(self, args, modal_num, with_weight=1, joint_way="Mformer_weight"):
        super().__init__()
        self.args = args
        self.modal_num = modal_num
        D, H = int(args.hidden_size), int(args.num_attention_heads)
        if D % H != 0:      # reference: assert at model/SNAG_tools.py:161
            raise AssertionError(f"hidden_size {D} is not a multiple of num_attention_heads {H}")
        if joint_way not in JOINT_WAYS:
            raise ValueError(f"joint_way must be one of {JOINT_WAYS}, got {joint_way!r}")
        self.joint_way = joint_way
        self.weight_raw = nn.Parameter(torch.ones(6, requires_grad=True))           # :28-29
        self.fusion_layer = nn.ModuleList(
            [_Layer(D, int(args.intermediate_size), bool(args.use_intermediate)) for _ in range(int(args.num_hidden_layers))])

    # -- helpers ---------------------------------------------------------------------------------
    def _cfg(self, N, M):
        a = self.args
        c = _Cfg()
        c.N, c.M, c.D, c.H = N, M, int(a.hidden_size), int(a.num_attention_heads)
        c.I, c.L, c.ffn = int(a.intermediate_size), int(a.num_hidden_layers), bool(a.use_intermediate)
        c.drop_p = self.DROP_P
        return c

    def _draw_masks(self, cfg, device):
        """Inverted-dropout keep masks of a training forward: ONE Philox launch over one byte buffer (snag_dropout_keep_mask), seeded
        from torch's CPU generator -- so ``torch.manual_seed`` makes a run reproducible, as it does for the reference's nn.Dropout."""
        p = cfg.drop_p
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        al = lambda n: (n + 63) // 64 * 64
        n_probs, n_tok = al(cfg.N * cfg.H * cfg.M * cfg.M), al(cfg.N * cfg.M * cfg.D)
        per_layer = n_probs + n_tok * (2 if cfg.ffn else 1)
        buf = ops.dropout_keep_mask((cfg.L * per_layer,), p, seed, device=device)
        masks = []
        for l in range(cfg.L):
            o = l * per_layer
            m = {"probs": buf[o:o + cfg.N * cfg.H * cfg.M * cfg.M].view(cfg.N, cfg.H, cfg.M, cfg.M),
                 "attn_out": buf[o + n_probs:o + n_probs + cfg.N * cfg.M * cfg.D].view(cfg.N, cfg.M, cfg.D)}
            if cfg.ffn:
                m["ffn_out"] = buf[o + n_probs + n_tok:o + n_probs + n_tok + cfg.N * cfg.M * cfg.D].view(cfg.N, cfg.M, cfg.D)
            masks.append(m)
        return masks

    def forward_stacked(self, x0, drop_masks=None):
        """Same as forward() for tokens that are already stacked [N, M, D]."""
        _lib.require_gpu()
        if x0.dim() != 3 or x0.shape[2] != int(self.args.hidden_size):
            raise ValueError(f"x0 must be [N, M, {self.args.hidden_size}], got {tuple(x0.shape)}")
        if x0.shape[1] > 6:
            raise ValueError("at most 6 modality tokens")
        cfg = self._cfg(x0.shape[0], x0.shape[1])
        if drop_masks is None and self.training:
            drop_masks = self._draw_masks(cfg, x0.device)
        masks = _check_masks(drop_masks, cfg, x0.device)
        params = []
        for layer in self.fusion_layer:
            params += _layer_params(layer, cfg.ffn)
        return _FusionFn.apply(cfg, masks, x0.contiguous(), self.weight_raw, *params)

    def forward(self, embs, drop_masks=None):
        embs = [e for e in embs if e is not None]                                    # :33
        if not embs:
            raise ValueError("MformerFusion.forward: every modality embedding is None")
        return self.forward_stacked(torch.stack(embs, dim=1), drop_masks)            # :36

    def readout(self, joint, joint_fz, hidden, graph_index=-1):
        """MKGC ``joint_way`` read-out (a9): hd_mean = hidden.mean(1), hd_graph = hidden[:, g]."""
        if self.joint_way == "Mformer_hd_mean":
            return _TokenMeanFn.apply(hidden)
        if self.joint_way == "Mformer_hd_graph":
            return hidden[:, graph_index, :]
        return joint if self.joint_way == "Mformer_weight" else joint_fz


class _TokenMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden):
        ops._chk(hidden, "hidden", ndim=3)
        N, M, D = hidden.shape
        out = torch.empty(N, D, device=hidden.device, dtype=f32)
        call("snag_token_mean", ptr(hidden), N, M, D, ptr(out), stream())
        ctx.M = M
        return out

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.M).unsqueeze(1).expand(-1, ctx.M, -1)
