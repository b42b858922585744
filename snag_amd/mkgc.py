"""MKGC hot path (BASELINE.json north_star): per-modality projections -> modality tokens ->
Mformer fusion -> ``joint_way`` read-out -> margin score of 1 positive + NEG_NUM negative tails,
forward and backward, as one stream of HIP launches over preallocated buffers.

The reference's MKGC driver is not in the tree (SURVEY.md par. 0.1); this engine follows the README
knobs (README.md:43-80: ``num_proj``, ``use_intermediate``, ``joint_way``, noise/mask ratio,
``EMB_DIM``, ``NUM_BATCH``, ``MARGIN``, ``NEG_NUM``) and re-uses, operator by operator, what IS in the
tree: ``MultiModalEncoder.forward`` (model/SNAG_tools.py:122-156) with the GAT removed,
``MformerFusion`` (:23-51), the noise injection (model/SNAG.py:66-98) and the TransE-L2^2 margin
scorer (model/MSNEA.py:110-122).

Every learnable tensor is a view into ONE flat fp32 buffer (and its gradient into one flat gradient
buffer), so the data-parallel exchange is a single RCCL all-reduce and the optimiser a single launch.
"""
import ctypes
import os
import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream

f32 = torch.float32

@dataclass
class MKGCConfig:
    n_ent: int = 15000
    n_rel: int = 558
    dim: int = 128                      # EMB_DIM
    feat_dims: List[int] = field(default_factory=lambda: [4096, 768])   # visual, text raw feature widths
    num_proj: int = 2                   # README "num_proj: 1 / 2"
    joint_way: str = "Mformer_hd_graph"
    num_heads: int = 2
    num_layers: int = 1
    use_intermediate: int = 0
    intermediate_size: int = 400
    margin: float = 12.0                # MARGIN
    loss: str = "hinge"                 # "hinge" (north_star) | "self_adversarial" (optional a7 epilogue, --adv_temp)
    adv_temp: float = 1.0
    neg_num: int = 32                   # NEG_NUM
    add_noise: int = 1                  # NOISE
    noise_ratio: float = 0.2
    mask_ratio: float = 0.7
    dropout: float = 0.1                # nn.Dropout(0.1) inside the Bert blocks (SNAG_tools.py:169,216)
    proj_dropout: float = 0.0           # num_proj=2: ProjectionHead's F.dropout between ReLU and l2 (model/layers.py:142-147); its rate is a
                                        # constructor argument there and the MKGC README names none, hence off by default
    lr: float = 1e-4                    # LR
    weight_decay: float = 1e-4
    readout_only: bool = True           # Mformer_hd_graph consumes hidden[:, g, :] alone: the last layer computes the query / attention row /
                                        # output projection / LayerNorm (/ FFN) for the structural token only (same values, 2/3 of the GEMM work)
    fold_proj: bool = True              # read-out-only, one layer, num_proj = 2: the projected tokens enter the layer through their keys /
                                        # values alone, so KV_m = mid_m (Wkv l2_m)^T + b is computed straight from the first projection's
                                        # output -- the second projection's GEMM (and its share of the backward) is folded into the
                                        # key / value projection (fp32 re-association of (mid l2^T) Wkv^T: equal to rounding, not bit-equal)
    scorer: str = "segmented"           # training form of the hinge scorer: "segmented" = per-triple pass + per-destination pass, no float atomics,
                                        # bit-reproducible, gradient rows written (snag_score_margin_fwd_bwd_seg); "atomic" = one launch that
                                        # scatter-adds with fp32 atomics (snag_score_margin_fwd_bwd)
    deterministic: bool = False         # with scorer="atomic": fixed-point integer atomics instead (bit-reproducible, two launches)
    clip: float = 1.0                   # torch.nn.utils.clip_grad_norm_ of the reference's train loop (main.py:266-268, --clip default 1.0); 0 = off
    use_pool: int = 0                   # README POOL=1 / ``use_pool`` (README.md:68,79): every raw feature table is average-pooled ONCE at load "for
                                        # dimensionality reduction and uniformity in dimensions" -- F_m -> pool_dim columns (snag_adaptive_avg_pool_rows);
                                        # tables already at most pool_dim wide are kept.  No code in the tree: PARITY UNPINNED (oracle.adaptive_avg_pool_rows)
    pool_dim: int = 256
    pool_views: Optional[List[Optional[tuple]]] = None     # per modality (H, W, oh, ow): pool the row as an H x W image to oh x ow instead of the
                                        # 1-D pool (e.g. (64, 64, 16, 16) for a 4096-wide and (12, 64, 4, 64) for a 768-wide table); None = 1-D
    seed: int = 1234

    @property
    def n_tokens(self):
        return len(self.feat_dims) + 1


def cosine_lr(step, total_steps, warmup_steps, base_lr, num_cycles=0.5):
    """Learning rate of the reference's ``--scheduler cos`` (run_snag.sh; src/utils.py:75-76 -> transformers'
    ``get_cosine_schedule_with_warmup``) after ``step`` optimiser steps: linear warm-up, then half a cosine to zero.  The engine reads
    ``cfg.lr`` at every ``optimizer_step``, so ``eng.cfg.lr = cosine_lr(...)`` before the step is the whole integration."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    progress = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * progress)))


def linear_lr(step, total_steps, warmup_steps, base_lr):
    """``--scheduler linear`` (src/utils.py:71-74 -> ``get_linear_schedule_with_warmup``)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    return base_lr * max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))


def _align(n, a=64):
    return (n + a - 1) // a * a


class MKGCHotPath:
    """Owns parameters, activations and the launch sequence of one training / scoring step."""

    def __init__(self, cfg: MKGCConfig, feats: List[torch.Tensor], device="cuda", shard=None):
        """``shard`` (snag_amd.parallel.RowShard): data-parallel runs own the entity ROWS in equal shards -- this rank projects and
        fuses rows [shard.lo, shard.lo + shard.n_loc) only, the [N, dim] read-out table is all-gathered before scoring and its
        gradient reduce-scattered back (SURVEY.md par. 8e).  Default: one rank owning every row.  ``feats``: the full [N, F_m]
        tables (sliced here) or this rank's rows only."""
        _lib.require_gpu()
        from dataclasses import replace
        from .parallel import RowShard
        self.dev = torch.device(device)
        self.lib = _lib.load()
        self.pooled_from = None
        if cfg.use_pool:
            # README POOL=1: the step never sees the raw tables again -- only the pooled ones stay in HBM, and the engine's configuration
            # from here on is the pooled widths (a checkpoint records them with use_pool = 0 and ``pooled_from`` = the raw widths)
            feats, dims = self._pool_features(cfg, feats)
            self.pooled_from = list(cfg.feat_dims)
            cfg = replace(cfg, feat_dims=dims, use_pool=0, pool_views=None)
        self.cfg = cfg
        N, R, D, M = cfg.n_ent, cfg.n_rel, cfg.dim, cfg.n_tokens
        self.shard = shard if shard is not None else RowShard(N)
        if self.shard.n_rows != N:
            raise ValueError(f"shard covers {self.shard.n_rows} rows, the entity table has {N}")
        Nl, Ns, lo = self.shard.n_loc, self.shard.n_shard, self.shard.lo
        # README.md:51 joint_way: the two hidden-state read-outs ([N, dim]) and the two weighted concatenations of the normalised INPUT
        # tokens ([N, M*dim]: Mformer_weight = attention-derived weights, learnable_weight = softmax(weight_raw); a1's joint / joint_fz).
        # "atten_weight": non-transformer attention over the tokens -- one learnable vector a [D], w = softmax_m(relu(x_m . a)), read-out
        # sum_m w_m x_m [N, dim] (the arithmetic of EVA_tools.SelfAttention, model/EVA_tools.py:190-232, the only code of that name in the
        # tree); the Mformer is not run and owns no parameters in that mode.
        if cfg.joint_way not in ("Mformer_hd_graph", "Mformer_hd_mean", "Mformer_weight", "atten_weight", "learnable_weight"):
            raise ValueError("joint_way must be Mformer_hd_graph, Mformer_hd_mean, Mformer_weight, atten_weight or learnable_weight")
        self.atten = cfg.joint_way == "atten_weight"
        if self.atten and (cfg.dim % 4 or cfg.dim > 512 or cfg.n_tokens > 6):
            raise ValueError(f"atten_weight: hidden size must be a multiple of 4 up to 512 with at most 6 tokens, got dim={cfg.dim}, tokens={cfg.n_tokens}")
        self.n_layers = 0 if self.atten else cfg.num_layers
        if cfg.scorer not in ("segmented", "atomic"):
            raise ValueError(f"scorer must be 'segmented' or 'atomic', got {cfg.scorer!r}")
        self.concat = cfg.joint_way in ("Mformer_weight", "learnable_weight")
        self.sdim = M * D if self.concat else D               # width of the read-out the scorer gathers (and of the relation rows)
        if self.sdim > 512:
            raise ValueError(f"read-out width {self.sdim} > 512: the scorer's row registers stop there")
        if len(feats) != len(cfg.feat_dims):
            raise ValueError("one feature table per entry of feat_dims")
        self.feats = []
        self.feat_stats = None
        for f, fd in zip(feats, cfg.feat_dims):
            if f.dtype != f32 or tuple(f.shape) not in ((N, fd), (Nl, fd)):
                raise ValueError(f"feature table must be fp32 [{N}, {fd}] (or this rank's [{Nl}, {fd}]), got {tuple(f.shape)}")
            if self.shard.sharded and cfg.add_noise and tuple(f.shape) != (N, fd):
                raise ValueError("the noise statistics are those of the WHOLE clean table (model/SNAG.py:77-84): pass the full feature tables")
        if self.shard.sharded and cfg.add_noise:
            from . import ops
            self.feat_stats = [ops.colstats(f.to(self.dev).contiguous()) for f in feats]      # before slicing: every rank holds the same constants
        for f, fd in zip(feats, cfg.feat_dims):
            if tuple(f.shape) == (N, fd) and Nl != N:
                f = f[lo:lo + Nl]
            self.feats.append(f.to(self.dev).contiguous())
        # The first projection of every modality is X W^T with X [N, F] and F in the thousands: both operands are streamed in the layout
        # they are stored in -- the row-major table (K-contiguous A) and the nn.Linear weight [D, F] (K-contiguous B).  Rounds 1-2 kept a
        # transposed copy of every feature table for this GEMM (row-contiguous operands measured 3 % faster inside the step: 101.6 vs
        # 105.0 us for the visual projection + its reduce); it cost 290 MB, a full rewrite at every noise refresh (the refresh now touches
        # only the rows in the new or the previous mask) and a second layout for the weights.
        self.g_slot = M - 1                                   # structural token last
        self.tail_riders = os.environ.get("SNAG_TAIL_RIDERS", "1") != "0"      # (the variable: tools/ab_env.sh A/B only) one rank: the Mformer backward's last reduce rides in the first projections' weight-gradient launch (False: a launch of its own; the same bits -- tests)
        self.co_launch = True        # folded key / value products ride in snag_fusion_fwd's launches (False: launches of their own; the same bits -- tests)
        self.force_one_launch = False    # True: the one-launch read-out forward / fused backward below the library's row thresholds too (tests)
        I, L, ffn = cfg.intermediate_size, self.n_layers, bool(cfg.use_intermediate)

        # ---- flat parameter layout ---------------------------------------------------------------
        self.slices = {}
        off = 0

        def add(name, *shape, reserve=0):
            nonlocal off
            n = int(math.prod(shape))
            self.slices[name] = (off, tuple(shape))
            off = _align(off + max(n, reserve))

        # row-sharded: the entity table's region is padded to whole shards (n_pad rows; the pad rows hold zero parameters and zero
        # gradients for ever), so that the in-place all-gather of the owners' gradient rows applies whatever N and the rank count are
        add("ent", N, D, reserve=self.shard.n_pad * D if self.shard.sharded else 0)
        add("rel", R, self.sdim)
        for i, fd in enumerate(cfg.feat_dims):
            if cfg.num_proj == 1:
                add(f"proj{i}.weight", D, fd)      # nn.Linear layout [out, in]
                add(f"proj{i}.bias", D)
            else:
                add(f"proj{i}.l1", D, fd)
                add(f"proj{i}.l2", D, D)
        add("weight_raw", 6)
        if self.atten:
            add("atten.a", D)
        for l in range(L):
            add(f"layer{l}.wqkv", 3 * D, D); add(f"layer{l}.bqkv", 3 * D)
            add(f"layer{l}.wo", D, D); add(f"layer{l}.bo", D)
            add(f"layer{l}.ln1_g", D); add(f"layer{l}.ln1_b", D)
            if ffn:
                add(f"layer{l}.w1", I, D); add(f"layer{l}.b1", I)
                add(f"layer{l}.w2", D, I); add(f"layer{l}.b2", D)
                add(f"layer{l}.ln2_g", D); add(f"layer{l}.ln2_b", D)
        self.n_params = off
        # the reference's parameter groups (set_optim, src/utils.py:47-55): biases and weight_raw are not decayed
        # (LayerNorm.weight is, as there: only the substrings 'bias' / 'weight_raw' / 'multi_loss_layer' opt out)
        def undecayed(name):
            leaf = name.split(".")[-1]
            return name == "weight_raw" or leaf == "bias" or (leaf.startswith("b") and leaf[1:] in ("qkv", "o", "1", "2")) or leaf.endswith("_b")
        names = sorted(self.slices, key=lambda k: self.slices[k][0])
        ends = [self.slices[k][0] for k in names[1:]] + [off]
        self.param_groups = [(k, e, 0.0 if undecayed(k) else cfg.weight_decay) for k, e in zip(names, ends)]
        self.flat = torch.zeros(off, device=self.dev, dtype=f32)
        self.gflat = torch.zeros(off, device=self.dev, dtype=f32)
        self.exp_avg = torch.zeros(off, device=self.dev, dtype=f32)
        self.exp_avg_sq = torch.zeros(off, device=self.dev, dtype=f32)
        self.p = {k: self.flat[o:o + math.prod(s)].view(s) for k, (o, s) in self.slices.items()}
        self.g = {k: self.gflat[o:o + math.prod(s)].view(s) for k, (o, s) in self.slices.items()}
        self.g_ent_pad = self.gflat[:self.shard.n_pad * D].view(self.shard.n_pad, D) if self.shard.sharded else self.g["ent"]
        self._init_params()
        self._seg_end = torch.tensor([e for _, e, _ in self.param_groups], device=self.dev, dtype=torch.int64)
        self._seg_wd = torch.tensor([w for _, _, w in self.param_groups], device=self.dev, dtype=f32)
        self._seg_lr = torch.ones(len(self.param_groups), device=self.dev, dtype=f32)
        if cfg.deterministic:
            self._acc_ent = torch.zeros(N, self.sdim, device=self.dev, dtype=torch.int64)
            self._acc_rel = torch.zeros(R, self.sdim, device=self.dev, dtype=torch.int64)
        self._clip = torch.ones(2, device=self.dev, dtype=f32)          # [clip coefficient, gradient norm] of the last step
        self._clip_work = torch.empty(1024, device=self.dev, dtype=f32)

        # ---- activations / scratch ---------------------------------------------------------------
        dev = self.dev
        # Everything between the raw features and the read-out lives on this rank's Nl rows (Nl = N with one rank)
        self.x0 = torch.empty(Nl, M, D, device=dev)
        # read-out-only last layer: hidden IS the [rows, D] read-out, and its gradient the compact scatter target of the scorer
        self.ro = bool(cfg.readout_only) and cfg.joint_way == "Mformer_hd_graph"
        if self.shard.sharded and not (self.ro or self.concat or self.atten or cfg.joint_way == "Mformer_hd_mean"):
            raise ValueError("row-sharded fusion exchanges a compact [rows, dim] read-out: use readout_only=True with Mformer_hd_graph")
        # compact read-outs hold Ns = shard.n_shard rows (the collectives move equal chunks; rows past Nl stay zero)
        # (atten_weight: ``hidden`` is the [rows, D] attention read-out, ``aw`` its weights / pre-activations for the backward)
        self.hidden = torch.zeros((Ns, D) if (self.ro or self.atten) else (Nl, M, D), device=dev)
        self.d_hidden = None if (self.ro or self.concat or self.atten) else torch.empty(Nl, M, D, device=dev)
        self.aw = torch.empty(max(Nl, 1), 2 * M, device=dev) if self.atten else None
        self._aw_work = torch.empty(max(1, self.lib.snag_atten_weight_work_floats(D)), device=dev) if self.atten else None
        # concatenation read-outs: joint (or joint_fz) [rows, M*D] straight from the fusion epilogue, its gradient the scatter target
        self.joint = torch.zeros(Ns, M * D, device=dev) if self.concat else None
        self.d_x0 = torch.empty(Nl, M, D, device=dev)
        self.mid = [torch.empty(Nl, D, device=dev) for _ in cfg.feat_dims] if cfg.num_proj == 2 else []
        self.d_mid = [torch.empty(Nl, D, device=dev) for _ in cfg.feat_dims] if cfg.num_proj == 2 else None
        self.repr_mean = torch.zeros(Ns, D, device=dev) if cfg.joint_way == "Mformer_hd_mean" else None
        # compact [rows, D] gradient of the read-out (hd_mean, read-out-only hd_graph): zeroed once here; afterwards whoever READS it
        # in the backward clears it behind itself (snag_fusion_bwd zero_d_hidden / snag_token_mean_bwd zero_src), so the fused
        # scorer of the next step finds it clean without a fill launch
        self.d_repr = torch.zeros(Ns, self.sdim, device=dev) if (self.ro or self.concat or self.atten or cfg.joint_way == "Mformer_hd_mean") else None
        self._d_repr_clean = True
        # row-sharded: the table the scorer reads is the all-gather of every rank's read-outs, its gradient is reduce-scattered back
        self.repr_full = torch.zeros(self.shard.n_pad, self.sdim, device=dev) if self.shard.sharded else None
        self.d_repr_full = torch.zeros(self.shard.n_pad, self.sdim, device=dev) if self.shard.sharded else None
        lib = self.lib
        self.save = torch.empty(lib.snag_fusion_save_floats(Nl, M, D, cfg.num_heads, I, L, int(ffn)), device=dev)
        ws = lib.snag_fusion_work_floats(Nl, M, D, cfg.num_heads, I, L, int(ffn))
        for fd in cfg.feat_dims:
            # (the first projections and their weight gradients run two modalities at a time, each in its own half of the scratch)
            ws = max(ws, 2 * (lib.snag_gemm_workspace_floats(fd, D, Nl) + 64), 2 * (lib.snag_gemm_workspace_floats(Nl, D, fd) + 64), 256 * max(2 * D, fd))
        ws = max(ws, 2 * (lib.snag_gemm_workspace_floats(D, D, Nl) + 64), lib.snag_gemm_workspace_floats(Nl, D, D) + 64)
        self.work = torch.empty(ws, device=dev)
        # the second projections' weight gradients keep their K-split slabs here until the first projections' reduce launch takes them along
        self.fold = bool(cfg.fold_proj) and self.ro and L == 1 and cfg.num_proj == 2 and len(cfg.feat_dims) == M - 1
        self.work_l2 = torch.empty(2 * (lib.snag_gemm_workspace_floats(2 * D if self.fold else D, D, Nl) + 64), device=dev) if cfg.num_proj == 2 else None
        # projection folding: Wf_m = Wkv l2_m [2D, D] (remade every step) and its gradient
        self.wf = [torch.empty(2 * D, D, device=dev) for _ in cfg.feat_dims] if self.fold else None
        self.d_wf = [torch.empty(2 * D, D, device=dev) for _ in cfg.feat_dims] if self.fold else None
        H = cfg.num_heads
        # all dropout keep-masks of the step live in ONE byte buffer (views below), so one Philox launch draws them all
        n_probs, n_tok = _align(Nl * H * M * M), _align(Nl * M * D)
        per_layer = n_probs + n_tok * (2 if ffn else 1)
        self.keep_all = torch.ones(max(1, L * per_layer), device=dev, dtype=torch.uint8)
        self.keep_probs = [self.keep_all[l * per_layer:l * per_layer + Nl * H * M * M].view(Nl, H, M, M) for l in range(L)]
        self.keep_attn = [self.keep_all[l * per_layer + n_probs:l * per_layer + n_probs + Nl * M * D].view(Nl * M, D) for l in range(L)]
        self.keep_ffn = ([self.keep_all[l * per_layer + n_probs + n_tok:l * per_layer + n_probs + n_tok + Nl * M * D].view(Nl * M, D) for l in range(L)]
                         if ffn else None)
        # ProjectionHead dropout masks (num_proj=2, proj_dropout > 0): one [rows, D] keep mask per modality, drawn per step
        self.keep_proj = (torch.ones(len(cfg.feat_dims), max(Nl, 1), D, device=dev, dtype=torch.uint8)
                          if (cfg.num_proj == 2 and cfg.proj_dropout > 0) else None)
        # Every mask that covers entity rows is a REGION of a global Philox stream, keyed on the global index of its first element
        # (snag_bernoulli_regions): a rank that owns rows [lo, lo + Nl) draws exactly the bytes the single rank draws for them, so the
        # G-rank step equals the 1-rank step with dropout and noise on (SURVEY.md par. 8e)
        regs = []
        for l in range(L):
            regs.append((cfg.seed * 131 + 17 * l + 1, lo * H * M * M, Nl * H * M * M, cfg.dropout, self.keep_probs[l]))
            # (the read-out-only last layer consumes the output-dropout masks of the read-out token alone: D bytes of every M D -- the
            # same bytes the dense region draws for them, a third of the Philox work)
            only_g = ((M * D, self.g_slot * D, D),) if (self.ro and l == L - 1 and D % 4 == 0) else ()
            regs.append((cfg.seed * 131 + 17 * l + 2, lo * M * D, Nl * M * D, cfg.dropout, self.keep_attn[l]) + only_g)
            if ffn:
                regs.append((cfg.seed * 131 + 17 * l + 3, lo * M * D, Nl * M * D, cfg.dropout, self.keep_ffn[l]) + only_g)
        if len(regs) > _lib.MAX_MASK_REGIONS:
            raise NotImplementedError(f"{len(regs)} dropout masks per step: the step prologue draws at most {_lib.MAX_MASK_REGIONS} (num_layers <= "
                                      f"{_lib.MAX_MASK_REGIONS // (3 if ffn else 2)})")
        self._drop_regions = self._regions(regs)
        self._proj_regions = (self._regions([(cfg.seed * 131 + 7 + i, lo * D, Nl * D, cfg.proj_dropout, self.keep_proj[i]) for i in range(len(cfg.feat_dims))])
                              if self.keep_proj is not None else None)
        self.tick = torch.zeros(1, device=dev, dtype=torch.int64)      # device-side step counter
        # noise state (model/SNAG.py:86-98)
        self.noisy = [f.clone() for f in self.feats]
        self.ent_noise = torch.zeros(N, D, device=dev)                 # whole table on every rank (same seed): a rank reads its rows
        self.ent_noise_mask = torch.zeros(N, device=dev, dtype=torch.uint8)
        # row masks of the feature-noise refresh, two generations per modality (the refresh rewrites only rows in the new or the old mask)
        self.row_mask = [[torch.zeros(max(Nl, 1), device=dev, dtype=torch.uint8) for _ in range(2)] for _ in cfg.feat_dims]
        self.noise_active = False
        self.noise_epoch = 0
        self.training = True
        self.probe = None                # optional {tag: timer with .wrap(fn)}: bench.py's per-kernel HIP-event timers
        self.l1_grouped = False          # the two first projections ran as one grouped launch (bench.py's roofline object describes it)
        self.marks = None                # optional list: (phase name, HIP event on the launch stream) appended at the step's phase boundaries
        self._loss_part = None
        self._score_bwd_scale = None
        self._ticket = torch.zeros(_lib.SCORE_TICKET_WORDS, device=dev, dtype=torch.int32)      # last-block ticket of the fused scorer (self-resetting)
        self.loss = torch.zeros((), device=dev)
        self._build_descs()

    # ------------------------------------------------------------------------------------------
    def _pool_features(self, cfg, feats):
        """README POOL=1 (README.md:68,79): each raw table wider than ``pool_dim`` -> [rows, pool_dim] by one adaptive average pool on the
        device (1-D, or the (H, W) -> (oh, ow) image view ``pool_views`` names for that modality)."""
        from . import ops
        out, dims = [], []
        views = cfg.pool_views or [None] * len(feats)
        if len(views) != len(feats):
            raise ValueError("pool_views: one entry (or None) per feature table")
        for f, fd, view in zip(feats, cfg.feat_dims, views):
            if f.dtype != f32 or f.dim() != 2 or f.shape[1] != fd:
                raise ValueError(f"feature table must be fp32 [rows, {fd}], got {tuple(f.shape)} {f.dtype}")
            if view is not None:
                H, W, oh, ow = (int(v) for v in view)
                f = ops.adaptive_avg_pool_rows(f.to(self.dev).contiguous(), (oh, ow), view_hw=(H, W))
            elif fd > cfg.pool_dim:
                f = ops.adaptive_avg_pool_rows(f.to(self.dev).contiguous(), int(cfg.pool_dim))
            out.append(f)
            dims.append(int(f.shape[1]))
        return out, dims

    def _init_params(self):
        """Reference initialisers: entity table N(0, 1/sqrt(N)) (SNAG_tools.py:85), nn.Linear defaults, LN 1/0, weight_raw 1."""
        cfg = self.cfg
        g = torch.Generator().manual_seed(cfg.seed)
        N, D = cfg.n_ent, cfg.dim

        def lin(shape_out_in):
            bound = 1.0 / math.sqrt(shape_out_in[1])            # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +)
            return (torch.rand(shape_out_in, generator=g) * 2 - 1) * bound

        def bias(n, fan_in):
            return (torch.rand(n, generator=g) * 2 - 1) / math.sqrt(fan_in)

        self.p["ent"].copy_(torch.randn(N, D, generator=g) / math.sqrt(N))
        self.p["rel"].copy_(torch.randn(cfg.n_rel, self.sdim, generator=g) * math.sqrt(2.0 / (cfg.n_rel + self.sdim)))   # xavier normal
        for i, fd in enumerate(cfg.feat_dims):
            if cfg.num_proj == 1:
                self.p[f"proj{i}.weight"].copy_(lin((D, fd)))
                self.p[f"proj{i}.bias"].copy_(bias(D, fd))
            else:
                self.p[f"proj{i}.l1"].copy_(lin((D, fd)))
                self.p[f"proj{i}.l2"].copy_(lin((D, D)))
        self.p["weight_raw"].fill_(1.0)
        if self.atten:          # nn.init.xavier_uniform on the [1, D] parameter (model/EVA_tools.py:197-200): U(+-sqrt(6 / (1 + D)))
            self.p["atten.a"].copy_((torch.rand(D, generator=g) * 2 - 1) * math.sqrt(6.0 / (1 + D)))
        I = cfg.intermediate_size
        for l in range(self.n_layers):
            self.p[f"layer{l}.wqkv"].copy_(torch.cat([lin((D, D)) for _ in range(3)], 0))
            self.p[f"layer{l}.bqkv"].copy_(torch.cat([bias(D, D) for _ in range(3)], 0))
            self.p[f"layer{l}.wo"].copy_(lin((D, D)))
            self.p[f"layer{l}.bo"].copy_(bias(D, D))
            self.p[f"layer{l}.ln1_g"].fill_(1.0)
            self.p[f"layer{l}.ln1_b"].zero_()
            if cfg.use_intermediate:
                self.p[f"layer{l}.w1"].copy_(lin((I, D))); self.p[f"layer{l}.b1"].copy_(bias(I, D))
                self.p[f"layer{l}.w2"].copy_(lin((D, I))); self.p[f"layer{l}.b2"].copy_(bias(D, I))
                self.p[f"layer{l}.ln2_g"].fill_(1.0); self.p[f"layer{l}.ln2_b"].zero_()

    def _build_descs(self):
        cfg = self.cfg
        L, ffn = self.n_layers, bool(cfg.use_intermediate)
        self._layers = (_lib.FusionLayer * L)()
        self._grads = (_lib.FusionLayerGrad * L)()
        names = ["wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b"] + (["w1", "b1", "w2", "b2", "ln2_g", "ln2_b"] if ffn else [])
        for l in range(L):
            for nm in names:
                setattr(self._layers[l], nm, ptr(self.p[f"layer{l}.{nm}"]))
                setattr(self._grads[l], nm, ptr(self.g[f"layer{l}.{nm}"]))
        d = _lib.FusionDesc()
        d.N, d.M, d.D, d.H, d.I, d.L = self.shard.n_loc, cfg.n_tokens, cfg.dim, cfg.num_heads, cfg.intermediate_size, L
        d.use_intermediate = int(ffn)
        d.drop_p = cfg.dropout
        d.layers = ctypes.cast(self._layers, ctypes.POINTER(_lib.FusionLayer))
        d.weight_raw = ptr(self.p["weight_raw"])
        d.x0 = ptr(self.x0)
        d.hidden = ptr(self.hidden)          # joint / joint_fz / weight_norm are not needed by the hd_* read-outs
        if cfg.joint_way == "Mformer_weight":
            d.joint = ptr(self.joint)
        elif cfg.joint_way == "learnable_weight":
            d.joint_fz = ptr(self.joint)
        d.readout_slot_p1 = self.g_slot + 1 if self.ro else 0
        d.fold_others = int(self.fold)
        d.save, d.save_floats = ptr(self.save), self.save.numel()
        d.work, d.work_floats = ptr(self.work), self.work.numel()
        if self.fold:
            q_off, k_off = ctypes.c_int64(), ctypes.c_int64()
            call("snag_fusion_readout_offsets", ctypes.byref(d), ctypes.byref(q_off), ctypes.byref(k_off))
            self._qkv_ptr = self.save.data_ptr() + 4 * q_off.value            # [rows * M, 3 D] q/k/v activation of the read-out layer
            self._dkv_ptr = self.work.data_ptr() + 4 * k_off.value            # [rows * M, 2 D] gradient of its keys / values
        self._fdesc = d
        b = _lib.FusionBwdDesc()
        b.d_hidden = ptr(self.d_repr) if self.ro else ptr(self.d_hidden)      # None for the concatenation read-outs
        b.zero_d_hidden = 1 if (self.ro and cfg.scorer != "segmented") else 0      # the segmented scorer overwrites its target: nothing to clear
        if cfg.joint_way == "Mformer_weight":
            b.d_joint = ptr(self.d_repr)
        elif cfg.joint_way == "learnable_weight":
            b.d_joint_fz = ptr(self.d_repr)
        b.d_x0 = ptr(self.d_x0)
        b.grads = ctypes.cast(self._grads, ctypes.POINTER(_lib.FusionLayerGrad))
        # the hd_* read-outs never touch joint_fz, so weight_raw has no gradient: its slice of the (zero-initialised) flat gradient
        # buffer simply stays zero -- passing no pointer saves the library a 24-byte memset launch per step
        b.d_weight_raw = ptr(self.g["weight_raw"]) if cfg.joint_way == "learnable_weight" else None
        b.accumulate = 0
        self._bdesc = b
        # The scorer's per-destination pass carries the gradient through the read-out rows' LayerNorm itself (snag_score_margin_fwd_bwd_seg_ln:
        # the row never travels through memory, the LayerNorm backward launch disappears) when the scorer's table IS that LayerNorm's output:
        # read-out-only hd_graph without FFN, one rank, at most 16,384 rows (the partial slab holds 1,024 rows of 16)
        self._ln_fuse = (self.ro and not ffn and not self.shard.sharded and cfg.scorer == "segmented" and cfg.loss == "hinge" and self.sdim % 4 == 0
                         and -(-self.shard.n_loc // 16) <= 1024)
        if self._ln_fuse:
            offs = [ctypes.c_int64() for _ in range(5)]
            call("snag_fusion_readout_ln_offsets", ctypes.byref(d), *[ctypes.byref(o) for o in offs])
            t_off, st_off, dt_off, dtm_off, part_off = (o.value for o in offs)
            tl = _lib.LnBwdTail()
            tl.t, tl.stats = self.save.data_ptr() + 4 * t_off, self.save.data_ptr() + 4 * st_off
            tl.gamma = ptr(self.p[f"layer{L - 1}.ln1_g"])
            tl.dt, tl.part = self.work.data_ptr() + 4 * dt_off, self.work.data_ptr() + 4 * part_off
            self._ln_tail, self._ln_dtm = tl, self.work.data_ptr() + 4 * dtm_off
        self._set_dropout(self.training)

    def _set_dropout(self, on):
        for l in range(self.n_layers):
            self._layers[l].keep_probs = ptr(self.keep_probs[l]) if on else None
            self._layers[l].keep_attn_out = ptr(self.keep_attn[l]) if on else None
            self._layers[l].keep_ffn_out = ptr(self.keep_ffn[l]) if (on and self.keep_ffn) else None
        self._bdesc.f = self._fdesc
        self._bdesc.f.n_co_fwd = 0           # (the forward's co-launched products are the forward call's alone)
        if getattr(self, "_ln_fuse", False):
            # the read-out token's rows of the output projection's dropout mask (row stride M * D), as the LayerNorm backward takes them
            cfg, tl, l = self.cfg, self._ln_tail, self.cfg.num_layers - 1
            if on:
                tl.keep, tl.keep_ld, tl.keep_scale = self.keep_attn[l].data_ptr() + self.g_slot * cfg.dim, cfg.n_tokens * cfg.dim, 1.0 / (1.0 - cfg.dropout)
                tl.dtm = self._ln_dtm
            else:
                tl.keep, tl.keep_ld, tl.keep_scale, tl.dtm = None, 0, 1.0, None

    def train(self, mode=True):
        self.training = bool(mode)
        self._set_dropout(self.training and self.cfg.dropout > 0)
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------------------------------
    def _gemm(self, *a, **k):
        call("snag_gemm_f32", ctypes.byref(self._gemm_desc(*a, **k)), self._st)

    w_round = 256           # workgroups the weight gradients of the merged backward launch are split into (0: the library's choice; tools/ab experiments)
    l1_round = 256          # workgroups one launch of eight-wave tiles should fill (one per CU; 512 = two per CU: tools/ab experiments only --
                            # the kernel's 176 registers admit one eight-wave workgroup per CU)

    @classmethod
    def _balanced_splits(cls, shapes):
        """K splits for long-K problems [(M, K)] that share ONE launch of eight-wave 256-row tiles (snag_gemm_f32_group): every
        workgroup should walk about the same number of k-tiles and all row tiles x splits should fit one round of 256 workgroups."""
        tiles = [(m + 255) // 256 for m, _ in shapes]
        nk = [(k + 15) // 16 for _, k in shapes]
        per_wg = max(1.0, sum(t * k for t, k in zip(tiles, nk)) / float(cls.l1_round))
        cap = [max(1, k // 64) for _, k in shapes]
        s = [max(1, min(c, int(k / per_wg))) for k, c in zip(nk, cap)]
        while True:      # left-over workgroups go to the problem with the longest k-range per workgroup
            i = max(range(len(s)), key=lambda j: nk[j] / s[j])
            if s[i] >= cap[i] or sum(t * x for t, x in zip(tiles, s)) + tiles[i] > cls.l1_round:
                return s
            s[i] += 1

    def _long_k_group(self, descs, shapes):
        """The pointer array of ``_grouped_long_k``'s one launch (K splits balanced, reduces deferred), or None when the shapes do not qualify."""
        if len(descs) != 2 or self.cfg.dim > 128 or any(k < 512 for _, k in shapes) or max(m for m, _ in shapes) < 1024:
            return None
        order = sorted(range(2), key=lambda j: -shapes[j][0])          # the problem that picks the eight-wave tile on its own goes first
        splits = self._balanced_splits([shapes[j] for j in order])
        group = (ctypes.POINTER(_lib.GemmDesc) * 2)()
        for pos, j in enumerate(order):
            descs[j].defer_reduce, descs[j].splits = 1, splits[pos]
            group[pos] = ctypes.pointer(descs[j])
        return group

    def _grouped_long_k(self, descs, shapes, tags=None, small=None, prepare_fusion=False, launched=None):
        """Two long-K split-K GEMMs (the modalities' first projections, or their weight gradients) as ONE launch of eight-wave tiles with
        balanced K splits + ONE reduce; False when the shapes do not qualify (the caller then launches them one by one).
        ``small``: an array of independent snag_small_mm products that ride in the reduce launch (snag_gemm_reduce_group_mm).
        ``launched``: the group (``_long_k_group``) has gone out already -- inside snag_fusion_bwd, as its ``next_group`` -- only the reduce is left."""
        group = launched if launched is not None else self._long_k_group(descs, shapes)
        if group is None:
            return False
        tag = tags[0] if tags else ""
        if launched is None:
            self._probed(tag, call, "snag_gemm_f32_group", group, 2, self._st)
        if small is not None:
            self._probed((tag + "_reduce") if tag else "", call, "snag_gemm_reduce_group_mm", group, 2, small, len(small), self._st)
        elif prepare_fusion:
            # the reduce launch also prepares the weight images of the one-launch read-out forward (parameter-sized, independent of the reduce)
            ready = ctypes.c_int32(0)
            self._probed((tag + "_reduce") if tag else "", call, "snag_fusion_fwd_prepare_reduce", ctypes.byref(self._fdesc), group, 2, ctypes.byref(ready), self._st)
            self._fdesc.weight_images_ready = ready.value
            self._bdesc.wo_images_ready = ready.value
        else:
            self._probed((tag + "_reduce") if tag else "", call, "snag_gemm_reduce_pair", ctypes.byref(descs[0]), ctypes.byref(descs[1]), self._st)
        return True

    def _fold_kv_descs(self):
        """Folded second projections, forward: K|V_m = mid_m Wf_m^T + b_kv straight into the layer's q/k/v activation."""
        cfg = self.cfg
        D, MD, Nl = cfg.dim, cfg.n_tokens * cfg.dim, self.shard.n_loc
        return [self._gemm_desc(Nl, 2 * D, D, ptr(self.mid[i]), D, 1, ptr(self.wf[i]), D, 1, self._qkv_ptr + 4 * (i * 3 * D + D), 3 * MD,
                                bias=self.p["layer0.bqkv"].data_ptr() + 4 * D, splitk=False) for i in range(len(cfg.feat_dims))]

    def _fold_bwd_descs(self, defer, merged=False):
        """Folded second projections, backward: (d Wf_m = dKV_m^T mid_m, K split over the rows, slabs in work_l2), (d mid_m = (dKV_m Wf_m) *
        [mid_m > 0] (* keep)) for every modality; dKV_m sits in the Mformer backward's scratch (rows n * M + m of its [rows * M, 2 D] block)."""
        cfg = self.cfg
        D, M, Nl = cfg.dim, cfg.n_tokens, self.shard.n_loc
        MD, dkv = M * D, self._dkv_ptr
        nmod = len(cfg.feat_dims)
        dl2 = [self._gemm_desc(2 * D, D, Nl, dkv + 4 * i * 2 * D, 2 * MD, 0, ptr(self.mid[i]), D, 0, ptr(self.d_wf[i]), D, splitk=True, half=i % 2,
                               scratch=self.work_l2) for i in range(nmod)]
        # inside the merged backward launch (snag_fusion_bwd, fold_others) the weight gradients share ONE round of ~256 workgroups: the layer's own
        # dWqkv (3D / 64 tiles) and dWo (D / 64) + these (2D / 64 each); the library applies the same rule to its two
        # (``merged``: three tokens -- what snag_fusion_bwd runs as that one launch)
        want = max(1, self.w_round // max(1, (4 * D + nmod * 2 * D) // 64)) if (self.w_round and merged and M == 3) else 0
        for d in dl2:
            d.defer_reduce, d.splits = int(defer), want
        dmid = [self._gemm_desc(Nl, D, 2 * D, dkv + 4 * i * 2 * D, 2 * MD, 1, ptr(self.wf[i]), D, 0, ptr(self.d_mid[i]), D, act=4, aux=ptr(self.mid[i]),
                                ldaux=D, keep=self._proj_keep(i), keep_scale=1.0 / (1.0 - cfg.proj_dropout), splitk=False) for i in range(nmod)]
        return dl2, dmid

    def _dl1_descs(self, noisy):
        """The first projections' weight gradients feat_m^T d(projection input) [fd, D], stored transposed straight into the [D, fd] gradients."""
        cfg = self.cfg
        D, MD, Nl = cfg.dim, cfg.n_tokens * cfg.dim, self.shard.n_loc
        dx0 = self.d_x0.data_ptr()
        dl1 = []
        for i, fd in enumerate(cfg.feat_dims):
            feat = self.noisy[i] if noisy else self.feats[i]
            dslot = dx0 + 4 * i * D
            if cfg.num_proj == 1:
                # dW^T [fd, D] = feat^T dTok, stored transposed: straight into the [D, fd] gradient ; db = colsum(dTok)
                dl1.append(self._gemm_desc(fd, D, Nl, ptr(feat), fd, 0, dslot, MD, 0, ptr(self.g[f"proj{i}.weight"]), fd, splitk=True, half=i % 2))
            else:
                dl1.append(self._gemm_desc(fd, D, Nl, ptr(feat), fd, 0, ptr(self.d_mid[i]), D, 0, ptr(self.g[f"proj{i}.l1"]), fd, splitk=True,
                                           half=i % 2))   # dl1^T [fd, D], stored transposed
        for d in dl1:
            d.c_transposed = 1
        return dl1

    def _fold_chain_products(self, i0=0):
        """snag_small_mm array for modalities i0, i0 + 1: d l2_m = Wkv^T d Wf_m (disjoint outputs) and ONE two-term problem d Wkv += sum_m d Wf_m l2_m^T."""
        cfg = self.cfg
        D, nmod = cfg.dim, len(cfg.feat_dims)
        wkv = self.p["layer0.wqkv"].data_ptr() + 4 * D * D
        gkv = self.g["layer0.wqkv"].data_ptr() + 4 * D * D
        n_here = min(2, nmod - i0)
        probs = (_lib.SmallMM * (n_here + 1))()
        for j in range(n_here):
            i = i0 + j
            a = probs[j]
            a.M, a.N, a.K, a.accumulate = D, D, 2 * D, 0
            a.A, a.sam, a.sak = wkv, 1, D
            a.B, a.sbk, a.sbn = ptr(self.d_wf[i]), D, 1
            a.C, a.ldc = ptr(self.g[f"proj{i}.l2"]), D
        b2 = probs[n_here]
        b2.M, b2.N, b2.K, b2.accumulate = 2 * D, D, D, 1
        b2.A, b2.sam, b2.sak = ptr(self.d_wf[i0]), D, 1
        b2.B, b2.sbk, b2.sbn = ptr(self.p[f"proj{i0}.l2"]), 1, D
        if n_here == 2:
            b2.A2, b2.B2 = ptr(self.d_wf[i0 + 1]), ptr(self.p[f"proj{i0 + 1}.l2"])
        b2.C, b2.ldc = gkv, D
        return probs

    def _gemm_pair(self, d0, d1):
        """Two same-shaped GEMMs (the two modalities' second projections / their backward) as ONE launch: each is a grid of fewer
        workgroups than the chip has CUs.  Split-K workspaces: the two halves of the engine's scratch."""
        call("snag_gemm_f32_pair", ctypes.byref(d0), ctypes.byref(d1), self._st)

    def _gemm_deferred_pairs(self, descs, tags=None):
        """Independent split-K GEMMs of DIFFERENT shape, two at a time (descs[i] uses workspace half i % 2): both kernels run with
        their K-split slabs left in place, then ONE launch reduces both (snag_gemm_reduce_pair) -- one reduce launch per pair
        instead of one per problem.  ``tags``: per-problem probe names for bench.py's launch timers (the pair's reduce is timed as
        tags[i] + "_reduce")."""
        st = self._st
        for i in range(0, len(descs), 2):
            pair = descs[i:i + 2]
            for j, d in enumerate(pair):
                d.defer_reduce = 1 if len(pair) == 2 else 0
                tag = tags[i + j] if tags else ""
                self._probed(tag, call, "snag_gemm_f32", ctypes.byref(d), st)
            if len(pair) == 2:
                self._probed((tags[i] + "_reduce") if tags and tags[i] else "", call, "snag_gemm_reduce_pair", ctypes.byref(pair[0]), ctypes.byref(pair[1]), st)

    def _gemm_desc(self, M, N, K, A, lda, akc, B, ldb, bkc, C, ldc, *, bias=None, act=0, pre=None, add1=None, ld1=0,
                   aux=None, ldaux=0, splitk=True, keep=None, keep_scale=1.0, half=None, scratch=None):
        d = _lib.GemmDesc()
        d.M, d.N, d.K = M, N, K
        d.a_kcontig, d.b_kcontig = int(akc), int(bkc)
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc = A, lda, B, ldb, C, ldc
        d.bias, d.alpha, d.act, d.pre_act = bias, 1.0, act, pre
        d.add1, d.ld1 = add1, ld1
        d.aux, d.ldaux = aux, ldaux
        if keep is not None:
            d.keep_mask, d.keep_scale = keep, keep_scale
        if splitk:
            work = self.work if scratch is None else scratch
            if half is None:
                d.workspace, d.workspace_floats = ptr(work), work.numel()
            else:                         # member `half` (0 / 1) of a pair: its own half of the scratch
                n = work.numel() // 2 // 64 * 64
                d.workspace, d.workspace_floats = work.data_ptr() + 4 * n * half, n
        return d

    def _proj_keep(self, i):
        """Device pointer of modality i's ProjectionHead keep mask (None in eval mode / with proj_dropout = 0)."""
        if self.keep_proj is None or not self.training:
            return None
        return self.keep_proj[i].data_ptr()

    def _mark(self, name):
        """Phase boundary on the launch stream (bench.py's phases_us: fusion forward, all-gather wait, scorer, reduce-scatter wait,
        backward, exposed all-reduce wait, clip + AdamW).  The compute stream waits for every collective, so an event behind one
        includes the time the stream sat waiting for it."""
        if self.marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))

    def _probed(self, tag, fn, *a, **k):
        tm = self.probe.get(tag) if self.probe else None
        return tm.wrap(fn)(*a, **k) if tm is not None else fn(*a, **k)

    @staticmethod
    def _regions(rows):
        """(seed, offset, n, p, tensor) rows -> a host array of snag_mask_region."""
        arr = (_lib.MaskRegion * len(rows))()
        for a, row in zip(arr, rows):
            seed, off, n, p, out = row[:5]
            a.seed, a.offset, a.n, a.p, a.invert, a.out = int(seed) & (2 ** 64 - 1), int(off), int(n), float(p), 0, out.data_ptr()
            if len(row) > 5:         # (period, skip, take): only those bytes of every period are drawn
                a.period, a.skip, a.take = (int(v) for v in row[5])
        return arr

    def update_noise(self, seed: Optional[int] = None):
        """Once per epoch (README "noise-level: epoch", main.py:253-254 -> SNAG.update_noise model/SNAG.py:86-98) -- or per step
        ("noise-level: step").  Launches: ONE for the three row masks (rand < ratio, drawn inverted: no xor pass), one refresh per
        modality that rewrites only the rows in the new or the previous mask, the entity table's column statistics and its noise
        rows.  Every stream is keyed on global rows: all ranks draw one experiment."""
        cfg = self.cfg
        st = stream()
        from . import ops
        if self.feat_stats is None:            # clean-table statistics are constants (get_mean_std, :77-84)
            self.feat_stats = [ops.colstats(f) for f in self.feats]
            self._stat_work = torch.empty(self.lib.snag_colstats_work_floats(cfg.dim), device=self.dev)
            self._ent_stats = (torch.empty(cfg.dim, device=self.dev), torch.empty(cfg.dim, device=self.dev))
        elif getattr(self, "_stat_work", None) is None:
            self._stat_work = torch.empty(self.lib.snag_colstats_work_floats(cfg.dim), device=self.dev)
            self._ent_stats = (torch.empty(cfg.dim, device=self.dev), torch.empty(cfg.dim, device=self.dev))
        self.noise_epoch += 1
        base = (cfg.seed if seed is None else seed) * 1000003 + self.noise_epoch * 7919
        N, D, Nl, lo = cfg.n_ent, cfg.dim, self.shard.n_loc, self.shard.lo
        gen = self.noise_epoch & 1
        rows = [(base + 11 * i + 1, lo, Nl, float(cfg.noise_ratio), self.row_mask[i][gen]) for i in range(len(self.feats))]
        rows.append((base + 102, 0, N, float(cfg.noise_ratio) * 0.5, self.ent_noise_mask))
        regs = self._regions(rows)
        for a in regs:
            a.invert = 1                       # row mask = rand < ratio (model/SNAG.py:68, :97)
        call("snag_bernoulli_regions", regs, len(regs), None, st)
        for i, f in enumerate(self.feats):
            mean, std = self.feat_stats[i]
            call("snag_noise_refresh", ptr(f), Nl, f.shape[1], ptr(mean), ptr(std), ptr(self.row_mask[i][gen]),
                 ptr(self.row_mask[i][gen ^ 1]) if self.noise_active else None, base + 11 * i + 2, lo, float(cfg.mask_ratio),
                 ptr(self.noisy[i]), None, 0, st)
        mean, std = self._ent_stats
        call("snag_colstats", ptr(self.p["ent"]), N, D, None, N, ptr(mean), ptr(std), ptr(self._stat_work), st)
        call("snag_gaussian_fill", base + 101, N, D, ptr(mean), ptr(std), ptr(self.ent_noise), st)
        self.noise_active = True

    # ------------------------------------------------------------------------------------------
    def _use_seg(self):
        """The atomic-free scorer needs a compact [rows, width] gradient target it may overwrite whole, and 16-byte rows."""
        return self.cfg.scorer == "segmented" and self.d_repr is not None and self.sdim % 4 == 0

    def _seg_workspace(self, B, K):
        key = (B, K)
        if getattr(self, "_seg_key", None) != key:
            n = self.lib.snag_score_seg_work_bytes(self.cfg.n_ent, self.cfg.n_rel, B, K, self.sdim)
            self._seg_work = torch.zeros((n + 255) // 256 * 256, device=self.dev, dtype=torch.uint8)     # zeroed once; launches keep it ready
            self._seg_key = key
        return self._seg_work

    def _zero_score_grads(self):
        """Zero the scorer's scatter targets; returns (pointer, row stride) of the [N, dim] read-out gradient."""
        cfg = self.cfg
        self.g["rel"].zero_()
        if self.shard.sharded:               # the scorer scatters over ALL entity rows; the owners get their sums by reduce-scatter
            self.d_repr_full.zero_()
            return self.d_repr_full.data_ptr(), self.sdim
        if self.d_repr is not None:
            if self.concat or not self._d_repr_clean:       # (hd_*: only after a forward(with_backward=True) that no backward() followed)
                self.d_repr.zero_()
            self._d_repr_clean = False
            return self.d_repr.data_ptr(), self.sdim
        self.d_hidden.zero_()
        return self.d_hidden.data_ptr() + 4 * self.g_slot * cfg.dim, cfg.n_tokens * cfg.dim

    def forward(self, h, r, t, neg_t, want_scores=False, with_backward=False, grad_scale=1.0, sample=None):
        """Projections -> tokens -> fusion -> read-out -> scores/loss.  Returns the device scalar loss.

        ``with_backward``: a ``backward(grad_scale)`` follows (training).  The hinge scorer then runs forward and backward as ONE
        launch here (snag_score_margin_fwd_bwd) and ``backward`` starts at the read-out gradient.

        ``sample=(seed, known_sorted)``: ``neg_t`` is an OUTPUT -- the step's negative tails are drawn into it by the same launch
        that writes the structural token and the dropout masks (snag_step_prologue; the draws are those of
        ``ops.sample_negatives(h, r, t, ..., seed, known_sorted, tick=self.tick)``)."""
        cfg = self.cfg
        N, D, M = cfg.n_ent, cfg.dim, cfg.n_tokens
        Nl, lo = self.shard.n_loc, self.shard.lo               # this rank's entity rows (all of them with one rank)
        st = self._st = stream()
        self._bdesc.ln_bwd_done = 0
        noisy = self.training and cfg.add_noise and self.noise_active
        if self.keep_proj is not None and self.training:
            call("snag_bernoulli_regions", self._proj_regions, len(self._proj_regions), ptr(self.tick), st)
        MD = M * D
        x0p = self.x0.data_ptr()
        # one launch: (negative tails,) structural token, Bert dropout masks -- three independent, under-filled grids side by side
        drop = self.training and cfg.dropout > 0
        B = int(h.shape[0])
        if sample is not None and B * cfg.neg_num > 0:
            s_seed, known = sample
            assert neg_t.dtype == torch.int32 and neg_t.is_contiguous() and tuple(neg_t.shape) == (B, cfg.neg_num)
            s_args = (int(s_seed), ptr(self.tick), ptr(h), ptr(r), ptr(t), B, cfg.neg_num, N, cfg.n_rel, ptr(known), 0 if known is None else known.numel(), ptr(neg_t))
        else:
            s_args = (0, ptr(self.tick), None, None, None, 0, 0, N, cfg.n_rel, None, 0, None)
        # folded second projections: Wf_m = Wkv l2_m (parameter-sized, depends on nothing but the parameters) rides in the same launch
        # (up to four modalities; more: the products get a launch of their own further down)
        nmod = len(cfg.feat_dims)
        wf_probs, n_wf = None, 0
        if self.fold and nmod <= 4:
            wkv = self.p["layer0.wqkv"].data_ptr() + 4 * D * D
            wf_probs, n_wf = (_lib.SmallMM * nmod)(), nmod
            for i, pm in enumerate(wf_probs):
                pm.M, pm.N, pm.K, pm.accumulate = 2 * D, D, D, 0
                pm.A, pm.sam, pm.sak = wkv, D, 1
                pm.B, pm.sbk, pm.sbn = ptr(self.p[f"proj{i}.l2"]), D, 1
                pm.C, pm.ldc = ptr(self.wf[i]), D
        call("snag_step_prologue", *s_args, self.p["ent"].data_ptr() + 4 * lo * D, Nl, D, (self.ent_noise.data_ptr() + 4 * lo * D) if noisy else None,
             (self.ent_noise_mask.data_ptr() + lo) if noisy else None, float(cfg.mask_ratio) * 0.5, x0p + 4 * self.g_slot * D, MD,
             self._drop_regions if drop else None, len(self._drop_regions) if drop else 0, wf_probs, n_wf, st)
        l1 = []
        for i, fd in enumerate(cfg.feat_dims):
            feat = self.noisy[i] if noisy else self.feats[i]                 # [Nl, fd] row-major: a K-contiguous operand
            fa, fld, fkc = ptr(feat), fd, 1
            slot = x0p + 4 * i * D
            if cfg.num_proj == 1:
                l1.append(self._gemm_desc(Nl, D, fd, fa, fld, fkc, ptr(self.p[f"proj{i}.weight"]), fd, 1, slot, MD,
                                          bias=ptr(self.p[f"proj{i}.bias"]), half=i % 2))
            else:   # ProjectionHead: Linear(no bias) -> ReLU -> dropout -> Linear(no bias)   (model/layers.py:135-148)
                wb, wld, wkc = ptr(self.p[f"proj{i}.l1"]), fd, 1
                l1.append(self._gemm_desc(Nl, D, fd, fa, fld, fkc, wb, wld, wkc, ptr(self.mid[i]), D, act=1,
                                          keep=self._proj_keep(i), keep_scale=1.0 / (1.0 - cfg.proj_dropout), half=i % 2))
        # the modalities' first projections: K-split GEMMs of different K.  Two modalities: ONE launch of eight-wave tiles with the K
        # splits balanced across both (visual 3 + text 1 at the DB15K shape: 236 workgroups walking 86 / 48 k-tiles instead of a 236-
        # and a 177-workgroup launch) + one reduce; otherwise two kernels and ONE reduce launch per pair
        self._fdesc.weight_images_ready = 0
        self._bdesc.wo_images_ready = 0
        self._fdesc.force_one_launch = self._bdesc.f.force_one_launch = int(self.force_one_launch)
        co = self.fold and nmod <= 2 and self.co_launch
        if co:      # the folded key / value products ride in snag_fusion_fwd: described now, so that the first projections' reduce launch can prepare its weights
            kv = self._fold_kv_descs()
            self._co_fwd = kv
            for j, d in enumerate(kv):
                self._fdesc.co_fwd[j] = ctypes.pointer(d)
            self._fdesc.n_co_fwd = nmod
        self.l1_grouped = self._grouped_long_k(l1, [(Nl, fd) for fd in cfg.feat_dims], ["proj_l1_fwd"], prepare_fusion=co and n_wf > 0)
        if not self.l1_grouped:
            self._gemm_deferred_pairs(l1, ["proj_l1_fwd"] + [""] * (len(l1) - 1))
        if self.fold:
            # folded second projections: Wf_m = Wkv l2_m (parameter-sized: one small launch), then the projected tokens' keys / values
            # straight from mid_m into the layer's q/k/v activation; the tokens themselves are never formed
            wkv = self.p["layer0.wqkv"].data_ptr() + 4 * D * D
            for i0 in range(0, nmod if n_wf == 0 else 0, 4):
                probs = (_lib.SmallMM * min(4, nmod - i0))()
                for j, pm in enumerate(probs):
                    i = i0 + j
                    pm.M, pm.N, pm.K, pm.accumulate = 2 * D, D, D, 0
                    pm.A, pm.sam, pm.sak = wkv, D, 1
                    pm.B, pm.sbk, pm.sbn = ptr(self.p[f"proj{i}.l2"]), D, 1
                    pm.C, pm.ldc = ptr(self.wf[i]), D
                call("snag_small_matmul", probs, len(probs), st)
            # (co: launched by snag_fusion_fwd -- ONE launch for the whole read-out forward at hidden_size 128 (csrc/readout_fused.hip), otherwise
            # one grouped launch with the read-out token's q/k/v projection)
            if not co:
                kv = self._fold_kv_descs()
                self._fdesc.n_co_fwd = 0
                for i in range(0, nmod - 1, 2):
                    self._gemm_pair(kv[i], kv[i + 1])
                if nmod % 2:
                    call("snag_gemm_f32", ctypes.byref(kv[-1]), st)
        elif cfg.num_proj == 2:       # the second projections, two modalities per launch
            l2 = [self._gemm_desc(Nl, D, D, ptr(self.mid[i]), D, 1, ptr(self.p[f"proj{i}.l2"]), D, 1, x0p + 4 * i * D, MD) for i in range(len(cfg.feat_dims))]
            for i in range(0, len(l2) - 1, 2):
                self._gemm_pair(l2[i], l2[i + 1])
            if len(l2) % 2:
                call("snag_gemm_f32", ctypes.byref(l2[-1]), st)
        if self.atten:
            call("snag_atten_weight_fwd", x0p, Nl, M, D, ptr(self.p["atten.a"]), ptr(self.hidden), D, ptr(self.aw), st)
        else:
            call("snag_fusion_fwd", ctypes.byref(self._fdesc), st)
        if cfg.joint_way == "Mformer_hd_mean":
            call("snag_token_mean", ptr(self.hidden), Nl, M, D, ptr(self.repr_mean), st)
            self._repr, self._repr_ld = self.repr_mean.data_ptr(), D
        elif self.ro or self.atten:
            self._repr, self._repr_ld = self.hidden.data_ptr(), D
        elif self.concat:
            self._repr, self._repr_ld = self.joint.data_ptr(), MD
        else:
            self._repr, self._repr_ld = self.hidden.data_ptr() + 4 * self.g_slot * D, MD
        D = self.sdim                                          # from here on: the width the scorer works at
        self._mark("fusion_fwd")
        if self.shard.sharded:
            # every rank's compact read-outs -> the one [N, D] table all ranks score against (the only data-path collective of the
            # forward; stream-ordered: the compute stream waits for it, the host does not)
            self.shard.all_gather(self.repr_full, self.repr_mean if cfg.joint_way == "Mformer_hd_mean" else (self.joint if self.concat else self.hidden))
            self._repr, self._repr_ld = self.repr_full.data_ptr(), D
            self._mark("all_gather")
        B, K = neg_t.shape
        if self._loss_part is None or self._loss_part.shape[0] != B or self._neg.shape[1] != K:       # keyed on (B, K): eval scores one dummy negative
            self._loss_part = torch.empty(B, device=self.dev)
            self._pos = torch.empty(B, device=self.dev)
            self._neg = torch.empty(B, K, device=self.dev)
        ib = 4 if h.dtype == torch.int32 else 8
        self._batch = (h, r, t, neg_t, ib, B, K)
        self._score_bwd_scale = None
        if with_backward and cfg.loss == "hinge" and K > 0 and self._use_seg():
            # training, default: scores, losses and ALL gradient rows of the read-out and the relation table without float atomics
            # (no zero-fill of the targets: every row is written)
            if self.shard.sharded:
                tgt, tld = self.d_repr_full.data_ptr(), self.sdim
            else:
                tgt, tld = self.d_repr.data_ptr(), self.sdim
            work = self._seg_workspace(B, K)
            if self._ln_fuse:
                self._probed("score_fwd_bwd", call, "snag_score_margin_fwd_bwd_seg_ln", self._repr, N, ptr(self.p["rel"]), cfg.n_rel, D,
                             ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(cfg.margin), float(grad_scale), ptr(self._pos), ptr(self._neg),
                             ptr(self._loss_part), ptr(self.loss), ptr(self.g["rel"]), ptr(work), work.numel(), ctypes.byref(self._ln_tail), st)
            else:
                self._probed("score_fwd_bwd", call, "snag_score_margin_fwd_bwd_seg", self._repr, self._repr_ld, N, ptr(self.p["rel"]), cfg.n_rel, D,
                             ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(cfg.margin), float(grad_scale), ptr(self._pos), ptr(self._neg),
                             ptr(self._loss_part), ptr(self.loss), ptr(self._ticket), tgt, tld, ptr(self.g["rel"]), 1, ptr(work), work.numel(), st)
            self._bdesc.ln_bwd_done = int(self._ln_fuse)
            self._score_bwd_scale = float(grad_scale)
            self._d_repr_clean = False
            self._mark("score")
            if want_scores:
                return self.loss, self._pos, self._neg
            return self.loss
        if with_backward and cfg.loss == "hinge" and not cfg.deterministic and K > 0:
            # training: the scorer's forward AND backward in one launch (every row gathered once, summed loss included)
            d_repr, d_ld = self._zero_score_grads()
            self._probed("score_fwd_bwd", call, "snag_score_margin_fwd_bwd", self._repr, self._repr_ld, N, ptr(self.p["rel"]), cfg.n_rel, D,
                         ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K, float(cfg.margin), float(grad_scale), ptr(self._pos), ptr(self._neg),
                         ptr(self._loss_part), ptr(self.loss), ptr(self._ticket), d_repr, d_ld, ptr(self.g["rel"]), st)
            self._score_bwd_scale = float(grad_scale)
            self._mark("score")
            if want_scores:
                return self.loss, self._pos, self._neg
            return self.loss
        self._probed("score_fwd", call, "snag_score_margin_fwd", self._repr, self._repr_ld, N, ptr(self.p["rel"]), cfg.n_rel, D, ptr(h), ptr(r),
                     ptr(t), ptr(neg_t), ib, B, K, float(cfg.margin), ptr(self._pos), ptr(self._neg), ptr(self._loss_part), st)
        if cfg.loss == "self_adversarial":
            if getattr(self, "_dpos", None) is None or self._dpos.shape[0] != B or self._dneg.shape[1] != K:
                self._dpos, self._dneg = torch.empty(B, device=self.dev), torch.empty(B, K, device=self.dev)
            call("snag_selfadv_loss", ptr(self._pos), ptr(self._neg), B, K, float(cfg.margin), float(cfg.adv_temp), ptr(self._loss_part),
                 ptr(self._dpos), ptr(self._dneg), st)
        elif cfg.loss != "hinge":
            raise ValueError(f"unknown loss {cfg.loss!r}")
        call("snag_reduce_sum", ptr(self._loss_part), B, ptr(self.loss), st)
        if want_scores:
            return self.loss, self._pos, self._neg
        return self.loss

    def grad_buckets(self):
        """Contiguous slices of the flat gradient buffer in the order the backward finishes them:
        [entity + relation tables] and [fusion parameters] (both final right after the fusion backward), then one slice per
        modality's projection weights (final after that modality's backward GEMMs; the 4096-wide visual one comes first so its
        2 MB exchange hides behind the remaining modalities' GEMMs)."""
        first_proj = self.slices["proj0.weight" if self.cfg.num_proj == 1 else "proj0.l1"][0]
        fusion0 = self.slices["weight_raw"][0]
        out = [self.gflat[:first_proj], self.gflat[fusion0:]]          # (row-sharded runs split the first one: see backward)
        starts = [self.slices[f"proj{i}.weight" if self.cfg.num_proj == 1 else f"proj{i}.l1"][0] for i in range(len(self.cfg.feat_dims))] + [fusion0]
        out += [self.gflat[starts[i]:starts[i + 1]] for i in range(len(self.cfg.feat_dims))]
        return out

    def backward(self, grad_scale=1.0, reducer=None):
        """Gradient of grad_scale * loss w.r.t. every parameter, written (not added) into the flat gradient buffer.

        ``reducer`` (snag_amd.parallel.GradAllReduce) gets ``start(bucket)`` as soon as a bucket is final (see grad_buckets), so
        that the exchange of the 7.7 MB entity-table gradient overlaps the projection backward GEMMs and only the last, smallest
        modality's weights are exchanged after the compute has finished; ``finish()`` is the caller's."""
        cfg = self.cfg
        N, D, M = cfg.n_ent, cfg.dim, cfg.n_tokens
        Nl, lo = self.shard.n_loc, self.shard.lo
        st = self._st
        MD = M * D
        h, r, t, neg_t, ib, B, K = self._batch
        noisy = self.training and cfg.add_noise and self.noise_active
        fused_done = self._score_bwd_scale is not None
        if fused_done and self._score_bwd_scale != float(grad_scale):
            raise ValueError(f"forward(with_backward=True, grad_scale={self._score_bwd_scale}) was followed by backward(grad_scale={grad_scale})")
        self._score_bwd_scale = None
        if not fused_done:
            d_repr, d_ld = self._zero_score_grads()
        up = (ptr(self._dpos), ptr(self._dneg)) if cfg.loss == "self_adversarial" else (None, None)
        if fused_done:
            pass                                       # the scorer's gradients were scattered by the fused launch in forward()
        elif cfg.deterministic:
            self._acc_ent.zero_()
            self._acc_rel.zero_()
            Ds = self.sdim
            call("snag_score_margin_bwd_fixed", self._repr, self._repr_ld, N, ptr(self.p["rel"]), cfg.n_rel, Ds, ptr(h), ptr(r), ptr(t), ptr(neg_t),
                 ib, B, K, float(cfg.margin), float(grad_scale), *up, ptr(self._acc_ent), ptr(self._acc_rel), st)
            call("snag_fixed_to_float", ptr(self._acc_ent), N, Ds, d_repr, d_ld, 0, st)
            call("snag_fixed_to_float", ptr(self._acc_rel), cfg.n_rel, Ds, ptr(self.g["rel"]), Ds, 0, st)
        else:
            call("snag_score_margin_bwd", self._repr, self._repr_ld, N, ptr(self.p["rel"]), cfg.n_rel, self.sdim, ptr(h), ptr(r), ptr(t), ptr(neg_t),
                 ib, B, K, float(cfg.margin), float(grad_scale), *up, d_repr, d_ld, ptr(self.g["rel"]), st)
        if self.shard.sharded:
            # every rank scattered its triples' gradients over ALL entity rows: the owners get the sums of their rows
            self.shard.reduce_scatter(self.d_repr, self.d_repr_full)
            self._mark("reduce_scatter")
        if cfg.joint_way == "Mformer_hd_mean":
            call("snag_token_mean_bwd", ptr(self.d_repr), Nl, M, D, ptr(self.d_hidden), 0 if cfg.scorer == "segmented" else 1, st)
        # the entity table's gradient (this rank's rows only; the others stay zero from construction) = the structural token's rows of
        # d_x0 through the noise blend: in read-out mode with one layer the Mformer backward's last GEMM writes it as a second copy of
        # its output, otherwise a launch of its own derives it from d_x0
        mirror = self.ro and cfg.num_layers == 1
        bd = self._bdesc
        bd.d_slot_out = (self.g["ent"].data_ptr() + 4 * lo * D) if mirror else None
        bd.d_slot_ld = D
        bd.d_slot_scale = float(np.float32(1.0) - np.float32(float(cfg.mask_ratio) * 0.5))       # 1.0f - a in float, as snag_entity_token_bwd forms it
        bd.d_slot_rowmask = (self.ent_noise_mask.data_ptr() + lo) if (mirror and noisy) else None
        nmod = len(cfg.feat_dims)
        co = self.fold and nmod <= 2
        if co:
            # folded second projections (see below): with one or two modalities their products on the key / value gradient ride in the
            # Mformer backward's own grouped launches (co_dx with d x of the read-out token, co_dw with d Wo / d Wqkv and ONE reduce)
            dl2, dmid = self._fold_bwd_descs(True, merged=True)
            self._co_bwd = (dl2, dmid)
            for j in range(nmod):
                bd.co_dw[j], bd.co_dx[j] = ctypes.pointer(dl2[j]), ctypes.pointer(dmid[j])
            bd.n_co_dw = bd.n_co_dx = nmod
        # One rank, folded second projections riding in the Mformer backward (co): the first projections' weight gradients -- the launch this
        # method would issue next, 247 eight-wave workgroups on 256 CUs -- go out from INSIDE snag_fusion_bwd, whose last reduce (the layer's
        # weight-gradient slabs + slab column sums, a 9 us launch) rides in them on the CUs they leave idle (snag_fusion_bwd_desc.next_group).
        # With a reducer the launches stay apart: the tables' bucket leaves right behind the fusion backward and travels under that GEMM.
        dl1, rider_group = None, None
        # (likewise the full layer's layer-0 weight-gradient reduce when the projections are single Linear layers: MKG-W-shaped configurations)
        if self.tail_riders and reducer is None and not self.atten and ((co and cfg.num_proj == 2) or (cfg.num_proj == 1 and not self.ro)):
            dl1 = self._dl1_descs(noisy)
            rider_group = self._long_k_group(dl1, [(fd, Nl) for fd in cfg.feat_dims])
        bd.next_group, bd.n_next_group = (ctypes.cast(rider_group, ctypes.c_void_p), 2) if rider_group is not None else (None, 0)
        self._rider_group = rider_group           # (keeps the pointer array alive for the call)
        if self.atten:
            call("snag_atten_weight_bwd", self.x0.data_ptr(), Nl, M, D, ptr(self.p["atten.a"]), ptr(self.aw), ptr(self.d_repr), D, ptr(self.d_x0),
                 ptr(self.g["atten.a"]), 0, ptr(self._aw_work), st)
        else:
            call("snag_fusion_bwd", ctypes.byref(bd), st)
        # (atomic scorer: snag_fusion_bwd / snag_token_mean_bwd cleared d_repr behind themselves; snag_atten_weight_bwd only reads it)
        self._d_repr_clean = cfg.scorer != "segmented" and not self.atten
        dx0 = self.d_x0.data_ptr()
        wk = ptr(self.work)
        if not mirror:
            call("snag_entity_token_bwd", dx0 + 4 * self.g_slot * D, MD, Nl, D, (self.ent_noise_mask.data_ptr() + lo) if noisy else None,
                 float(cfg.mask_ratio) * 0.5, self.g["ent"].data_ptr() + 4 * lo * D, 0, st)
        buckets = self.grad_buckets() if reducer is not None else None
        if reducer is not None:
            if self.shard.sharded:       # entity rows: each rank holds its own slice -> all-gather; relations: summed
                reducer.start_gather(self.g_ent_pad, self.shard)
                reducer.start(self.g["rel"].reshape(-1))
            else:
                reducer.start(buckets[0])
            if not self.fold:            # (folded second projections: the key / value weight gradient is completed further down)
                reducer.start(buckets[1])
        # two modalities, one rank: the four projection weight gradients (dl2 pair, dl1 x 2) leave their K-split slabs in place and
        # ONE launch at the end reduces all four
        one_reduce = reducer is None and cfg.num_proj == 2 and nmod == 2 and not co
        if co:
            pass
        elif self.fold:
            # folded second projections: from the gradient of the projected tokens' keys / values (left in the Mformer backward's scratch)
            #   d mid_m = (dKV_m Wf_m) * [mid_m > 0] (* keep)        [rows, D]   <- K = 2 D
            #   d Wf_m  = dKV_m^T mid_m                               [2 D, D]   <- K = rows (split, slabs in work_l2: the scratch holds dKV)
            # and, once d Wf_m is final, the chain rule onto the two factors of Wf_m = Wkv l2_m (parameter-sized, one small launch)
            dl2, dmid = self._fold_bwd_descs(one_reduce)
            for group in (dmid, dl2):
                for i in range(0, nmod - 1, 2):
                    self._gemm_pair(group[i], group[i + 1])
                if nmod % 2:
                    call("snag_gemm_f32", ctypes.byref(group[-1]), st)
        elif cfg.num_proj == 2:
            # dl2 = dTok^T mid and d mid = (dTok l2) * [mid > 0] (* keep): same shapes for every modality -> two per launch
            dl2 = [self._gemm_desc(D, D, Nl, dx0 + 4 * i * D, MD, 0, ptr(self.mid[i]), D, 0, ptr(self.g[f"proj{i}.l2"]), D, splitk=True, half=i % 2,
                                   scratch=self.work_l2 if one_reduce else None) for i in range(nmod)]
            for d in dl2:
                d.defer_reduce = int(one_reduce)
            dmid = [self._gemm_desc(Nl, D, D, dx0 + 4 * i * D, MD, 1, ptr(self.p[f"proj{i}.l2"]), D, 0, ptr(self.d_mid[i]), D, act=4, aux=ptr(self.mid[i]),
                                    ldaux=D, keep=self._proj_keep(i), keep_scale=1.0 / (1.0 - cfg.proj_dropout), splitk=False) for i in range(nmod)]
            for group in (dl2, dmid):
                for i in range(0, nmod - 1, 2):
                    self._gemm_pair(group[i], group[i + 1])
                if nmod % 2:
                    call("snag_gemm_f32", ctypes.byref(group[-1]), st)
        # folded second projections, the chain rule onto the two factors of Wf_m = Wkv l2_m (d Wf_m is final by now):
        #   d l2_m = Wkv^T d Wf_m ;  d Wkv += d Wf_m l2_m^T   (the read-out token's own contribution to d Wkv is already there)
        chain, chain_done = (self._fold_chain_products() if (self.fold and nmod <= 2) else None), False
        grouped = False             # both first-projection weight gradients went out as the one grouped launch (+ the chain products in its reduce)
        if dl1 is None:             # (described up front only where they ride with the fusion backward: with a reducer the host reaches that call sooner)
            dl1 = self._dl1_descs(noisy)
        if one_reduce:
            group = (ctypes.POINTER(_lib.GemmDesc) * 4)()
            for j, d in enumerate(dl2 + dl1):
                d.defer_reduce = 1
                group[j] = ctypes.pointer(d)
            call("snag_gemm_f32", ctypes.byref(dl1[0]), st)
            call("snag_gemm_f32", ctypes.byref(dl1[1]), st)
            call("snag_gemm_reduce_group", group, 4, st)
        elif self._grouped_long_k(dl1, [(fd, Nl) for fd in cfg.feat_dims], small=chain, launched=rider_group):
            # (both weight gradients in ONE launch of eight-wave tiles: alone, the visual one fills the chip exactly -- 16 row tiles x 16
            # K splits -- and the text one then runs as a second, under-filled launch; together, 13 splits each, every CU walks the same
            # k-range once: 78.5 + 30.4 -> 97.8 us.  The chain rule of projection folding -- parameter-sized products that depend on
            # nothing this launch writes -- rides in its reduce launch: two latency-bound launches side by side instead of in a row.)
            # With a reducer (N > 1) too, since round 5: the per-modality launches that let the visual bucket leave early cost the step ~20 us of
            # launches and two more collectives; now everything behind the tables -- projection weights + (folded) fusion parameters, 3 MB,
            # the contiguous tail of the flat buffer -- is ONE collective behind this launch's reduce.
            chain_done = chain is not None
            grouped = True
        elif reducer is None:       # weight gradients of the first projections: two kernels, ONE reduce launch per pair of modalities
            self._gemm_deferred_pairs(dl1)
        if self.fold:
            if not chain_done:
                for i0 in range(0, nmod, 2):           # two modalities per launch: their d l2 (disjoint outputs) and ONE two-term problem for d Wkv
                    probs = self._fold_chain_products(i0)
                    call("snag_small_matmul", probs, len(probs), st)
            if reducer is not None and not grouped:
                reducer.start(buckets[1])
        bias_pair = cfg.num_proj == 1 and reducer is None and nmod == 2
        if bias_pair:       # db = colsum(dTok) of both modalities in one pair of launches
            call("snag_colsum_pair", dx0, dx0 + 4 * D, Nl, D, MD, ptr(self.g["proj0.bias"]), ptr(self.g["proj1.bias"]), 0, wk, st)
        for i, fd in enumerate(cfg.feat_dims):
            if reducer is not None and not grouped:         # data parallel: each modality's bucket goes out as soon as its gradient is final
                call("snag_gemm_f32", ctypes.byref(dl1[i]), st)
            if cfg.num_proj == 1 and not bias_pair:
                call("snag_colsum", dx0 + 4 * i * D, Nl, D, MD, ptr(self.g[f"proj{i}.bias"]), 0, wk, st)
            if reducer is not None and not grouped:
                reducer.start(buckets[2 + i])
        if reducer is not None and grouped:
            first_proj = self.slices["proj0.weight" if cfg.num_proj == 1 else "proj0.l1"][0]
            reducer.start(self.gflat[first_proj:] if self.fold else self.gflat[first_proj:self.slices["weight_raw"][0]])
        self._mark("backward")

    def optimizer_step(self, grad_scale=1.0):
        cfg = self.cfg
        coef, advanced = None, 0
        if cfg.clip > 0:
            # global-norm clipping + AdamW + the step counter in two launches: the coefficient never leaves the device (every AdamW
            # workgroup derives it from the sum-of-squares partials), self._clip receives {coefficient, norm}
            call("snag_clip_adamw_step_groups", ptr(self.flat), ptr(self.gflat), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n_params,
                 float(cfg.lr), 0.9, 0.999, 1e-8, ptr(self._seg_end), ptr(self._seg_wd), ptr(self._seg_lr), len(self.param_groups),
                 ptr(self.tick), float(grad_scale), float(cfg.clip), ptr(self._clip), ptr(self._clip_work), self._st)
            return
        call("snag_adamw_step_groups", ptr(self.flat), ptr(self.gflat), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n_params,
             float(cfg.lr), 0.9, 0.999, 1e-8, ptr(self._seg_end), ptr(self._seg_wd), ptr(self._seg_lr), len(self.param_groups), 0,
             ptr(self.tick), float(grad_scale), coef, advanced, self._st)
        if not advanced:
            call("snag_tick", ptr(self.tick), self._st)

    def train_step(self, h, r, t, neg_t, allreduce=None, sample=None):
        """forward + backward (+ gradient exchange) + AdamW.  ``allreduce(gflat)`` sums gradients across ranks.
        ``sample=(seed, known_sorted)``: the negatives are drawn into ``neg_t`` by the step's first launch (see ``forward``)."""
        self.forward(h, r, t, neg_t, with_backward=True, sample=sample)
        if allreduce is not None and hasattr(allreduce, "start"):
            self.backward(reducer=allreduce)
            allreduce.finish()
        else:
            self.backward()
            if allreduce is not None:
                if self.shard.sharded:
                    # this rank wrote ITS rows of the entity-table gradient only; the others still hold what the previous exchange
                    # left there, and a summed exchange would add them in once per rank
                    lo, hi = self.shard.lo, self.shard.lo + self.shard.n_loc
                    self.g_ent_pad[:lo].zero_()
                    self.g_ent_pad[hi:].zero_()
                allreduce(self.gflat)
        self.optimizer_step()
        return self.loss

    def _check_triples(self, h, r, t):
        """The kernels trust their indices (as a device-side embedding gather does); the training set is checked ONCE here -- one host
        sync per distinct (buffer, length), not per step -- and refused the way an out-of-range index into nn.Embedding is."""
        key = (h.data_ptr(), r.data_ptr(), t.data_ptr(), h.shape[0])
        if getattr(self, "_checked_triples", None) == key or h.shape[0] == 0:
            return
        if not (h.shape == r.shape == t.shape and h.dim() == 1):
            raise ValueError(f"train_epoch: h, r, t must be 1-D of equal length, got {tuple(h.shape)}, {tuple(r.shape)}, {tuple(t.shape)}")
        ends = torch.stack([h.min(), h.max(), t.min(), t.max(), r.min(), r.max()]).tolist()
        if ends[0] < 0 or ends[2] < 0 or max(ends[1], ends[3]) >= self.cfg.n_ent:
            raise IndexError(f"train_epoch: entity index out of range [0, {self.cfg.n_ent}): heads {ends[0]}..{ends[1]}, tails {ends[2]}..{ends[3]}")
        if ends[4] < 0 or ends[5] >= self.cfg.n_rel:
            raise IndexError(f"train_epoch: relation index out of range [0, {self.cfg.n_rel}): {ends[4]}..{ends[5]}")
        self._checked_triples = key

    def train_epoch(self, h, r, t, batch_size, known_sorted=None, allreduce=None, seed=None, rank=None, world=None):
        """One epoch in the order the README's MKGC loop implies (noise refresh once per epoch -- "noise-level: epoch" --, then every
        batch: NEG_NUM negative tails drawn on the device, forward, backward, exchange, clip, AdamW).  ``h, r, t``: this rank's int32
        training triples on the device.  Returns the summed loss as a device scalar.

        Under data parallelism (``allreduce`` given and a process group up) every rank runs the SAME number of steps
        (``parallel.common_step_count``; its own triples are cut into that many near-equal batches), and the rank is folded into the
        sampler's seed so that the ranks do not draw the identical negative-candidate stream."""
        import torch.distributed as dist
        from . import ops
        from .parallel import common_step_count, step_bounds
        n = h.shape[0]
        self._check_triples(h, r, t)
        dp = allreduce is not None and dist.is_available() and dist.is_initialized()
        group = getattr(allreduce, "group", None)          # a sub-group reducer: counts and ranks are those of ITS group
        if rank is None:
            rank = dist.get_rank(group) if dp else 0
        if world is None:
            world = dist.get_world_size(group) if dp else 1
        epoch_seed = self.cfg.seed * 104729 + self.noise_epoch if seed is None else seed
        sample_seed = epoch_seed * world + rank
        if self.cfg.add_noise:
            self.update_noise()
        # the permutation is per rank too: every rank shuffles its OWN shard
        perm = torch.randperm(n, device=self.dev, generator=torch.Generator(device=self.dev).manual_seed(sample_seed))
        steps = common_step_count(n, batch_size, group=group) if dp else -(-n // batch_size)
        total = torch.zeros((), device=self.dev, dtype=f32)
        for lo, hi in (step_bounds(n, steps) if dp else [(lo, min(n, lo + batch_size)) for lo in range(0, n, batch_size)]):
            idx = perm[lo:hi]
            hb, rb, tb = h[idx].contiguous(), r[idx].contiguous(), t[idx].contiguous()
            # drawn by the step's first launch; a rank with fewer items than steps runs a zero-gradient step, still part of every collective
            neg = torch.empty(hi - lo, self.cfg.neg_num, device=self.dev, dtype=torch.int32)
            total += self.train_step(hb, rb, tb, neg, allreduce=allreduce, sample=(sample_seed, known_sorted))
        return total

    # ---- hipGraph replay of the whole step ------------------------------------------------------------------------------
    def capture_step(self, batch, n_neg, sampler=None, sample=None):
        """Record forward + backward + clip + AdamW (and, when ``sampler`` is given, the on-device negative sampling) ONCE as a
        hipGraph over static index buffers; ``replay_step(h, r, t)`` then costs three small copies and one graph launch on the
        host.  Everything that varies per step already lives on the device: dropout seeds and Adam's bias correction read the
        tick counter, the clip coefficient never leaves HBM.  Single-process only (collectives are not captured); the epoch-level
        ``update_noise`` stays outside the graph (it writes the same buffers the graph reads).
        ``sampler(h, r, t, out)`` must enqueue on the current stream without allocating; ``sample=(seed, known_sorted)`` instead
        draws the negatives in the step's first launch (see ``forward``)."""
        dev = self.dev
        self._g_h = torch.zeros(batch, device=dev, dtype=torch.int32)
        self._g_r = torch.zeros(batch, device=dev, dtype=torch.int32)
        self._g_t = torch.zeros(batch, device=dev, dtype=torch.int32)
        self._g_neg = torch.zeros(batch, n_neg, device=dev, dtype=torch.int32)
        self._g_sampler = sampler

        def body():
            if sampler is not None:
                sampler(self._g_h, self._g_r, self._g_t, self._g_neg)
            self.train_step(self._g_h, self._g_r, self._g_t, self._g_neg, sample=sample)

        tick0 = self.tick.clone()
        snap = (self.flat.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # warm-up off the capture: buffers sized, code objects loaded
            body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for dst, src in zip((self.flat, self.exp_avg, self.exp_avg_sq), snap):      # the warm-up step must not count
            dst.copy_(src)
        self.tick.copy_(tick0)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            body()
        for dst, src in zip((self.flat, self.exp_avg, self.exp_avg_sq), snap):      # capture does not execute, but be explicit
            dst.copy_(src)
        self.tick.copy_(tick0)
        return self._graph

    def replay_step(self, h, r, t, neg_t=None):
        """One training step through the captured graph; returns the device scalar loss."""
        self._g_h.copy_(h, non_blocking=True)
        self._g_r.copy_(r, non_blocking=True)
        self._g_t.copy_(t, non_blocking=True)
        if neg_t is not None:
            self._g_neg.copy_(neg_t, non_blocking=True)
        self._graph.replay()
        return self.loss

    def entity_representation(self):
        """[N, dim] fused entity read-out of the last forward (a strided view of ``hidden`` for Mformer_hd_graph)."""
        if self.shard.sharded:
            return self.repr_full[:self.cfg.n_ent]
        if self.cfg.joint_way == "Mformer_hd_mean":
            return self.repr_mean
        if self.concat:
            return self.joint
        return self.hidden if (self.ro or self.atten) else self.hidden[:, self.g_slot, :]

    def link_prediction(self, h, r, t, filter_mask=None, top_k=(1, 10), method="exact", known_sorted=None):
        """Eval-mode fusion of the whole table, then the rank of each true tail against all entities (Hits@1/10, MR, MRR);
        ``method``: see snag_amd.evalsim.link_prediction."""
        from . import evalsim
        was = self.training
        self.eval()
        dummy = t.view(-1, 1)[:, :1].contiguous()
        self.forward(h, r, t, dummy)                     # fills hidden / read-out; the tiny score launch is discarded
        out = evalsim.link_prediction(self.entity_representation(), self.p["rel"], h, r, t, filter_mask, top_k, method=method, known_sorted=known_sorted)
        self.train(was)
        return out

    # parameters as a name -> tensor dict (every weight in the nn.Linear layout [out, in], as stored)
    def _logical(self, d):
        return dict(d)

    def named_parameters(self):
        return self._logical(self.p)

    # ---- checkpoints ---------------------------------------------------------------------------------------------------------
    def fusion_state_dict(self):
        """The Mformer part under the REFERENCE's state_dict keys (``weight_raw``, ``fusion_layer.{l}.attention.self.query.weight`` ...,
        model/SNAG_tools.py:159-298): loads into the reference's ``MformerFusion`` / ``snag_amd.fusion.MformerFusion`` unchanged."""
        D, out = self.cfg.dim, {"weight_raw": self.p["weight_raw"].clone()}
        for l in range(self.n_layers):
            pre = f"fusion_layer.{l}."
            for j, nm in enumerate(("query", "key", "value")):
                out[pre + f"attention.self.{nm}.weight"] = self.p[f"layer{l}.wqkv"][j * D:(j + 1) * D].clone()
                out[pre + f"attention.self.{nm}.bias"] = self.p[f"layer{l}.bqkv"][j * D:(j + 1) * D].clone()
            out[pre + "attention.output.dense.weight"] = self.p[f"layer{l}.wo"].clone()
            out[pre + "attention.output.dense.bias"] = self.p[f"layer{l}.bo"].clone()
            out[pre + "attention.output.LayerNorm.weight"] = self.p[f"layer{l}.ln1_g"].clone()
            out[pre + "attention.output.LayerNorm.bias"] = self.p[f"layer{l}.ln1_b"].clone()
            if self.cfg.use_intermediate:
                out[pre + "intermediate.dense.weight"] = self.p[f"layer{l}.w1"].clone()
                out[pre + "intermediate.dense.bias"] = self.p[f"layer{l}.b1"].clone()
                out[pre + "output.dense.weight"] = self.p[f"layer{l}.w2"].clone()
                out[pre + "output.dense.bias"] = self.p[f"layer{l}.b2"].clone()
                out[pre + "output.LayerNorm.weight"] = self.p[f"layer{l}.ln2_g"].clone()
                out[pre + "output.LayerNorm.bias"] = self.p[f"layer{l}.ln2_b"].clone()
        return out

    def load_fusion_state_dict(self, sd):
        """Inverse of ``fusion_state_dict`` (a reference checkpoint's fusion keys; unused ``output.*`` entries of FFN-less layers are ignored,
        as the reference constructs but never calls them, SNAG_tools.py:277)."""
        D = self.cfg.dim
        with torch.no_grad():
            self.p["weight_raw"].copy_(sd["weight_raw"])
            for l in range(self.n_layers):
                pre = f"fusion_layer.{l}."
                for j, nm in enumerate(("query", "key", "value")):
                    self.p[f"layer{l}.wqkv"][j * D:(j + 1) * D].copy_(sd[pre + f"attention.self.{nm}.weight"])
                    self.p[f"layer{l}.bqkv"][j * D:(j + 1) * D].copy_(sd[pre + f"attention.self.{nm}.bias"])
                for src, dst in (("attention.output.dense.weight", "wo"), ("attention.output.dense.bias", "bo"),
                                 ("attention.output.LayerNorm.weight", "ln1_g"), ("attention.output.LayerNorm.bias", "ln1_b")):
                    self.p[f"layer{l}.{dst}"].copy_(sd[pre + src])
                if self.cfg.use_intermediate:
                    for src, dst in (("intermediate.dense.weight", "w1"), ("intermediate.dense.bias", "b1"), ("output.dense.weight", "w2"),
                                     ("output.dense.bias", "b2"), ("output.LayerNorm.weight", "ln2_g"), ("output.LayerNorm.bias", "ln2_b")):
                        self.p[f"layer{l}.{dst}"].copy_(sd[pre + src])

    def state_dict(self):
        """Everything a resumed run needs: parameters (logical [out, in] layouts), both Adam moments, the device step counter, the
        noise epoch and the cached noise tensors."""
        from dataclasses import asdict
        moments = {}
        for name, buf in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            views = {k: buf[o:o + math.prod(shp)].view(shp) for k, (o, shp) in self.slices.items()}
            moments[name] = {k: v.clone() for k, v in self._logical(views).items()}
        return {"cfg": asdict(self.cfg), "pooled_from": self.pooled_from, "params": {k: v.clone() for k, v in self.named_parameters().items()}, **moments,
                "tick": self.tick.clone(), "noise_epoch": self.noise_epoch, "noise_active": self.noise_active,
                "noisy": [x.clone() for x in self.noisy], "ent_noise": self.ent_noise.clone(), "ent_noise_mask": self.ent_noise_mask.clone(),
                "row_mask": [[m.clone() for m in pair] for pair in self.row_mask]}

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, v in sd["params"].items():
                self.named_parameters()[k].copy_(v)
            for name, buf in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                views = self._logical({k: buf[o:o + math.prod(shp)].view(shp) for k, (o, shp) in self.slices.items()})
                for k, v in sd[name].items():
                    views[k].copy_(v)
            self.tick.copy_(sd["tick"])
            self.noise_epoch, self.noise_active = int(sd["noise_epoch"]), bool(sd["noise_active"])
            for i, x in enumerate(sd["noisy"]):
                self.noisy[i].copy_(x)
            self.ent_noise.copy_(sd["ent_noise"])
            self.ent_noise_mask.copy_(sd["ent_noise_mask"])
            if sd.get("row_mask") is not None:
                for pair, saved in zip(self.row_mask, sd["row_mask"]):    # the next refresh takes back the rows of the CURRENT mask
                    for m, x in zip(pair, saved):
                        m.copy_(x)
            else:
                # a checkpoint from before the incremental refresh: which rows carry noise is unknown, so every row counts as noisy -- the
                # next update_noise takes the noise back from all rows its new mask does not cover (stale noisy rows otherwise)
                for pair in self.row_mask:
                    for m in pair:
                        m.fill_(1)

    def save_checkpoint(self, path):
        """``torch.save`` of ``state_dict()`` (the reference's checkpoints are ``torch.save(model.state_dict())`` too, main.py:482-500)."""
        torch.save(self.state_dict(), path)
        return path

    def load_checkpoint(self, path):
        self.load_state_dict(torch.load(path, map_location=self.dev))
        return self

    def named_gradients(self):
        return self._logical(self.g)
