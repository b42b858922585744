"""BASELINE.json configs[4], the stress configuration: |E| = 1M, dim 256, 128 negative tails, bf16 tables, 8 GPUs.

The path is the negative-sampled margin scorer alone (reference operator: ``MSNEA.kge`` term 1, model/MSNEA.py:110-122, which scores
the embedding tables directly) at a size where the table (512 MB in bf16) no longer fits any cache and, across GPUs, the gradient
exchange -- not the kernels -- is the bound (SURVEY.md par. 8e "cfg 5").  Layout, per rank:

  * ``ent16`` [n_pad, D] bf16, ``rel16`` [R, D] bf16: REPLICATED tables, what ``snag_score_margin_fwd_bwd_bf16`` gathers from;
  * ``ent_master`` [n_shard, D] fp32 + Adam moments: the rows this rank OWNS (``parallel.RowShard``; 1/G of table and optimiser state);
  * ``d_ent_full`` [n_pad, D] fp32: the scatter-add target of this rank's triples (fp32 accumulation of bf16-table gradients).

One step: sample -> fused bf16 scorer (scores, loss, scattered gradients) -> REDUCE-SCATTER of the table gradient to the owners
(+ all-reduce of the small relation gradient) -> AdamW on the owned fp32 rows, writing their bf16 copy in the same pass ->
ALL-GATHER of the updated bf16 rows.  No dense all-reduce of the table ever happens (SURVEY par. 8e option 2).
``train_step`` returns per-phase HIP-event timers when asked, so that comm and compute are reported separately.
"""
from dataclasses import dataclass

import torch

from . import _lib
from ._lib import call, ptr, stream
from .parallel import RowShard

f32 = torch.float32


@dataclass
class StressConfig:
    n_ent: int = 1_000_000
    n_rel: int = 1000
    dim: int = 256
    neg_num: int = 128
    margin: float = 12.0
    lr: float = 1e-4
    weight_decay: float = 1e-4
    init_scale: float = 0.05            # tables ~ N(0,1) * 0.05 (SURVEY.md par. 8d "Config 5")
    scorer: str = "touched"             # "touched": segmented form over the batch's touched destinations (no float atomics, bit-reproducible,
                                        # snag_score_margin_fwd_bwd_touched_bf16); "atomic": one launch scatter-adding with fp32 atomics
    deterministic: bool = False         # with scorer="atomic": fixed-point scatter (bit-reproducible), two launches instead of the fused one
    seed: int = 1234


class StressHotPath:
    def __init__(self, cfg: StressConfig, device="cuda", shard: RowShard = None, ent_init=None, rel_init=None):
        _lib.require_gpu()
        self.cfg, self.dev = cfg, torch.device(device)
        N, R, D = cfg.n_ent, cfg.n_rel, cfg.dim
        if D % 8:
            raise ValueError("dim must be a multiple of 8 (16-byte bf16 chunks)")
        self.shard = shard if shard is not None else RowShard(N)
        if self.shard.n_rows != N:
            raise ValueError(f"shard covers {self.shard.n_rows} rows, the entity table has {N}")
        sh, dev = self.shard, self.dev
        g = torch.Generator(device=dev).manual_seed(cfg.seed)          # every rank draws the same full table, then keeps its rows
        full = torch.zeros(sh.n_pad, D, device=dev)
        if ent_init is not None:
            full[:N].copy_(ent_init)
        else:
            full[:N].normal_(generator=g).mul_(cfg.init_scale)
        self.rel_master = torch.empty(R, D, device=dev)
        if rel_init is not None:
            self.rel_master.copy_(rel_init)
        else:
            self.rel_master.normal_(generator=g).mul_(cfg.init_scale)
        self.ent16 = torch.empty(sh.n_pad, D, device=dev, dtype=torch.bfloat16)
        self.rel16 = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
        call("snag_f32_to_bf16", ptr(full), full.numel(), ptr(self.ent16), stream())
        call("snag_f32_to_bf16", ptr(self.rel_master), self.rel_master.numel(), ptr(self.rel16), stream())
        self.ent_master = full[sh.lo:sh.lo + sh.n_shard].clone()      # owned rows only; the full fp32 copy is dropped
        del full
        self.ent_m, self.ent_v = torch.zeros_like(self.ent_master), torch.zeros_like(self.ent_master)
        self.rel_m, self.rel_v = torch.zeros_like(self.rel_master), torch.zeros_like(self.rel_master)
        self.d_ent_full = torch.zeros(sh.n_pad, D, device=dev)
        self.d_ent = torch.empty(sh.n_shard, D, device=dev) if sh.sharded else self.d_ent_full
        self.d_rel = torch.zeros(R, D, device=dev)
        if cfg.deterministic:
            self._acc_ent = torch.zeros(N, D, device=dev, dtype=torch.int64)
            self._acc_rel = torch.zeros(R, D, device=dev, dtype=torch.int64)
        self.tick = torch.zeros(1, device=dev, dtype=torch.int64)
        self._ticket = torch.zeros(_lib.SCORE_TICKET_WORDS, device=dev, dtype=torch.int32)
        self.loss = torch.zeros((), device=dev)
        self._bufs = None
        self._touched_work = None
        if cfg.scorer not in ("touched", "atomic"):
            raise ValueError(f"scorer must be 'touched' or 'atomic', got {cfg.scorer!r}")

    def _outputs(self, B, K):
        if self._bufs is None or self._bufs[0].shape[0] != B or self._bufs[1].shape[1] != K:
            self._bufs = (torch.empty(B, device=self.dev), torch.empty(B, K, device=self.dev), torch.empty(B, device=self.dev))
        return self._bufs

    def score(self, h, r, t, neg_t):
        """Forward only: (pos [B], neg [B,K], loss) from the bf16 tables."""
        cfg = self.cfg
        B, K = neg_t.shape
        pos, neg, part = self._outputs(B, K)
        ib = 4 if h.dtype == torch.int32 else 8
        call("snag_score_margin_fwd_bf16", ptr(self.ent16), cfg.dim, cfg.n_ent, ptr(self.rel16), cfg.n_rel, cfg.dim, ptr(h), ptr(r), ptr(t), ptr(neg_t), ib,
             B, K, float(cfg.margin), ptr(pos), ptr(neg), ptr(part), stream())
        call("snag_reduce_sum", ptr(part), B, ptr(self.loss), stream())
        return pos, neg, self.loss

    def train_step(self, h, r, t, neg_t, timers=None):
        """One optimiser step; returns the device scalar loss.  ``timers``: dict that receives (start, end) torch events per phase
        ("score", "reduce_scatter", "adamw", "all_gather"): bench.py reports comm and compute separately from them."""
        import torch.distributed as dist
        cfg, sh = self.cfg, self.shard
        N, R, D = cfg.n_ent, cfg.n_rel, cfg.dim
        B, K = neg_t.shape
        pos, neg, part = self._outputs(B, K)
        ib = 4 if h.dtype == torch.int32 else 8
        st = stream()

        def phase(name):
            if timers is None:
                return None
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            timers.setdefault(name, []).append(e)
            return e

        phase("t0")
        if cfg.deterministic or cfg.scorer != "touched":
            self.d_rel.zero_()                   # (the touched scorer writes every relation row itself)
        if cfg.deterministic:
            self._acc_ent.zero_()
            self._acc_rel.zero_()
            self.score(h, r, t, neg_t)
            call("snag_score_margin_bwd_fixed_bf16", ptr(self.ent16), D, N, ptr(self.rel16), R, D, ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K,
                 float(cfg.margin), 1.0, None, None, ptr(self._acc_ent), ptr(self._acc_rel), st)
            call("snag_fixed_to_float", ptr(self._acc_ent), N, D, ptr(self.d_ent_full), D, 0, st)
            call("snag_fixed_to_float", ptr(self._acc_rel), R, D, ptr(self.d_rel), D, 0, st)
        elif cfg.scorer == "touched":
            # scores, loss, every touched entity gradient row and EVERY relation row WRITTEN (the rest of d_ent_full is zero: cleared row by row below)
            if self._touched_work is None or self._touched_work[0] != (B, K):
                nb = _lib.load().snag_score_touched_work_bytes(sh.n_pad, R, B, K, D)
                self._touched_work = ((B, K), torch.zeros((nb + 255) // 256 * 256, device=self.dev, dtype=torch.uint8))
            work = self._touched_work[1]
            call("snag_score_margin_fwd_bwd_touched_bf16", ptr(self.ent16), D, sh.n_pad, ptr(self.rel16), R, D, ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K,
                 float(cfg.margin), 1.0, ptr(pos), ptr(neg), ptr(part), ptr(self.loss), ptr(self.d_ent_full), D, ptr(self.d_rel), ptr(work), work.numel(), st)
        else:
            call("snag_score_margin_fwd_bwd_bf16", ptr(self.ent16), D, N, ptr(self.rel16), R, D, ptr(h), ptr(r), ptr(t), ptr(neg_t), ib, B, K,
                 float(cfg.margin), 1.0, ptr(pos), ptr(neg), ptr(part), ptr(self.loss), ptr(self._ticket), ptr(self.d_ent_full), D, ptr(self.d_rel), st)
        phase("t1")
        if sh.sharded:
            sh.reduce_scatter(self.d_ent, self.d_ent_full)                    # owners receive the summed gradient of their rows
            if sh.world > 1:
                dist.all_reduce(self.d_rel, group=sh.group)
        phase("t2")
        call("snag_adamw_step_bf16", ptr(self.ent_master), ptr(self.d_ent), ptr(self.ent_m), ptr(self.ent_v), self.ent_master.numel(), float(cfg.lr), 0.9, 0.999,
             1e-8, float(cfg.weight_decay), 0, ptr(self.tick), 1.0, self.ent16.data_ptr() + 2 * sh.lo * D, st)
        call("snag_adamw_step_bf16", ptr(self.rel_master), ptr(self.d_rel), ptr(self.rel_m), ptr(self.rel_v), self.rel_master.numel(), float(cfg.lr), 0.9, 0.999,
             1e-8, float(cfg.weight_decay), 0, ptr(self.tick), 1.0, ptr(self.rel16), st)
        call("snag_tick", ptr(self.tick), st)
        # the scatter target is cleared row by row where this batch touched it (136 MB instead of 1 GB at the config-5 size)
        if not cfg.deterministic:
            call("snag_zero_touched_rows", ptr(self.d_ent_full), D, D, ptr(h), ptr(t), ptr(neg_t), ib, B, K, st)
        phase("t3")
        if sh.sharded:
            sh.all_gather(self.ent16, self.ent16[sh.lo:sh.lo + sh.n_shard])   # every rank's updated bf16 rows, in place
        phase("t4")
        return self.loss

    def master_table(self):
        """This rank's owned fp32 rows [n_loc, D] (tests)."""
        return self.ent_master[:self.shard.n_loc]
