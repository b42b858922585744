#!/usr/bin/env python3
"""Benchmark of the MKGC hot path: triples/sec (fusion+score) at |E|=15k, dim=128, neg=32.

    python bench.py --gpus 1 --steps 40 --warmup 10
    python bench.py --gpus 8 --steps 40 --warmup 10        # starts the launcher below itself (one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of B=1024 synthetic DB15K-shaped triples per
GPU (BASELINE.json configs[1]): per-modality projections (num_proj=2) of the noisy feature tables ->
modality tokens -> 1-layer/2-head Mformer over ALL 15,000 entities (as the reference's
joint_emb_generat does each step, model/SNAG.py:166-181) -> Mformer_hd_graph read-out -> margin
score of 1 positive + 32 negative tails (drawn on the device each step, filtered against the known triples), its
backward through the whole chain, the gradient all-reduce (N>1, RCCL) and a fused AdamW update.  The epoch-level noise refresh (README "noise-level:
epoch") runs inside the timed region every 78th step.  All inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Share of the 1-GPU step (profiles/r03_step_timeline.txt, 519 us) that every rank repeats identically at N > 1:
#   replicated fusion: everything except sampling + scoring (~36 us) is full-table work done N times;
#   row-sharded: only the gradient norm + AdamW over the full 10.9 MB flat buffer (~27 us) is repeated.
REPLICATED_FRACTION_REPLICATED = 0.93
REPLICATED_FRACTION_SHARDED = 0.05
TIMER_EVERY = 4          # per-launch HIP events on every 4th timed step
SPLIT_PROLOGUE = bool(os.environ.get("BENCH_SPLIT_PROLOGUE"))
STEPS_PER_EPOCH = 78          # 79,222 DB15K training triples / 1024  (SURVEY.md par. 8d)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_MFMA_PEAK_TFLOPS = 2516.6          # dense (guide: ~2.5 PF); 16x the f32 matrix rate
F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)


def synth(cfg, batch, n_batches, seed, device):
    """SURVEY.md par. 8(d) generators (seed 1234): visual = row-normalised U(0,1), text = N(0,1); uniform triples/negatives."""
    g = torch.Generator().manual_seed(seed)
    N = cfg.n_ent
    vis = torch.rand(N, cfg.feat_dims[0], generator=g)
    vis = vis / vis.norm(dim=1, keepdim=True)
    txt = torch.randn(N, cfg.feat_dims[1], generator=g)
    gi = torch.Generator().manual_seed(seed + 1 + int(os.environ.get("RANK", "0")))      # sampling differs per rank, weights do not
    # one epoch of positives (78 batches ~ 79,222 DB15K training triples), resident on the device; negatives are drawn on the
    # device every step (snag_sample_negatives, known-triple filter over this rank's positives)
    h = torch.randint(0, N, (n_batches * batch,), generator=gi, dtype=torch.int32).to(device)
    r = torch.randint(0, cfg.n_rel, (n_batches * batch,), generator=gi, dtype=torch.int32).to(device)
    t = torch.randint(0, N, (n_batches * batch,), generator=gi, dtype=torch.int32).to(device)
    batches = [tuple(x[i * batch:(i + 1) * batch] for x in (h, r, t)) for i in range(n_batches)]
    from snag_amd import ops
    known = ops.triple_keys(h, r, t, N, cfg.n_rel)
    return [vis, txt], batches, known


class KernelTimer:
    """HIP events on the launch stream around one tagged launch per step (torch events record on the current stream,
    which is the stream every snag_amd launch uses)."""

    def __init__(self):
        self.pairs = []

    def wrap(self, fn):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.pairs.append((e0, e1))
            return out
        return inner

    def mean_us(self, skip=0):
        v = [a.elapsed_time(b) * 1e3 for a, b in self.pairs[skip:]]
        return sum(v) / max(1, len(v)), len(v)


def score_bytes(cfg, n_triples):
    """SURVEY.md par. 8(d): reads (K+3) rows of D fp32 + (K+3) int32 ids, writes (K+1) fp32 scores per triple."""
    return n_triples * ((cfg.neg_num + 3) * cfg.dim * 4 + (cfg.neg_num + 3) * 4 + (cfg.neg_num + 1) * 4)


def roofline_dominant(cfg, timer, rows=None, reduce_timer=None, grouped=False):
    """The launch that takes the largest share of the step: the first projection over this rank's entity rows -- the visual one
    alone, or (``grouped``: two modalities, snag_gemm_f32_group) the visual and the text one in ONE launch of eight-wave tiles with
    balanced K splits; work and bytes are then those of both.

    fp32 in / fp32 accumulate; the multiplies run as six bf16 MFMA products of exactly split operands (include/snag_hip.h, snag_gemm_f32)."""
    us, n = timer.mean_us()
    if reduce_timer is not None:        # + the whole reduce launch (conservative: all of it counted here)
        us += reduce_timer.mean_us()[0]
    rows = cfg.n_ent if rows is None else rows
    dims = list(cfg.feat_dims[:2]) if grouped else [cfg.feat_dims[0]]
    flops = sum(2.0 * rows * fd * cfg.dim for fd in dims)
    alg_bytes = sum(4 * (rows * fd + fd * cfg.dim + rows * cfg.dim) for fd in dims)
    ach = flops / (us * 1e-6) / 1e12 if us else 0.0
    traffic, src = None, None
    try:  # PMC passes cannot run inside this process; the committed per-launch figure is quoted when it is for this very shape
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_dominant.json")) as f:
            pmc = json.load(f)
        if (pmc["n_ent"], pmc["feat0"], pmc["dim"], bool(pmc.get("grouped", False))) == (rows, cfg.feat_dims[0], cfg.dim, bool(grouped)):
            # the timed call is the GEMM kernel AND its split-K reduce launch: both kernels' HBM bytes
            traffic, src = pmc["fetch_bytes"] + pmc["write_bytes"] + pmc.get("fixup_fetch_bytes", 0) + pmc.get("fixup_write_bytes", 0), pmc["source"]
    except (OSError, KeyError, ValueError):
        pass
    shapes = " + ".join(f"[{rows},{fd}]x[{cfg.dim},{fd}]^T" for fd in dims)
    kern = (f"gemm_tile_group_kernel<256,128,32,128,A K-contiguous,B K-contiguous> (8-wave ping-pong tiles; the visual and the text projection in ONE launch, K splits "
            f"balanced across both) + the split-K reduce of the visual one ({shapes}, fp32 via 6 split-bf16 MFMA products)" if grouped else
            f"gemm_tile_kernel<256,128,32,128,A K-contiguous,B K-contiguous> (8-wave ping-pong) + splitk_reduce_multi_kernel (shared with the text projection, "
            f"counted in full) (visual projection: {shapes}, fp32 via 6 split-bf16 MFMA products, 4 K-splits)")
    return {"kernel": kern,
            # the bound of the kernel AS BUILT: six bf16 MFMA products per f32 multiply-add -> bf16 dense peak / 6 (37.5 us for this
            # launch; its HBM floor, 255.5 MB at 8 TB/s, is 32 us).  frac is therefore the share of the matrix pipe's time the
            # launch (kernel + fix-up) uses; against the f32 MFMA peak (157.3 TF), which bounded the previous form of this kernel
            # (0.66 in round 1), the same time reads as frac_of_f32_mfma_peak.
            "bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK_TFLOPS / 6, "unit": "TFLOP/s (f32 multiply-adds)", "frac": ach / (BF16_MFMA_PEAK_TFLOPS / 6),
            "peak_basis": "bf16 dense MFMA peak 2516.6 TF / 6 products per f32 multiply-add",
            "time_vs_round1_f32_mfma_bound": ach / F32_MFMA_PEAK_TFLOPS,      # (the kernel issues no f32 MFMAs: how this launch compares with the bound of the round-1 kernel)
            "hbm_frac": alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us else 0.0,
            "traffic": traffic, "traffic_unit": "bytes/launch (HBM, rocprofv3 PMC)", "traffic_source": src,
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_us": us, "avg_us_source": "HIP events (torch events on the launch stream) around the GEMM launch + the whole shared reduce launch, inside the timed "
                                           "region; an event pair adds ~5-10 us of stream bubbles: the rocprofv3 per-kernel durations under profiles/ are the tighter figure",
            "launches_timed": n, "algorithmic_flops_per_launch": flops}


def score_train_bytes(cfg, n_triples):
    """SURVEY.md par. 8(d), forward + hinge backward in one launch: the forward's bytes (rows read ONCE) plus the read-modify-write
    of (K+3) gradient rows of D fp32 per triple (no upstream-gradient read in hinge mode)."""
    return score_bytes(cfg, n_triples) + n_triples * (cfg.neg_num + 3) * cfg.dim * 4 * 2


def roofline_score(cfg, batch, timer, ln_fused=False, ln_dropout=True):
    """north_star's kernel in its training form (snag_score_margin_fwd_bwd_seg): per 1024-triple batch a per-triple pass (every table
    row gathered ONCE; scores, losses, q / dq rows, one 8-byte entry per touched destination) and a per-destination pass (gathers the
    L2-resident q / dq rows, applies the normalisation backward once, WRITES each gradient row: no float atomics, no zero-fill,
    bit-reproducible), vs the 8 TB/s HBM figure.  The 7.7 MB table is cache-resident at this size: `achieved` is ALGORITHMIC bytes
    (SURVEY par. 8d: rows read once + read-modify-write of the K+3 gradient rows per triple) / time of the two launches together.
    ``hbm_resident`` (added by main) holds the forward kernel AND the training form (``train_bf16_B*``) on the 1M x 256 table that no
    cache holds; ``batch_8192`` the same two launches with eight times the work."""
    us, n = timer.mean_us()
    by = score_train_bytes(cfg, batch)
    # the entry the headline engine calls since round 4 also carries the gradient through the read-out rows' LayerNorm inside pass B
    # (snag_score_margin_fwd_bwd_seg_ln): the separate LayerNorm-backward launch's bytes -- the pre-LN rows read, dT and dT * keep written,
    # the keep mask read -- belong to the timed launches, while the scorer's own gradient rows are no longer written
    # (the dT * keep rows and the keep-mask bytes only when the engine runs with the output dropout on: snag_ln_bwd_tail.dtm set)
    ln_bytes = cfg.n_ent * cfg.dim * (4 + 4 + ((4 + 1) if ln_dropout else 0)) - cfg.n_ent * cfg.dim * 4 if ln_fused else 0
    by += ln_bytes
    ach = by / (us * 1e-6) / 1e9 if us else 0.0
    atomic_bytes = batch * (cfg.neg_num + 3) * cfg.dim * 4
    return {"kernel": ("score_seg_a_kernel + score_seg_b_kernel<LN> (snag_score_margin_fwd_bwd_seg_ln: fwd + hinge bwd + summed loss + the LayerNorm backward of "
                       "the read-out rows, no float atomics)" if ln_fused else
                       "score_seg_a_kernel + score_seg_b_kernel (snag_score_margin_fwd_bwd_seg: fwd + hinge bwd + summed loss, no float atomics)"),
            "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_us": us, "launches_timed": n,
            "algorithmic_bytes_per_launch": by, "layernorm_backward_bytes_included": ln_bytes, "forward_only_bytes_per_launch": score_bytes(cfg, batch),
            "fp32_atomic_form": {"entry": "snag_score_margin_fwd_bwd (MKGCConfig.scorer='atomic')", "atomic_bytes_per_launch": atomic_bytes,
                                 "atomic_floor_us": atomic_bytes / 1.3e12 * 1e6},
            "timed": "HIP events around the entry (both passes), inside the timed region; rocprofv3 per-kernel times: profiles/"}


def score_train_larger_batch(eng, cfg, device, B=8192):
    """The training scorer (same entry, same 15k-entity read-out table) on 8 x the batch: at B = 1024 the two passes are launch- and
    round-trip-bound (18.6 MB of gathers would take 2.3 us at 8 TB/s); this shows what the kernels do once there is work per launch."""
    from snag_amd._lib import call, ptr, stream
    from snag_amd import _lib
    N, R, D, K = cfg.n_ent, cfg.n_rel, eng.sdim, cfg.neg_num
    g = torch.Generator().manual_seed(11)
    h, r, t = (torch.randint(0, hi, (B,), generator=g, dtype=torch.int32).to(device) for hi in (N, R, N))
    neg = torch.randint(0, N, (B, K), generator=g, dtype=torch.int32).to(device)
    table = eng.entity_representation().contiguous()
    pos, ng, lp, loss = torch.empty(B, device=device), torch.empty(B, K, device=device), torch.empty(B, device=device), torch.empty((), device=device)
    de, dr = torch.empty(N, D, device=device), torch.empty(R, D, device=device)
    nb = _lib.load().snag_score_seg_work_bytes(N, R, B, K, D)
    work = torch.zeros(nb, device=device, dtype=torch.uint8)
    tick = torch.zeros(_lib.SCORE_TICKET_WORDS, device=device, dtype=torch.int32)
    run = lambda: call("snag_score_margin_fwd_bwd_seg", ptr(table), D, N, ptr(eng.p["rel"]), R, D, ptr(h), ptr(r), ptr(t), ptr(neg), 4, B, K, float(cfg.margin), 1.0,
                       ptr(pos), ptr(ng), ptr(lp), ptr(loss), ptr(tick), ptr(de), D, ptr(dr), 1, ptr(work), nb, stream())
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    by = B * ((K + 3) * D * 4 * 3 + (K + 3) * 4 + (K + 1) * 4)
    return {"batch": B, "avg_us": us, "achieved": by / us / 1e3, "unit": "GB/s", "frac": by / us / 1e3 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": by,
            "timed": "20 back-to-back calls of the entry (both passes), HIP events, after the timed region"}


def untimed_extras(eng, batches, args, device, sample=None):
    """Reported next to the headline: the forward-only rate (SURVEY.md par. 8d asks for both).  Measured on every rank straight BEFORE
    the warm-up steps (forward passes change no state): the same kernels as the step, so the device enters the driver's short warm-up
    (5 steps = 2 ms) at the clocks it holds under this load.  From idle the device's power management takes ~35 ms of load to settle
    (tools/probe_ramp.py: step 1 447 us, step 8 520, step 25 450, step 60 430), so the rate is taken over 300 passes (60 ms)."""
    out = {}
    neg = torch.randint(0, eng.cfg.n_ent, (args.batch, eng.cfg.neg_num), device=device, dtype=torch.int32)
    probe, eng.probe = eng.probe, None            # no per-launch events here: creating them would make this loop host-bound
    for i in range(3):
        eng.forward(*batches[i % len(batches)], neg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = max(args.steps, 300)
    for i in range(n):
        eng.forward(*batches[i % len(batches)], neg)
    torch.cuda.synchronize()
    out["forward_only"] = {"value": args.batch * n / (time.perf_counter() - t0), "unit": "triples/s", "passes": n}
    # ... and forward + backward without the optimiser (par. 8d names both): gradients are overwritten by every pass, parameters, moments
    # and the step counter stay as they are.  The last thing the device does before the warm-up steps: the step's own kernels but clip + AdamW.
    if sample is not None:
        t0 = time.perf_counter()
        n2 = max(args.steps, 150)
        for i in range(n2):
            eng.forward(*batches[i % len(batches)], neg, with_backward=True, grad_scale=1.0, sample=sample)
            eng.backward(1.0)
        torch.cuda.synchronize()
        out["forward_backward"] = {"value": args.batch * n2 / (time.perf_counter() - t0), "unit": "triples/s", "passes": n2}
    eng.probe = probe
    return out


def noise_refresh_us(eng, reps=8):
    """One noise refresh (README 'noise-level'; model/SNAG.py:86-98: three row masks, both feature tables' blend incl. their transposed
    copies, the entity table's column statistics and noise rows), HIP events around ``update_noise`` on the launch stream, mean of
    ``reps`` back-to-back refreshes after two untimed ones.  State (noise epoch, masks, noisy tables) moves on as in training."""
    for _ in range(2):
        eng.update_noise()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        eng.update_noise()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def phases_summary(phase_marks, world, row_sharded, cfg, eng):
    """Mean duration of every phase of the step between consecutive HIP events on the compute stream (rank 0).  The compute stream
    WAITS for each collective, so "all_gather" / "reduce_scatter" / "allreduce_wait" are the time the stream could not compute: a
    collective's exposed cost, not its wire time."""
    acc = {}
    for marks in phase_marks:
        for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
            acc.setdefault(name, []).append(a.elapsed_time(b) * 1e3)
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    comm = sum(out.get(k, 0.0) for k in ("all_gather", "reduce_scatter", "allreduce_wait"))
    total = sum(out.values())
    D, W = cfg.dim, eng.sdim
    ns = eng.shard.n_shard
    out["comm_exposed_us"] = comm
    out["compute_us"] = total - comm
    out["steps_sampled"] = len(phase_marks)
    out["order"] = [n for n, _ in phase_marks[0][1:]]
    if world > 1:
        out["comm_bytes_per_rank"] = ({"all_gather_readout_recv": (world - 1) * ns * W * 4, "reduce_scatter_readout_grad_recv": (world - 1) * ns * W * 4,
                                       "all_gather_entity_grad_recv": (world - 1) * ns * D * 4,
                                       "all_reduce_dense_grads": 4 * (eng.n_params - eng.slices["rel"][0])} if row_sharded
                                      else {"all_reduce_flat_grads": 4 * eng.n_params})
    return out


def device_copy_ceiling(device, reps=5):
    """Device-to-device copy rate of this GPU (1 GiB src -> 1 GiB dst), reported next to the nominal 8 TB/s."""
    src = torch.empty(1 << 28, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return reps * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def score_hbm_resident(device):
    """The scoring kernel where it IS HBM-bound (BASELINE.json configs[4], the 1M-entity stress shape: |E| = 1M, D = 256, K = 128; a 1 GB fp32 / 512 MB bf16 table, far
    beyond the 256 MB Infinity Cache), uniform indices, per launch: the north_star's "fraction of the HBM roofline on the neg-sampled
    scoring kernel" without the launch-latency floor that dominates the 18.6 MB launch of the 15k-entity step."""
    from snag_amd._lib import call, ptr, stream
    N, R, D, K = 1_000_000, 1000, 256, 128
    g = torch.Generator().manual_seed(5)
    ent = torch.empty(N, D, device=device).normal_()
    rel = torch.randn(R, D, generator=g).to(device)
    tables = {"f32": (ent, rel, "snag_score_margin_fwd", 4), "bf16": (ent.to(torch.bfloat16), rel.to(torch.bfloat16), "snag_score_margin_fwd_bf16", 2)}
    out = {"table": "|E|=1M x 256, K=128 negatives, uniform indices", "peak_GBs": HBM_PEAK_GBS}
    for dt, (e, rl, fn, esz) in tables.items():
        for B in (1024, 32768):
            idx = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).to(device) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
            pos, ng, lp = torch.empty(B, device=device), torch.empty(B, K, device=device), torch.empty(B, device=device)
            run = lambda: call(fn, ptr(e), D, N, ptr(rl), R, D, ptr(idx[0]), ptr(idx[1]), ptr(idx[2]), ptr(idx[3]), 4, B, K, 12.0, ptr(pos), ptr(ng), ptr(lp), stream())
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            by = B * ((K + 3) * D * esz + (K + 3) * 4 + (K + 1) * 4)
            out[f"{dt}_B{B}"] = {"avg_us": us, "achieved": by / us / 1e3, "unit": "GB/s", "frac": by / us / 1e3 / HBM_PEAK_GBS, "triples_per_s": B / us * 1e6}
    # the TRAINING form on the same table (bf16 rows, fp32 gradient rows): forward bytes + read-modify-write of the (K + 3) gradient rows per
    # triple (SURVEY par. 8d) over the time of the entry's launches (segmented form over the batch's entries: no float atomics).  The entry is
    # called straight through the C ABI over preallocated buffers, as the stress engine calls it: the Python wrapper's per-call allocations are
    # host time, and at B = 1,024 they, not the device, paced a back-to-back loop
    from snag_amd import _lib
    e16, r16 = tables["bf16"][0], tables["bf16"][1]
    d_ent, d_rel = torch.zeros(N, D, device=device), torch.zeros(R, D, device=device)
    for B in (1024, 8192):
        idx = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).to(device) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
        pos, ng, lp, loss = torch.empty(B, device=device), torch.empty(B, K, device=device), torch.empty(B, device=device), torch.empty((), device=device)
        nb = _lib.load().snag_score_touched_work_bytes(N, R, B, K, D)
        work = torch.zeros((nb + 255) // 256 * 256, device=device, dtype=torch.uint8)
        run = lambda: call("snag_score_margin_fwd_bwd_touched_bf16", ptr(e16), D, N, ptr(r16), R, D, ptr(idx[0]), ptr(idx[1]), ptr(idx[2]), ptr(idx[3]), 4, B, K, 12.0, 1.0,
                           ptr(pos), ptr(ng), ptr(lp), ptr(loss), ptr(d_ent), D, ptr(d_rel), ptr(work), work.numel(), stream())
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        by = B * ((K + 3) * D * 2 + (K + 3) * 4 + (K + 1) * 4) + B * (K + 3) * D * 4 * 2
        out[f"train_bf16_B{B}"] = {"entry": "snag_score_margin_fwd_bwd_touched_bf16 (pass A writes the single-entry gradient rows itself; pass B one lane per record + the "
                                            "relation role; hot-row launch)", "avg_us": us, "achieved": by / us / 1e3, "unit": "GB/s",
                                   "frac": by / us / 1e3 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": by, "triples_per_s": B / us * 1e6}
        del work
    return out


def roofline_sim(device, reps=5):
    """BASELINE configs[3] (DBP15K ja_en MMEA eval: 10,500 aligned test pairs x 1,200-wide joint embeddings, CSLS k = 3; SURVEY par. 8d):
    the similarity GEMM with its distance epilogue (`snag_pairwise_dist`, src/utils.py:202-218) against the MFMA roofline of the six
    split-bf16 products, and the whole `rank_metrics` (GEMM + CSLS neighbourhoods + ranks + summaries, main.py:376-436), HIP events on
    the launch stream, mean of ``reps`` back-to-back calls after one warm-up.  Returns (record, left, right) -- the embeddings feed the
    oracle check of the cpu_baseline leg."""
    from snag_amd import evalsim
    n, d, k = 10500, 1200, 3
    g = torch.Generator().manual_seed(4321)
    left = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1)
    # planted signal (SURVEY par. 8d): right = normalise(left + noise); noise of norm ~8 puts the true pair ~4 sigma above a random one, so
    # Hits@1 is neither 0 nor 1
    right = torch.nn.functional.normalize(left + 8.0 * torch.randn(n, d, generator=g) / d ** 0.5, dim=1)
    L, Rt = left.to(device), right.to(device)

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps, out

    gemm_us, _ = timed(lambda: evalsim.pairwise_distances(L, Rt, _padded=True))
    eval_us, res = timed(lambda: evalsim.rank_metrics(L, Rt, csls_k=k, top_k=(1, 10, 50)))
    flops = 2.0 * n * n * d
    peak = BF16_MFMA_PEAK_TFLOPS / 6.0
    rec = {"workload": f"configs[3]: {n} x {n} x {d} similarity GEMM + distance epilogue, CSLS k={k}, ranks both directions", "bound": "mfma",
           "gemm_avg_us": gemm_us, "achieved": flops / gemm_us / 1e6, "peak": peak, "unit": "TFLOP/s", "frac": flops / gemm_us / 1e6 / peak,
           "peak_is": "bf16 dense MFMA peak / 6 (f32 products as six split-bf16 MFMA products)", "flops_per_launch": flops,
           "matrix_bytes": n * n * 4, "rank_metrics_ms": eval_us / 1e3, "hits_l2r": res["hits_l2r"], "hits_r2l": res["hits_r2l"],
           "timed": f"{reps} back-to-back calls, HIP events, outside the timed region"}
    return rec, left, right


def cpu_baseline_sim_check(left, right, device, rows=512, k=3, window=5e-6, full=True):
    """Part of the cpu_baseline leg: the configs[3] workload, GPU path vs the CPU oracle's restated Runner._test (oracle.rank_metrics) --
    north_star: "Hits@1/10 equal to the CPU reference", "bit-exact for ranking indices".  A ``rows``-pair sub-block, and (``full``) the whole
    10,500 x 10,500 problem: Hits, the number of ranks that differ per direction and how many of those are proven near-ties (a candidate within
    ``window`` of the true match in the oracle's CSLS-adjusted distances; the window must cover twice the largest GPU-vs-oracle difference)."""
    from oracle import ref_cpu as O
    from snag_amd import evalsim
    l, r = left[:rows].contiguous(), right[:rows].contiguous()
    got = evalsim.rank_metrics(l.to(device), r.to(device), csls_k=k, top_k=(1, 10))
    ref = O.rank_metrics(l, r, csls_k=k, top_k=(1, 10))
    same = bool((got["rank_l2r"].cpu() == ref["rank_l2r"]).all()) and bool((got["rank_r2l"].cpu() == ref["rank_r2l"]).all())
    out = {"rows": rows, "hits_l2r_gpu": got["hits_l2r"], "hits_l2r_oracle": ref["hits_l2r"], "hits_r2l_gpu": got["hits_r2l"], "hits_r2l_oracle": ref["hits_r2l"],
           "hits_equal": got["hits_l2r"] == ref["hits_l2r"] and got["hits_r2l"] == ref["hits_r2l"], "rank_arrays_equal": same,
           "n_rank_diffs": int((got["rank_l2r"].cpu() != ref["rank_l2r"]).sum()) + int((got["rank_r2l"].cpu() != ref["rank_r2l"]).sum())}
    if full:
        t0 = time.perf_counter()
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        n = left.shape[0]
        res = evalsim.rank_metrics(left.to(device), right.to(device), csls_k=k, top_k=(1, 10))
        ro = O.rank_metrics(left, right, csls_k=k, top_k=(1, 10))
        d_o = ro["dist"]
        d_g = 1 - O.csls_sim(1 - res["dist"].cpu(), k)
        delta = float((d_g - d_o).abs().max())
        diag = torch.arange(n)
        fl = {"n": n, "max_abs_dist_diff": delta, "window_covers_2_delta": 2 * delta <= window,
              "hits_l2r_gpu": res["hits_l2r"], "hits_l2r_oracle": ro["hits_l2r"], "hits_r2l_gpu": res["hits_r2l"], "hits_r2l_oracle": ro["hits_r2l"]}
        for name, scores in (("l2r", d_o), ("r2l", d_o.t())):
            rep = near_tie_report(res[f"rank_{name}"].cpu(), ro[f"rank_{name}"], scores, diag, window)
            gk, ok = res[f"rank_{name}"].cpu(), ro[f"rank_{name}"]
            rep["n_hits_affecting_diffs"] = int((((gk < 1) != (ok < 1)) | ((gk < 10) != (ok < 10))).sum())
            fl[name] = rep
        fl["hits_equal"] = res["hits_l2r"] == ro["hits_l2r"] and res["hits_r2l"] == ro["hits_r2l"]
        fl["seconds"] = time.perf_counter() - t0
        out["full_size"] = fl
    return out


def oracle_param_dicts(eng, dtype):
    """The engine's flat-buffer views as the oracle's parameter dicts (reference state_dict keys), on the CPU in ``dtype``."""
    cfg, D = eng.cfg, eng.cfg.dim
    p = {k: v.detach().cpu().to(dtype) for k, v in eng.named_parameters().items()}
    proj = [({"weight": p[f"proj{i}.weight"], "bias": p[f"proj{i}.bias"]} if cfg.num_proj == 1 else {"l1": p[f"proj{i}.l1"], "l2": p[f"proj{i}.l2"]})
            for i in range(len(cfg.feat_dims))]
    fp = {"weight_raw": p["weight_raw"]}
    for l in range(eng.n_layers):
        pre = f"fusion_layer.{l}."
        for j, nm in enumerate(("query", "key", "value")):
            fp[pre + f"attention.self.{nm}.weight"] = p[f"layer{l}.wqkv"][j * D:(j + 1) * D]
            fp[pre + f"attention.self.{nm}.bias"] = p[f"layer{l}.bqkv"][j * D:(j + 1) * D]
        fp[pre + "attention.output.dense.weight"], fp[pre + "attention.output.dense.bias"] = p[f"layer{l}.wo"], p[f"layer{l}.bo"]
        fp[pre + "attention.output.LayerNorm.weight"], fp[pre + "attention.output.LayerNorm.bias"] = p[f"layer{l}.ln1_g"], p[f"layer{l}.ln1_b"]
        if cfg.use_intermediate:
            for src, dst in (("intermediate.dense.weight", "w1"), ("intermediate.dense.bias", "b1"), ("output.dense.weight", "w2"),
                             ("output.dense.bias", "b2"), ("output.LayerNorm.weight", "ln2_g"), ("output.LayerNorm.bias", "ln2_b")):
                fp[pre + src] = p[f"layer{l}.{dst}"]
    return p, proj, fp


def near_tie_report(got_rank, ref_rank, ref_scores, true_col, window, blocked=None):
    """Counts behind "bit-exact for ranking indices": how many ranks differ from the oracle's, and how many of those differences are PROVEN
    near-ties -- |rank_gpu - rank_oracle| <= the number of unfiltered candidates whose float64 oracle score lies within ``window`` of the true
    candidate's (only such a candidate can change sides under fp32 rounding)."""
    diff = (got_rank.long() - ref_rank.long()).abs()
    idx = diff.nonzero().flatten().tolist()
    explained = 0
    for b in idx:
        line = ref_scores[b]
        close = (line - line[int(true_col[b])]).abs() <= window
        if blocked is not None:
            close = close & ~blocked[b].bool()
        if int(diff[b]) <= int(close.sum()) - 1:
            explained += 1
    return {"n": int(got_rank.numel()), "ranks_equal": len(idx) == 0, "n_rank_diffs": len(idx), "n_near_tie_explained": explained,
            "n_unexplained": len(idx) - explained, "max_rank_diff": int(diff.max()) if diff.numel() else 0, "window": window}


def hits_parity(eng, feats_cpu, known, device, n_queries=512, window=5e-6, seed=777):
    """BASELINE metric, second half ("Hits@1/10 parity"; north_star: "Hits@1/10 equal to the CPU reference ... bit-exact for ranking indices"),
    on THIS engine after its timed steps: filtered tail prediction of ``n_queries`` held-out triples against all |E| candidates,
    ``eng.link_prediction`` (eval-mode fusion of the whole table, snag_score_margin_fwd + snag_rank_full_table) next to the oracle's
    ``mkgc_forward`` -> ``rank_full_table`` in float64 on the same parameters.  Half of the queries are uniform held-out triples (their true
    tail ranks anywhere in 0..|E|: the regime where near-ties are dense); the other half take as true tail the oracle's j-th best candidate,
    j uniform in [0, 30), so that Hits@1 and Hits@10 are neither 0 nor 1.  Hits@k = round(mean(rank < k), 4) as main.py:430-436 forms them;
    score function MSNEA.py:110-127.  Part of the cpu_baseline leg (it runs the oracle)."""
    from oracle import ref_cpu as O
    cfg = eng.cfg
    N, R = cfg.n_ent, cfg.n_rel
    g = torch.Generator().manual_seed(seed)
    h = torch.randint(0, N, (n_queries,), generator=g, dtype=torch.int32)
    r = torch.randint(0, R, (n_queries,), generator=g, dtype=torch.int32)
    t = torch.randint(0, N, (n_queries,), generator=g, dtype=torch.int32)
    t0 = time.perf_counter()
    p, proj, fp = oracle_param_dicts(eng, torch.float64)
    with torch.no_grad():
        dummy = torch.zeros(1, 1, dtype=torch.int64)
        _, _, _, oent = O.mkgc_forward([f.double() for f in feats_cpu], proj, p["ent"], fp, p["rel"], h[:1], r[:1], t[:1], dummy, num_heads=cfg.num_heads,
                                       margin=cfg.margin, joint_way=cfg.joint_way, use_intermediate=bool(cfg.use_intermediate), num_layers=cfg.num_layers,
                                       atten_a=p.get("atten.a"))
        s64 = O.full_table_scores(oent, p["rel"], h, r)                               # [B, N] float64
        half = n_queries // 2
        j = torch.randint(0, 30, (n_queries - half,), generator=g)
        order = torch.argsort(s64[half:], dim=1)
        t[half:] = order.gather(1, j.view(-1, 1)).flatten().to(torch.int32)
        # filtered protocol: every known (h, r, x) with x != t leaves the candidate list (keys as ops.triple_keys packs them)
        keys = known.cpu()
        blocked = torch.zeros(n_queries, N, dtype=torch.bool)
        base = (h.long() * R + r.long()) * N
        lo, hi = torch.searchsorted(keys, base), torch.searchsorted(keys, base + N)
        for b in range(n_queries):
            if hi[b] > lo[b]:
                blocked[b, (keys[lo[b]:hi[b]] - base[b])] = True
            blocked[b, int(t[b])] = False
        st = s64.gather(1, t.long().view(-1, 1))
        oref = ((s64 < st) & ~blocked).sum(dim=1)
    oracle_s = time.perf_counter() - t0
    res = eng.link_prediction(h.to(device), r.to(device), t.to(device), known_sorted=known, top_k=(1, 10))
    got = res["rank"].cpu()
    rep = near_tie_report(got, oref, s64, t, window, blocked)
    hits_o = [round(float((oref < k).double().mean()), 4) for k in (1, 10)]
    score_err = float((res["scores"].cpu().double() - s64).abs().max())
    rep.update({"queries": f"{half} uniform held-out triples + {n_queries - half} whose true tail is the oracle's j-th best candidate (j < 30); filtered against the "
                           f"{keys.numel()} known training triples; {N} candidates each",
                "hits1_gpu": res["hits"][0], "hits10_gpu": res["hits"][1], "hits1_oracle": hits_o[0], "hits10_oracle": hits_o[1],
                "hits_equal": res["hits"][0] == hits_o[0] and res["hits"][1] == hits_o[1],
                "n_hits_affecting_diffs": int((((got < 1) != (oref < 1)) | ((got < 10) != (oref < 10))).sum()),
                "mr_gpu": res["mr"], "mr_oracle": float((oref.double() + 1).mean()), "max_abs_score_err_vs_f64": score_err,
                "oracle": "oracle.mkgc_forward (eval mode) -> full_table_scores / count-less-than in float64, same parameters", "oracle_seconds": oracle_s})
    return rep


def roofline_fusion(device, cpu_probe=False, reps=20):
    """SURVEY par. 8(d): the Mformer fusion through the drop-in ``MformerFusion`` operator's own entries (snag_fusion_fwd / snag_fusion_bwd, the
    FULL layer and all four outputs -- joint, joint_fz, hidden, weight_norm, model/SNAG_tools.py:23-51 -- not the read-out-only form the step
    uses) at N = 15,000, M = 3, D = 128, H = 2, one layer, no FFN, against BOTH peaks: 4,620 algorithmic bytes and 398 kFLOP per entity.
    Launch loops over prebuilt descriptors (the nn.Module wrapper's allocations are host time, not kernel time), HIP events, eval-mode forward
    and the backward of all four outputs.  ``cpu_probe``: the oracle's fusion_forward on this host next to it (BASELINE.md par. 2: 47 ms)."""
    import ctypes
    from types import SimpleNamespace
    from snag_amd import _lib, fusion as F
    from snag_amd._lib import call, ptr, stream
    N, M, D, H = 15000, 3, 128, 2
    args = SimpleNamespace(hidden_size=D, num_attention_heads=H, num_hidden_layers=1, use_intermediate=0, intermediate_size=400)
    torch.manual_seed(99)
    mod = F.MformerFusion(args, modal_num=M).to(device).eval()
    g = torch.Generator().manual_seed(98)
    toks = [torch.randn(N, D, generator=g) for _ in range(M)]
    x0 = torch.stack(toks, 1).to(device).contiguous()
    with torch.no_grad():
        outs_mod = mod([t.to(device) for t in toks])
    cfg = mod._cfg(N, M)
    layer = mod.fusion_layer[0]
    ps = [q.detach() for q in F._layer_params(layer, False)]
    lt = [[F._stacked(ps[0:3]), F._stacked(ps[3:6])] + [q.contiguous() for q in ps[6:]]]
    lib = _lib.load()
    outs = (torch.empty(N, M * D, device=device), torch.empty(N, M * D, device=device), torch.empty(N, M, D, device=device), torch.empty(N, M, device=device))
    save = torch.empty(lib.snag_fusion_save_floats(N, M, D, H, 400, 1, 0), device=device)
    work = torch.empty(lib.snag_fusion_work_floats(N, M, D, H, 400, 1, 0), device=device)
    d, hold = F._fill_desc(cfg, lt, None, mod.weight_raw.detach(), x0, outs, save, work)
    b = _lib.FusionBwdDesc()
    b.f = d
    ups = [torch.randn_like(o) for o in outs]
    b.d_joint, b.d_joint_fz, b.d_hidden, b.d_weight_norm = (ptr(u) for u in ups)
    d_x0 = torch.empty_like(x0)
    b.d_x0 = ptr(d_x0)
    garr = (_lib.FusionLayerGrad * 1)()
    gl = [torch.empty(s, device=device) for s in ((3 * D, D), (3 * D,), (D, D), (D,), (D,), (D,))]
    for nm, tns in zip(("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b"), gl):
        setattr(garr[0], nm, ptr(tns))
    b.grads = ctypes.cast(garr, ctypes.POINTER(_lib.FusionLayerGrad))
    d_raw = torch.empty(6, device=device)
    b.d_weight_raw = ptr(d_raw)

    def timed(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    fwd_us = timed(lambda: call("snag_fusion_fwd", ctypes.byref(d), stream()))
    same = all(bool(torch.equal(a, c)) for a, c in zip(outs, outs_mod))                 # the loop above IS the module's forward
    bwd_us = timed(lambda: call("snag_fusion_bwd", ctypes.byref(b), stream()))
    fwd_bytes, fwd_flops = N * (3 * M * D * 4 + M * 4), N * (8 * M * D * D + 4 * M * M * D)
    bwd_bytes, bwd_flops = N * 4 * M * D * 4, 2 * fwd_flops            # reads x0, d_hidden, d_joint (+ d_joint_fz: same bytes class), writes d_x0; two products per forward product
    peak6 = BF16_MFMA_PEAK_TFLOPS / 6.0
    rec = {"workload": f"MformerFusion drop-in (full layer, 4 outputs): N={N}, M={M}, D={D}, H={H}, 1 layer, no FFN, eval-mode forward + backward of all four outputs",
           "entries": "snag_fusion_fwd / snag_fusion_bwd over the module's own descriptor", "loop_equals_module_forward": same,
           "fwd": {"avg_us": fwd_us, "algorithmic_bytes": fwd_bytes, "GBs": fwd_bytes / fwd_us / 1e3, "frac_hbm": fwd_bytes / fwd_us / 1e3 / HBM_PEAK_GBS,
                   "algorithmic_flops": fwd_flops, "TFLOPs": fwd_flops / fwd_us / 1e6, "frac_f32_mfma_peak": fwd_flops / fwd_us / 1e6 / F32_MFMA_PEAK_TFLOPS,
                   "frac_split_bf16_bound": fwd_flops / fwd_us / 1e6 / peak6, "entities_per_s": N / fwd_us * 1e6},
           "bwd": {"avg_us": bwd_us, "algorithmic_bytes": bwd_bytes, "GBs": bwd_bytes / bwd_us / 1e3, "frac_hbm": bwd_bytes / bwd_us / 1e3 / HBM_PEAK_GBS,
                   "algorithmic_flops": bwd_flops, "TFLOPs": bwd_flops / bwd_us / 1e6, "frac_f32_mfma_peak": bwd_flops / bwd_us / 1e6 / F32_MFMA_PEAK_TFLOPS,
                   "frac_split_bf16_bound": bwd_flops / bwd_us / 1e6 / peak6},
           "per_entity": {"bytes": 3 * M * D * 4 + M * 4, "flops": 8 * M * D * D + 4 * M * M * D, "source": "SURVEY.md par. 8(d): 4,620 B and 398 kFLOP per entity"},
           "peaks": {"hbm_GBs": HBM_PEAK_GBS, "f32_mfma_TFLOPs": F32_MFMA_PEAK_TFLOPS, "split_bf16_TFLOPs": peak6},
           "timed": f"{reps} back-to-back calls of each entry, HIP events, outside the timed region; rocprofv3 per-kernel times: profiles/r05_fusion_kernel_stats.txt"}
    if cpu_probe:
        from oracle import ref_cpu as O
        sd = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
        thr = min(16, len(os.sched_getaffinity(0)))
        torch.set_num_threads(thr)
        with torch.no_grad():
            O.fusion_forward(toks, sd, H, 1, False)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                ref = O.fusion_forward(toks, sd, H, 1, False)
                ts.append(time.perf_counter() - t0)
        ts.sort()
        err = max(float((a.cpu() - c).abs().max()) for a, c in zip(outs_mod, ref))
        rec["cpu_probe"] = {"fwd_ms": ts[len(ts) // 2] * 1e3, "threads": thr, "what": "oracle.fusion_forward (eval) on this host, median of 5; BASELINE.md par. 2 probed 47 ms on 8 threads",
                            "max_abs_err_gpu_vs_oracle": err, "speedup_fwd": ts[len(ts) // 2] * 1e6 / fwd_us}
    return rec


def cpu_baseline(cfg, feats, batch, budget_s, seed):
    """The CPU oracle (plain PyTorch restatement of the reference path) on this host's cores: same step definition."""
    from oracle import ref_cpu as O
    import math
    torch.manual_seed(seed)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    N, D, R = cfg.n_ent, cfg.dim, cfg.n_rel
    g = torch.Generator().manual_seed(seed)

    def lin(o, i):
        return ((torch.rand(o, i, generator=g) * 2 - 1) / math.sqrt(i)).requires_grad_(True)

    ent = (torch.randn(N, D, generator=g) / math.sqrt(N)).requires_grad_(True)
    rel = (torch.randn(R, D, generator=g) * 0.05).requires_grad_(True)
    proj = [{"l1": lin(D, fd), "l2": lin(D, D)} for fd in cfg.feat_dims]
    fp = {"weight_raw": torch.ones(6, requires_grad=True)}
    pre = "fusion_layer.0."
    for nm in ("query", "key", "value"):
        fp[pre + f"attention.self.{nm}.weight"] = lin(D, D)
        fp[pre + f"attention.self.{nm}.bias"] = torch.zeros(D, requires_grad=True)
    fp[pre + "attention.output.dense.weight"] = lin(D, D)
    fp[pre + "attention.output.dense.bias"] = torch.zeros(D, requires_grad=True)
    fp[pre + "attention.output.LayerNorm.weight"] = torch.ones(D, requires_grad=True)
    fp[pre + "attention.output.LayerNorm.bias"] = torch.zeros(D, requires_grad=True)
    params = [ent, rel] + [w for p in proj for w in p.values()] + list(fp.values())
    undecayed = [v for k, v in fp.items() if "bias" in k or k == "weight_raw"]                  # set_optim groups (src/utils.py:47-55)
    decayed = [p for p in params if not any(p is u for u in undecayed)]
    opt = torch.optim.AdamW([{"params": decayed, "weight_decay": cfg.weight_decay}, {"params": undecayed, "weight_decay": 0.0}], lr=cfg.lr)
    M = cfg.n_tokens
    noise = torch.randn(N, D, generator=g) * 0.01
    nmask = torch.rand(N, generator=g) < cfg.noise_ratio * 0.5
    h = torch.randint(0, N, (batch,), generator=g)
    r = torch.randint(0, R, (batch,), generator=g)
    t = torch.randint(0, N, (batch,), generator=g)
    neg = torch.randint(0, N, (batch, cfg.neg_num), generator=g)

    def step():
        masks = [{"probs": torch.rand(N, cfg.num_heads, M, M) >= 0.1, "attn_out": torch.rand(N, M, D) >= 0.1}]
        toks = [O.projection_head(f, p["l1"], p["l2"]) for f, p in zip(feats, proj)]
        toks.append(O.entity_noise_blend(ent, noise, nmask, cfg.mask_ratio))
        _, _, hidden, _ = O.fusion_forward(toks, fp, cfg.num_heads, 1, False, drop_masks=masks)
        loss, _, _ = O.score_margin(hidden[:, M - 1, :], rel, h, r, t, neg, cfg.margin)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if cfg.clip > 0:
            torch.nn.utils.clip_grad_norm_(params, cfg.clip)
        opt.step()
        return float(loss.detach())

    # thread count: more is not better for this op mix (many small ops); climb 8 -> 16 -> ... while it helps
    t_end = time.perf_counter() + budget_s
    best_thr, best = None, float("inf")
    thr = min(8, avail)
    while True:
        torch.set_num_threads(thr)
        step()                                  # warm-up at this thread count (allocator, thread pool)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if dt < best:
            best_thr, best = thr, dt
        if dt > best * 1.05 or thr >= avail or time.perf_counter() + 3 * dt > t_end:
            break
        thr = min(avail, thr * 2)
    torch.set_num_threads(best_thr)
    times = [best]
    while time.perf_counter() < t_end and len(times) < 20:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    # SURVEY.md par. 8(d) also asks for the two ends of the thread range: one thread, and every usable core
    ends = {}
    # (capped at 64 threads: with all 256 hardware threads of the GPU box's two EPYC 9575F this many-small-ops step was measured at
    # 38 s -- 280x slower than at 16 threads --, which would turn the baseline leg into minutes for a data point nobody can use)
    for name, n_thr in (("one_thread", 1), ("all_cores", min(avail, 64))):
        torch.set_num_threads(n_thr)
        step()                                  # warm-up at this thread count
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        ends[name] = {"value": batch / dt, "unit": "triples/s", "cores": n_thr, "usable_threads": avail, "ms_per_step": dt * 1e3}
    torch.set_num_threads(best_thr)
    model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": batch / med, "unit": "triples/s", "cores": best_thr, "kind": "port", "cpu": model, **ends,
            "sample": f"{len(times)} steps of the same workload (N={N}, B={batch}, K={cfg.neg_num}, fwd+bwd+clip+AdamW) at {best_thr} of "
                      f"{avail} usable threads, median {med * 1e3:.0f} ms/step, torch {torch.__version__} CPU"}


def run_stress(args, world, rank, device, backend, real_stdout):
    """BASELINE.json configs[4]: |E| = 1M, dim 256, 128 negative tails, bf16 tables, B = 1024 per GPU (snag_amd.stress).  One step =
    on-device sampling -> fused bf16 scorer (scores + loss + scattered fp32 gradients) -> reduce-scatter of the table gradient to the
    row owners (N > 1) -> AdamW on the owned fp32 rows + their bf16 copy -> all-gather of the updated bf16 rows (N > 1).
    Communication and compute are reported separately (HIP events on the compute stream, which waits for every collective)."""
    import torch.distributed as dist
    from snag_amd import ops
    from snag_amd.parallel import RowShard
    from snag_amd.stress import StressConfig, StressHotPath
    cfg = StressConfig(scorer=os.environ.get("BENCH_STRESS_SCORER", "touched"))
    shard = RowShard(cfg.n_ent, rank, world) if world > 1 else None
    eng = StressHotPath(cfg, device=device, shard=shard)
    B, K, N, R, D = args.batch, cfg.neg_num, cfg.n_ent, cfg.n_rel, cfg.dim
    gi = torch.Generator().manual_seed(cfg.seed + 1 + rank)
    n_batches = 16
    zipf = args.index_dist == "zipf"
    if zipf:        # SURVEY par. 8d, config 5's "realistic" variant: entity popularity ~ 1 / rank (Zipf, exponent 1.0) for heads, tails AND negatives
        cdf = torch.cumsum(1.0 / torch.arange(1, N + 1, dtype=torch.float64), 0)
        cdf /= cdf[-1].clone()
        perm = torch.randperm(N, generator=gi)                      # popular entities scattered over the table, not its first rows

        def draw(shape):
            u = torch.rand(shape, generator=gi, dtype=torch.float64)
            return perm[torch.searchsorted(cdf, u).clamp_(max=N - 1)].to(torch.int32)
        idx = [[draw((B,)).to(device), torch.randint(0, R, (B,), generator=gi, dtype=torch.int32).to(device), draw((B,)).to(device)] for _ in range(n_batches)]
        negs = [draw((B, K)).to(device) for _ in range(n_batches)]
    else:
        idx = [[torch.randint(0, hi, (B,), generator=gi, dtype=torch.int32).to(device) for hi in (N, R, N)] for _ in range(n_batches)]
    neg = torch.empty(B, K, device=device, dtype=torch.int32)
    seed = cfg.seed * 7919 + rank
    timers = {}

    def one_step(i, tm=None):
        h, r, t = idx[i % n_batches]
        if zipf:
            eng.train_step(h, r, t, negs[i % n_batches], timers=tm)          # popularity-weighted negatives, drawn ahead (the on-device sampler is uniform)
        else:
            ops.sample_negatives(h, r, t, K, N, R, seed, tick=eng.tick, out=neg)
            eng.train_step(h, r, t, neg, timers=tm)

    gc.collect()
    gc.disable()                         # (see main(): no full collection inside the timed loop, none between the warm-up and the timed region either)
    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i, timers)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank != 0:
        return

    def span(a, b):
        v = [x.elapsed_time(y) * 1e3 for x, y in zip(timers[a], timers[b])]
        return sum(v) / len(v)

    score_us, rs_us, adam_us, ag_us = span("t0", "t1"), span("t1", "t2"), span("t2", "t3"), span("t3", "t4")
    rows = B * (K + 3)
    fwd_bytes = rows * D * 2 + B * ((K + 3) * 4 + (K + 1) * 4)            # bf16 rows read once + ids + scores (SURVEY par. 8d, s = 2)
    alg_bytes = fwd_bytes + rows * D * 4 * 2                              # + read-modify-write of the fp32 gradient rows
    ach = alg_bytes / (score_us * 1e-6) / 1e9
    shard_rows = -(-N // world)
    out = {
        "metric": "triples/sec (negative-sampled margin scoring fwd+bwd + row-owned AdamW) at |E|=1M dim=256 neg=128, bf16 tables",
        "value": B * world * args.steps / elapsed, "unit": "triples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 tables, f32 arithmetic",
        "data": "synthetic",
        "config": {"workload": "configs[4]: stress, |E|=1M x 256 bf16 entity table (512 MB, replicated), 128 negative tails, "
                               + ("Zipf(1.0) entity popularity (heads, tails, negatives), " if zipf else "uniform indices, ") +
                               "fp32 master rows + Adam moments owned 1/N per rank", "n_ent": N, "n_rel": R, "dim": D, "neg": K, "batch_per_gpu": B,
                   "global_batch": B * world, "parallelism": f"dp{world} + row-owned optimiser state", "final_loss": float(eng.loss)},
        "phases_us": {"score_fwd_bwd (+ zeroing the relation gradient)": score_us, "reduce_scatter(table grad) + all_reduce(rel grad)": rs_us,
                      "adamw(owned rows) + bf16 copy + clear touched rows": adam_us, "all_gather(updated bf16 rows)": ag_us,
                      "compute_us": score_us + adam_us, "comm_us": rs_us + ag_us,
                      "comm_bytes_per_rank": {"reduce_scatter_fp32": (world - 1) * shard_rows * D * 4, "all_gather_bf16": (world - 1) * shard_rows * D * 2}},
        "roofline": {"kernel": "touched_a_kernel + touched_b_kernel + touched_hub_kernel <bf16 tables> (snag_score_margin_fwd_bwd_touched_bf16: segmented form over the "
                               "batch's entries, no float atomics, bit-reproducible), HBM-resident 512 MB table" if cfg.scorer == "touched" else
                               "score_fused_kernel<bf16 tables> (snag_score_margin_fwd_bwd_bf16, fp32 atomics), HBM-resident 512 MB table", "bound": "hbm",
                     "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_us": score_us,
                     "algorithmic_bytes_per_launch": alg_bytes, "atomic_bytes_per_launch": rows * D * 4, "atomic_floor_us": rows * D * 4 / 1.3e12 * 1e6,
                     "timed": "HIP events (torch events on the launch stream) around the entry's three launches + the 1 MB relation-gradient fill, inside the timed region"},
    }
    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline_stress(cfg, B, min(args.cpu_seconds, 30.0))
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


def cpu_baseline_stress(cfg, batch, budget_s):
    """The oracle's scorer + torch.optim.AdamW on this host: same step (bf16-rounded tables, fp32 arithmetic, dense AdamW over 1M x 256)."""
    from oracle import ref_cpu as O
    N, R, D, K = cfg.n_ent, cfg.n_rel, cfg.dim, cfg.neg_num
    g = torch.Generator().manual_seed(cfg.seed)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    thr = min(16, avail)
    torch.set_num_threads(thr)
    em = torch.nn.Parameter(torch.randn(N, D, generator=g) * cfg.init_scale)
    rm = torch.nn.Parameter(torch.randn(R, D, generator=g) * cfg.init_scale)
    opt = torch.optim.AdamW([em, rm], lr=cfg.lr, weight_decay=cfg.weight_decay)
    h, r, t = (torch.randint(0, hi, (batch,), generator=g) for hi in (N, R, N))
    neg = torch.randint(0, N, (batch, K), generator=g)

    def step():
        e16 = em.detach().bfloat16().float().requires_grad_(True)
        r16 = rm.detach().bfloat16().float().requires_grad_(True)
        loss, _, _ = O.score_margin(e16, r16, h, r, t, neg, cfg.margin)
        loss.backward()
        em.grad, rm.grad = e16.grad, r16.grad
        opt.step()

    step()
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < 10:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": batch / med, "unit": "triples/s", "cores": thr, "kind": "port",
            "sample": f"{len(times)} steps of the same workload (|E|={N}, D={D}, K={K}, B={batch}: oracle scorer fwd+bwd + torch.optim.AdamW over the whole "
                      f"table) at {thr} of {avail} usable threads, median {med * 1e3:.0f} ms/step"}


def pool_variant(args, feats, batches, known, rank_seed, device, steps=20, warmup=40):
    """The same step with the README's POOL=1 (``use_pool``, README.md:68,79; scripts/run_base.sh default): both raw tables average-pooled to 256
    columns once at load, so the first projections are [15000, 256] x [128, 256]^T.  A second engine, ``warmup`` + ``steps`` steps, wall clock
    around the timed ones (device synchronised on both sides); the pooling launch itself is load time and is reported separately."""
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    cfg = MKGCConfig(use_pool=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dev_feats = [f.to(device) for f in feats]
    e0.record()
    eng = MKGCHotPath(cfg, dev_feats, device=device)
    e1.record()
    del dev_feats
    neg = torch.empty(args.batch, cfg.neg_num, device=device, dtype=torch.int32)

    def step(i):
        if i % STEPS_PER_EPOCH == 0:
            eng.update_noise()
        h, r, t = batches[i % len(batches)]
        eng.forward(h, r, t, neg, with_backward=True, sample=(rank_seed, known))
        eng.backward()
        eng.optimizer_step()

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return {"what": "README POOL=1: 4096- and 768-wide tables adaptively average-pooled (1-D) to 256 columns at load; same step otherwise",
            "feat_dims": list(eng.cfg.feat_dims), "pooled_from": eng.pooled_from, "ms_per_step": dt / steps * 1e3, "value": args.batch * steps / dt, "unit": "triples/s",
            "steps": steps, "warmup": warmup, "engine_setup_ms (pooling launches + allocation + upload)": e0.elapsed_time(e1), "final_loss": float(eng.loss)}


def self_launch(n_gpus):
    """One rank per GPU under torch.distributed.run (rendezvous on 127.0.0.1, a free port), same arguments; returns its exit code.  The
    child's stdout (rank 0's JSON line) is this process's stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="positives per GPU per step (NUM_BATCH)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the CPU-baseline leg (0 = skip)")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--index-dist", default="uniform", choices=["uniform", "zipf"],
                    help="--config stress: entity indices uniform (worst-case locality) or Zipf(1.0) popularity (SURVEY par. 8d)")
    ap.add_argument("--config", default="db15k", choices=["db15k", "mkgw", "stress"],
                    help="db15k = BASELINE configs[1] (Mformer_hd_graph, num_proj=2, 558 relations); mkgw = configs[2] (Mformer_hd_mean, num_proj=1, 338 relations); "
                         "stress = configs[4] (|E|=1M, dim 256, 128 negatives, bf16 tables: scorer + row-owned AdamW, comm and compute reported separately)")
    ap.add_argument("--deterministic", action="store_true", help="bit-reproducible steps (fixed-point integer atomics in the scorer's backward)")
    ap.add_argument("--graph", action="store_true", help="replay the whole step (sampling included) as one captured hipGraph (1 GPU; no per-kernel timers)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch positives PER GPU (default); strong: --batch is the GLOBAL batch, split over the GPUs")
    ap.add_argument("--shard-fusion", action="store_true",
                    help="N > 1: every rank OWNS 1/N of the entity rows (projects and fuses them only; all-gather of the read-outs, reduce-scatter of their "
                         "gradient) instead of the default plain data parallelism (full replicas + ONE summed all-reduce of the gradients per step, "
                         "SURVEY.md par. 8e: the optional optimisation, off by default)")
    ap.add_argument("--replicate-fusion", action="store_true", help="(the default since round 5; accepted for older command lines)")
    ap.add_argument("--noise-level", default="epoch", choices=["epoch", "step"],
                    help="README 'noise-level': refresh the feature / entity noise once per epoch (every 78th step; run_snag.sh) or in EVERY step")
    ap.add_argument("--pool", action="store_true",
                    help="README POOL=1 (use_pool, README.md:68,79): both raw feature tables are average-pooled to 256 columns once at load; the step "
                         "then projects 256-wide features (the default run reports this variant as the extra object `pool_variant`)")
    ap.add_argument("--joint-way", default=None, choices=["Mformer_hd_graph", "Mformer_hd_mean", "Mformer_weight", "atten_weight", "learnable_weight"],
                    help="README joint_way (default: the config's)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (forward-only rate, copy ceiling, HBM-resident score probe)")
    args = ap.parse_args()

    # `python bench.py --gpus N` typed without the launcher: start `torch.distributed.run` ourselves -- as a CHILD process, before this
    # process has touched the GPU (an exec from a GPU-initialised process takes the box down) -- forward its one JSON line and its
    # return code.  Under the launcher (WORLD_SIZE set) nothing changes.  (The reference's intent: src/distributed_utils.py:14-35.)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    # RCCL prints its version banner on STDOUT when the first communicator comes up; the contract is ONE JSON line there, so
    # everything written to fd 1 before the result goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but the launcher started {world} rank(s): pass --nproc-per-node {args.gpus}")
    from snag_amd.build import ensure_built
    ensure_built(local)                  # no-op when the prebuilt libsnag_hip.so travelled with the tree
    import torch.distributed as dist
    # BENCH_BACKEND=gloo: dry run of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices, gloo carries the
    # collectives through host buffers) -- a functional check only, used by tests/test_bench_gpu.py; the real thing is "nccl" = RCCL
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    local_dev = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    if args.config == "stress":
        run_stress(args, world, rank, device, backend, real_stdout)
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    from snag_amd.parallel import GradAllReduce, RowShard
    if args.scaling == "strong":
        if args.batch % world:
            sys.exit(f"--scaling strong: the global batch {args.batch} must divide over {world} GPUs")
        args.batch //= world                               # from here on: positives per GPU
    cfg = MKGCConfig()                                     # DB15K-shaped defaults (configs[1])
    if args.config == "mkgw":                              # configs[2]: README "MKG-W 1 0 Mformer_hd_mean 0.2 0.7 epoch 1 2"
        cfg.joint_way, cfg.num_proj, cfg.n_rel = "Mformer_hd_mean", 1, 338
    cfg.deterministic = bool(args.deterministic)
    if args.joint_way:
        cfg.joint_way = args.joint_way
    cfg.use_pool = int(args.pool)
    feats, batches, known = synth(cfg, args.batch, STEPS_PER_EPOCH, cfg.seed, device)
    from snag_amd import ops
    neg_buf = torch.empty(args.batch, cfg.neg_num, device=device, dtype=torch.int32)
    rank_seed = cfg.seed * 7919 + int(os.environ.get("RANK", "0"))
    # N > 1, default: plain data parallelism -- the batch is sharded, every rank holds full replicas and the gradients are summed by ONE bucketed
    # all-reduce per step (north_star; SURVEY.md par. 8e).  --shard-fusion: the entity rows are owned in equal shards -- each rank projects +
    # fuses 1/N of the table, the [N, D] read-outs are all-gathered before scoring and their gradient reduce-scattered back (par. 8e's option)
    if args.shard_fusion and args.replicate_fusion:
        sys.exit("--shard-fusion and --replicate-fusion exclude each other")
    row_sharded = world > 1 and args.shard_fusion
    shard = RowShard(cfg.n_ent, rank, world) if row_sharded else None
    if os.environ.get("BENCH_L1_ROUND"):                 # A/B only (tools/): K splits of the grouped first projections sized for two workgroups per CU
        MKGCHotPath.l1_round = int(os.environ["BENCH_L1_ROUND"])
    if os.environ.get("BENCH_W_ROUND"):                  # A/B only: see MKGCHotPath.w_round
        MKGCHotPath.w_round = int(os.environ["BENCH_W_ROUND"])
    eng = MKGCHotPath(cfg, feats, device=device, shard=shard)
    raw_dims, cfg = list(cfg.feat_dims), eng.cfg               # (--pool: the engine's configuration carries the pooled widths)
    reducer = GradAllReduce(world) if world > 1 else None
    if os.environ.get("BENCH_FORCE_ALLREDUCE") and reducer is None:          # exercise the collective path on a 1-GPU box
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        reducer = GradAllReduce(1, force=True)
    gscale = 1.0                                           # loss is a SUM over triples (MSNEA.py:122): sharded sum needs no division

    timers = {}
    if not args.no_kernel_timers:
        import snag_amd.mkgc as mk
        # time the two launches the roofline objects describe: projection l1 GEMM (dominant) and the score forward
        timers["score_fwd_bwd"] = KernelTimer()
        timers["proj_l1_fwd"] = KernelTimer()
        timers["proj_l1_fwd_reduce"] = KernelTimer()     # the ONE reduce launch the visual and the text projection share
        eng.probe = timers

    noise_every = 1 if args.noise_level == "step" else STEPS_PER_EPOCH
    phase_marks = []                    # per timed step with phase events: [(name, event), ...] from "start" to "clip_adamw"

    def one_step(i):
        if eng.marks is not None:
            eng._mark("start")
        if cfg.add_noise and i % noise_every == 0:
            eng.update_noise()
            eng._mark("noise_refresh")
        h, r, t = batches[i % len(batches)]
        # negatives are drawn by the step's first launch (snag_step_prologue); the scorer runs fwd+bwd as one launch
        if SPLIT_PROLOGUE:      # A/B: the sampler as its own launch
            ops.sample_negatives(h, r, t, cfg.neg_num, cfg.n_ent, cfg.n_rel, rank_seed, known_sorted=known, tick=eng.tick, out=neg_buf)
        eng.forward(h, r, t, neg_buf, with_backward=True, grad_scale=gscale, sample=None if SPLIT_PROLOGUE else (rank_seed, known))
        eng.backward(gscale, reducer=reducer)         # buckets go out as they become final (overlaps the projection backward)
        if reducer is not None:
            reducer.finish()
            eng._mark("allreduce_wait")
        eng.optimizer_step()
        eng._mark("clip_adamw")

    if args.graph:
        if world > 1:
            sys.exit("--graph is single-process (collectives are not captured)")
        eng.probe = None
        timers = {}
        if cfg.add_noise:
            eng.update_noise()
        eng.capture_step(args.batch, cfg.neg_num, sample=(rank_seed, known))

        def one_step(i):                                  # noqa: F811  (graph replay variant)
            if cfg.add_noise and i % noise_every == 0 and i > 0:
                eng.update_noise()
            eng.replay_step(*batches[i % len(batches)])

    copy_gbs = device_copy_ceiling(device)
    extras = {}
    sim_inputs = None
    if world == 1 and not args.no_extras and args.config == "db15k":
        extras["roofline_sim"], *sim_inputs = roofline_sim(device)
        extras["roofline_fusion"] = roofline_fusion(device, cpu_probe=args.cpu_seconds > 0)
    if cfg.add_noise:
        extras["noise_refresh_us"] = noise_refresh_us(eng)
    # the host enqueues a step in ~140 us of Python; a full garbage collection in the middle of the loop (tens of ms with torch's heap)
    # would drain the launch queue and show up as device time -- collect now, keep the collector off while the K steps are timed.
    # (BEFORE the forward-only leg and the warm-up steps: a collection between the warm-up and the timed region idles the device for
    # tens of ms, and the timed steps then run through its power-management transient -- 0.505-0.516 ms per step at --steps 20 --warmup 5
    # against 0.44 when nothing idles the device in between; tools/probe_ramp.py)
    gc.collect()
    gc.disable()
    # LAST before the warm-up steps (the legs above hold CPU work -- the device would idle and drop its clocks again): the forward-only rate
    if not args.no_extras:
        extras.update(untimed_extras(eng, batches, args, device, sample=None if row_sharded else (rank_seed, known)))          # (a row-sharded backward holds collectives: forward-only there)
    step_no = 0
    for _ in range(args.warmup):
        one_step(step_no)
        step_no += 1
    for tm in timers.values():
        tm.pairs.clear()
    torch.cuda.synchronize()
    want_phases = (world > 1 or bool(os.environ.get("BENCH_PHASES"))) and not args.graph
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        # the per-launch HIP events sit INSIDE the timed region, on every TIMER_EVERY-th step only: an event pair around a launch
        # costs the step 5-10 us of stream bubbles (measured: 0.786-0.792 ms with events on every step vs 0.763 ms without)
        if timers:
            eng.probe = timers if k % TIMER_EVERY == TIMER_EVERY - 1 else None
        # phase events (N > 1, or BENCH_PHASES=1): on the timed steps that carry no per-launch timers, every TIMER_EVERY-th too
        if want_phases and k % TIMER_EVERY == 1:
            eng.marks = []
            phase_marks.append(eng.marks)
        one_step(step_no)
        eng.marks = None
        step_no += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss = float(eng.loss)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        out = {
            "metric": "triples/sec (fusion+score) at |E|=15k dim=128 neg=32", "value": value, "unit": "triples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: DB15K-shaped" if args.config == "db15k" else "configs[2]: MKG-W-shaped") +
                                   f" MKGC train step (proj num_proj={cfg.num_proj} over {raw_dims[0]}/{raw_dims[1]}-wide features"
                                   + (f" average-pooled to {cfg.feat_dims[0]}/{cfg.feat_dims[1]} at load (README POOL=1), " if eng.pooled_from else ", ") +
                                   f"noise 0.2/0.7, Mformer 1-layer/2-head over all 15000 entities, {cfg.joint_way}, "
                                   "margin 12, fwd+bwd+clip_grad_norm+AdamW" + ("+RCCL all-reduce" if world > 1 else "") + ")",
                       "n_ent": cfg.n_ent, "n_rel": cfg.n_rel, "dim": cfg.dim, "neg": cfg.neg_num, "batch_per_gpu": args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "final_loss": loss,
                       # what each rank computes per step: its batch shard's sampling + scoring always; projections + Mformer over
                       # ALL entities when replicated, over its 1/N of the rows when row-sharded; clip + AdamW on the full flat buffer
                       "fusion": "single rank" if world == 1 else ("row-sharded: each rank owns 1/N of the entity rows (all-gather of read-outs, "
                                 "reduce-scatter of their gradient)" if row_sharded else "replicated on every rank"),
                       "replicated_fraction": None if world == 1 else (REPLICATED_FRACTION_SHARDED if row_sharded else REPLICATED_FRACTION_REPLICATED)},
        }
        if phase_marks:
            out["phases_us"] = phases_summary(phase_marks, world, row_sharded, cfg, eng)
        if dist.is_initialized():          # what carried the collectives: "nccl" = RCCL over xGMI, one rank per GPU
            out["config"]["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
        out["config"]["noise_level"] = args.noise_level
        out["measured_copy_GBs"] = copy_gbs
        out.update(extras)
        if timers:
            out["roofline"] = roofline_dominant(cfg, timers["proj_l1_fwd"], rows=eng.shard.n_loc, reduce_timer=timers["proj_l1_fwd_reduce"], grouped=eng.l1_grouped)
            out["roofline_score"] = roofline_score(cfg, args.batch, timers["score_fwd_bwd"], ln_fused=bool(getattr(eng, "_ln_fuse", False)),
                                                   ln_dropout=bool(eng.training and cfg.dropout > 0))
            if not args.no_extras:
                out["roofline_score"]["batch_8192"] = score_train_larger_batch(eng, cfg, device)
                out["roofline_score"]["hbm_resident"] = score_hbm_resident(device)
        if world == 1 and args.cpu_seconds > 0:
            host_feats = [f.cpu() for f in eng.feats]           # (the clean tables the engine holds: pooled under --pool)
            out["cpu_baseline"] = cpu_baseline(cfg, host_feats, args.batch, args.cpu_seconds, cfg.seed)
            # the metric's second half, on the engine that just ran the timed steps
            out["hits_parity"] = hits_parity(eng, host_feats, known, device)
            if sim_inputs is not None:
                out["roofline_sim"]["oracle_check"] = cpu_baseline_sim_check(*sim_inputs, device)
        if world == 1 and not args.no_extras and args.config == "db15k" and not args.pool and not args.graph:
            out["pool_variant"] = pool_variant(args, feats, batches, known, rank_seed, device)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
