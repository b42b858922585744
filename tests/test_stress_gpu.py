"""GPU parity for BASELINE configs[4] (the 1M x 256 x 128-negatives bf16 stress path, snag_amd.stress) at sizes the oracle finishes in
seconds: the bf16-table scorer's training entries vs the CPU oracle on bf16-rounded tables (fp32 arithmetic on both sides), the
whole optimiser step vs torch.optim.AdamW, the row-sharded two-process form vs the single-process one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import assert_close
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(N, R, D, B, K, seed, idt=torch.int32):
    g = torch.Generator().manual_seed(seed)
    ent = (torch.randn(N, D, generator=g) * 0.5).bfloat16()
    rel = (torch.randn(R, D, generator=g) * 0.3).bfloat16()
    idx = [torch.randint(0, hi, sz, generator=g).to(idt) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
    return ent, rel, idx


def _bits(x):
    return x.view(torch.int16) if x.dtype == torch.bfloat16 else x


@pytest.mark.parametrize("D,K,B,idt", [(256, 128, 48, torch.int32), (256, 32, 257, torch.int64), (128, 32, 100, torch.int32), (64, 3, 50, torch.int32),
                                        (512, 9, 12, torch.int32), (8, 2, 9, torch.int32), (256, 130, 9, torch.int64), (104, 5, 21, torch.int32)])
@pytest.mark.parametrize("margin", [1.2, 12.0])
def test_bf16_scorer_training_entries_vs_oracle(D, K, B, idt, margin):
    from snag_amd._lib import call, ptr, stream
    from snag_amd import ops, _lib
    N, R = 501, 13
    ent, rel, idx = _mk(N, R, D, B, K, seed=D * 100 + K, idt=idt)
    e_ = ent.float().requires_grad_(True)            # the oracle computes in fp32 on the bf16-rounded values, as the kernel does
    r_ = rel.float().requires_grad_(True)
    loss, pos, ng = O.score_margin(e_, r_, *idx, margin)
    loss.backward()
    ed, rd = ent.to(DEV), rel.to(DEV)
    d = [x.to(DEV) for x in idx]
    ib = 4 if idt == torch.int32 else 8
    pos_g, neg_g, lp, lg = torch.empty(B, device=DEV), torch.empty(B, K, device=DEV), torch.empty(B, device=DEV), torch.empty((), device=DEV)
    de, dr = torch.zeros(N, D, device=DEV), torch.zeros(R, D, device=DEV)
    tick = torch.zeros(_lib.SCORE_TICKET_WORDS, device=DEV, dtype=torch.int32)
    call("snag_score_margin_fwd_bwd_bf16", ptr(ed), D, N, ptr(rd), R, D, *[ptr(x) for x in d], ib, B, K, margin, 1.0, ptr(pos_g), ptr(neg_g), ptr(lp), ptr(lg),
         ptr(tick), ptr(de), D, ptr(dr), stream())
    assert_close(pos_g, pos, atol=1e-5, what="pos")
    assert_close(neg_g, ng, atol=1e-5, what="neg")
    assert_close(lg, loss, atol=1e-4, rtol=1e-5, what="loss")
    assert_close(de, e_.grad, atol=2e-4, rtol=1e-4, what="d_ent")
    assert_close(dr, r_.grad, atol=2e-4, rtol=1e-4, what="d_rel")
    # the forward-only bf16 kernel gives the same scores; the two-launch backward the same gradients
    pf, nf, _ = ops.score_margin_fwd_bf16(ed, rd, *d, margin)
    assert_close(pf, pos_g, atol=3e-6)
    assert_close(nf, neg_g, atol=3e-6)
    de2, dr2 = torch.zeros(N, D, device=DEV), torch.zeros(R, D, device=DEV)
    call("snag_score_margin_bwd_bf16", ptr(ed), D, N, ptr(rd), R, D, *[ptr(x) for x in d], ib, B, K, margin, 1.0, None, None, ptr(de2), D, ptr(dr2), stream())
    assert_close(de2, e_.grad, atol=2e-4, rtol=1e-4)
    # fixed-point (bit-reproducible) scatter: two runs identical bits, values equal to rounding
    outs = []
    for _ in range(2):
        ae, ar = torch.zeros(N, D, device=DEV, dtype=torch.int64), torch.zeros(R, D, device=DEV, dtype=torch.int64)
        call("snag_score_margin_bwd_fixed_bf16", ptr(ed), D, N, ptr(rd), R, D, *[ptr(x) for x in d], ib, B, K, margin, 1.0, None, None, ptr(ae), ptr(ar), stream())
        outs.append((ae.clone(), ar.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert_close(outs[0][0].double() / 4294967296.0, e_.grad, atol=2e-4, rtol=1e-4)


def test_adamw_bf16_and_zero_touched_rows():
    from snag_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(1)
    n = 4096 * 3 + 8
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-2, weight_decay=1e-2)
    pd, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    out16 = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    tick = torch.zeros(1, device=DEV, dtype=torch.int64)
    for step in range(3):
        p.grad = gr * (step + 1)
        opt.step()
        call("snag_adamw_step_bf16", ptr(pd), ptr((gr * (step + 1)).to(DEV)), ptr(m), ptr(v), n, 1e-2, 0.9, 0.999, 1e-8, 1e-2, 0, ptr(tick), 1.0, ptr(out16), stream())
        call("snag_tick", ptr(tick), stream())
        assert_close(pd, p.detach(), atol=2e-6, rtol=2e-6, what=f"adamw step {step}")
        assert torch.equal(_bits(out16.cpu()), _bits(pd.cpu().bfloat16()))          # the bf16 copy is the RNE rounding of the new weights
    # clear exactly the touched rows
    N, D, B, K = 300, 64, 20, 7
    buf = torch.ones(N, D, device=DEV)
    h, t, neg = (torch.randint(0, N, s, generator=g, dtype=torch.int32).to(DEV) for s in ((B,), (B,), (B, K)))
    call("snag_zero_touched_rows", ptr(buf), D, D, ptr(h), ptr(t), ptr(neg), 4, B, K, stream())
    touched = torch.zeros(N, dtype=torch.bool)
    touched[torch.cat([h.cpu(), t.cpu(), neg.cpu().flatten()]).long()] = True
    assert torch.equal(buf.cpu().sum(1) == 0, touched) and float(buf[~touched.to(DEV)].min()) == 1.0


def _adam_close(got, ref, lr, steps, what):
    """Adam's update is lr * m / (sqrt(v) + eps): where a gradient entry is ~eps in size (an entity drawn once with a nearly
    orthogonal partner) its last-bit noise decides the sign of a full +-lr step.  Everything else must agree to fp32 rounding; the
    handful of ill-conditioned entries may differ by at most the steps they could have taken."""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs()
    bad = err > 3e-5 + 1e-5 * ref.abs()
    assert int(bad.sum()) <= max(3, int(2e-5 * bad.numel())), f"{what}: {int(bad.sum())} of {bad.numel()} entries differ (max {float(err.max()):.2e})"
    assert float(err.max()) <= 2.0 * lr * steps, f"{what}: max abs err {float(err.max()):.2e}"


def _cfg(N=3000, deterministic=False):
    from snag_amd.stress import StressConfig
    return StressConfig(n_ent=N, n_rel=11, dim=256, neg_num=128, margin=12.0, lr=1e-2, weight_decay=1e-2, deterministic=deterministic, seed=3)


def _inits(N):
    g = torch.Generator().manual_seed(9)
    return torch.randn(N, 256, generator=g) * 0.05, torch.randn(11, 256, generator=g) * 0.05


def _batches(N, n=3, B=64):
    g = torch.Generator().manual_seed(5)
    return [[torch.randint(0, hi, sz, generator=g, dtype=torch.int32) for hi, sz in ((N, (B,)), (11, (B,)), (N, (B,)), (N, (B, 128)))] for _ in range(n)]


@pytest.mark.parametrize("deterministic", [False, True])
def test_stress_step_vs_oracle(deterministic):
    """Three optimiser steps of the config-5 path at |E| = 3000: bf16 tables -> fp32 scores / loss / gradients -> AdamW on the fp32 master
    rows -> bf16 copy, against the oracle scorer + torch.optim.AdamW fed with the same bf16-rounded tables."""
    from snag_amd.stress import StressHotPath
    N = 3000
    ent0, rel0 = _inits(N)
    eng = StressHotPath(_cfg(N, deterministic), device=DEV, ent_init=ent0.to(DEV), rel_init=rel0.to(DEV))
    em, rm = torch.nn.Parameter(ent0.clone()), torch.nn.Parameter(rel0.clone())
    opt = torch.optim.AdamW([em, rm], lr=1e-2, weight_decay=1e-2)
    for step, batch in enumerate(_batches(N)):
        e16 = em.detach().bfloat16().float().requires_grad_(True)          # the scorer sees the bf16 copy; the gradient goes to the master
        r16 = rm.detach().bfloat16().float().requires_grad_(True)
        loss, _, _ = O.score_margin(e16, r16, *batch, 12.0)
        loss.backward()
        em.grad, rm.grad = e16.grad, r16.grad
        opt.step()
        lg = eng.train_step(*[x.to(DEV) for x in batch])
        assert_close(lg, loss.detach(), atol=1e-2, rtol=2e-6, what="loss")
        _adam_close(eng.master_table(), em.detach(), 1e-2, step + 1, "master rows")
        _adam_close(eng.rel_master, rm.detach(), 1e-2, step + 1, "relation master")
        # the replicated bf16 table is the rounding of the master (one bf16 ulp where the fp32 values straddle a rounding boundary)
        assert torch.equal(_bits(eng.ent16[:N].cpu()), _bits(eng.master_table().cpu().bfloat16()))
        assert float(eng.d_ent_full.abs().max()) == 0.0 or deterministic             # scatter target cleared behind the step
    if deterministic:
        eng2 = StressHotPath(_cfg(N, True), device=DEV, ent_init=ent0.to(DEV), rel_init=rel0.to(DEV))
        for batch in _batches(N):
            eng2.train_step(*[x.to(DEV) for x in batch])
        assert torch.equal(eng2.ent_master, eng.ent_master) and torch.equal(eng2.ent16, eng.ent16)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snag_amd.parallel import RowShard, shard_batch
        from snag_amd.stress import StressHotPath
        ent0, rel0 = _inits(N)
        sh = RowShard(N, rank, world)
        eng = StressHotPath(_cfg(N, True), device=DEV, shard=sh, ent_init=ent0.to(DEV), rel_init=rel0.to(DEV))
        assert eng.ent_master.shape[0] == sh.n_shard and eng.ent_m.shape[0] == sh.n_shard          # 1/world of weights and optimiser state
        losses = []
        for batch in _batches(N):
            local = shard_batch([x.to(DEV) for x in batch], rank, world)
            lt = eng.train_step(*local).detach().double().cpu().reshape(1)
            dist.all_reduce(lt)
            losses.append(float(lt))
        torch.cuda.synchronize()
        torch.save({"ent16": eng.ent16.cpu(), "rows": eng.master_table().cpu(), "lo": sh.lo, "losses": losses, "rel": eng.rel_master.cpu()}, f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N", [3000, 3001])
def test_two_rank_row_sharded_stress_step(tmp_path, N):
    """Two ranks (sharing the one device, gloo as transport): batch sharded, table gradient reduce-scattered to the row owners, AdamW on
    1/2 of the rows and moments each, updated bf16 rows all-gathered -- equal to the single-process run on the global batch
    (bit-reproducible mode on both sides: sums over ranks of fixed-point sums are order-free up to the final fp32 conversion)."""
    from snag_amd.stress import StressHotPath
    out = str(tmp_path / "st")
    mp.spawn(_worker, args=(2, _free_port(), N, out), nprocs=2, join=True)
    got = [torch.load(f"{out}.{r}") for r in range(2)]
    ent0, rel0 = _inits(N)
    eng = StressHotPath(_cfg(N, True), device=DEV, ent_init=ent0.to(DEV), rel_init=rel0.to(DEV))
    losses = [float(eng.train_step(*[x.to(DEV) for x in b])) for b in _batches(N)]
    for a, b in zip(got[0]["losses"], losses):
        assert abs(a - b) <= 1e-5 * abs(b)
    rows = torch.cat([g["rows"] for g in got], 0)
    assert rows.shape[0] == N
    assert_close(rows, eng.master_table(), atol=2e-6, rtol=1e-5, what="owned master rows")
    assert torch.equal(_bits(got[0]["ent16"]), _bits(got[1]["ent16"]))                        # every rank holds the same replicated table
    same = (_bits(got[0]["ent16"][:N]) == _bits(eng.ent16[:N].cpu())).float().mean()
    assert float(same) > 0.9999                                                                  # bf16 rounding of fp32 values equal to ~1e-6
    assert_close(got[0]["rel"], eng.rel_master, atol=2e-6, rtol=1e-5)


def test_stress_full_size_properties():
    """BASELINE configs[4] at its full size (|E| = 1M, dim 256, 128 negative tails, B = 1024) through size-independent properties: scores
    do not change when a table is scaled (rows are L2-normalised; a power of two is exact in bf16), a negative equal to the positive tail
    scores like it, repeated steps on one batch lower its loss, rows no triple touched move by the decoupled weight decay alone (their
    Adam moments stay zero), the bf16 table is the rounding of the master rows, and the scatter target is clean again after the step."""
    from snag_amd.stress import StressConfig, StressHotPath
    from snag_amd._lib import call, ptr, stream
    cfg = StressConfig(lr=1e-3, weight_decay=1e-2)
    eng = StressHotPath(cfg, device=DEV)
    N, R, K, B = cfg.n_ent, cfg.n_rel, cfg.neg_num, 1024
    g = torch.Generator().manual_seed(12)
    h, r, t = (torch.randint(0, hi, (B,), generator=g, dtype=torch.int32).to(DEV) for hi in (N, R, N))
    neg = torch.randint(0, N, (B, K), generator=g, dtype=torch.int32).to(DEV)
    neg[:, 7] = t
    pos, ng, loss0 = eng.score(h, r, t, neg)
    pos, ng, loss0 = pos.clone(), ng.clone(), float(loss0)
    assert torch.equal(ng[:, 7], pos) and float(pos.min()) >= 0.0 and float(ng.max()) <= 9.0 + 1e-4
    saved = eng.ent16.clone()
    eng.ent16.mul_(4.0)                                                    # exact in bf16
    pos2, ng2, _ = eng.score(h, r, t, neg)
    assert torch.equal(pos2, pos) and torch.equal(ng2, ng)
    eng.ent16.copy_(saved)
    master0 = eng.ent_master.clone()
    touched = torch.zeros(N, dtype=torch.bool, device=DEV)
    touched[torch.cat([h, t, neg.flatten()]).long()] = True
    losses = [float(eng.train_step(h, r, t, neg)) for _ in range(3)]
    torch.cuda.synchronize()
    assert abs(losses[0] - loss0) <= 1e-5 * loss0 and losses[2] < losses[0]
    assert float(eng.d_ent_full.abs().max()) == 0.0
    quiet = (~touched).nonzero().flatten()[:200_000]
    decay = (1.0 - cfg.lr * cfg.weight_decay) ** 3
    assert_close(eng.ent_master[quiet], master0[quiet] * decay, atol=1e-7, rtol=1e-6, what="untouched rows: weight decay only")
    assert float((eng.ent_master[touched] - master0[touched] * decay).abs().max()) > 1e-5           # the touched ones did learn
    assert torch.equal(_bits(eng.ent16[:N][quiet].cpu()), _bits(eng.ent_master[quiet].cpu().bfloat16()))
