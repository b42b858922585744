"""CPU (gloo, world_size 2): the data-parallel exchange -- batch sharding + one summed all-reduce of the flat
gradient buffer -- reproduces the single-process gradient.  The oracle stands in for the HIP kernels here
(no GPU in this container); the N>1 GPU path differs only in the backend string ("nccl" = RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as O
from snag_amd.parallel import GradAllReduce, shard_batch, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=0):
    g = torch.Generator().manual_seed(seed)
    N, R, D, B, K = 60, 5, 16, 22, 4
    ent = torch.randn(N, D, generator=g)
    rel = torch.randn(R, D, generator=g)
    batch = (torch.randint(0, N, (B,), generator=g), torch.randint(0, R, (B,), generator=g),
             torch.randint(0, N, (B,), generator=g), torch.randint(0, N, (B, K), generator=g))
    return ent, rel, batch


def _flat_grad(ent, rel, batch):
    e, r = ent.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    loss, _, _ = O.score_margin(e, r, *batch, 2.0)
    loss.backward()
    return torch.cat([e.grad.flatten(), r.grad.flatten()]), float(loss)


def _worker(rank, world, port, bucket_bytes, overlapped, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        ent, rel, batch = _make()
        local = shard_batch(batch, rank, world)
        flat, loss = _flat_grad(ent, rel, local)
        red = GradAllReduce(world, bucket_bytes=bucket_bytes)
        if overlapped:                       # the two-bucket start()/finish() form the training step uses
            cut = flat.numel() // 3
            red.start(flat[:cut])
            red.start(flat[cut:])
            red.finish()
        else:
            red(flat)
        lt = torch.tensor([loss], dtype=torch.float64)
        dist.all_reduce(lt)
        if rank == 0:
            torch.save({"flat": flat, "loss": float(lt)}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,overlapped", [(1 << 20, False), (256, False), (1 << 20, True), (512, True)])
def test_two_rank_allreduce_matches_single_process(tmp_path, bucket_bytes, overlapped):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), bucket_bytes, overlapped, out), nprocs=2, join=True)
    got = torch.load(out)
    ent, rel, batch = _make()
    ref, ref_loss = _flat_grad(ent, rel, batch)
    # the margin loss is a SUM over triples (MSNEA.py:122): summed shard gradients == full-batch gradient, no division
    assert torch.allclose(got["flat"], ref, atol=1e-5, rtol=1e-5)
    assert abs(got["loss"] - ref_loss) < 1e-3 * max(1.0, abs(ref_loss))


def _steps_worker(rank, world, port, n_total, bs, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snag_amd.parallel import common_step_count, step_bounds
        lo, hi = shard_range(n_total, rank, world)
        steps = common_step_count(hi - lo, bs)
        bounds = step_bounds(hi - lo, steps)
        # every rank enters exactly `steps` all-reduces: a mismatch would hang here, not fail
        for a, b in bounds:
            t = torch.tensor([float(b - a)])
            dist.all_reduce(t)
        torch.save({"steps": steps, "bounds": bounds, "n": hi - lo}, f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,bs", [(2049, 1024), (2048, 1024), (5, 4), (3, 1024)])
def test_ranks_agree_on_steps_per_epoch(tmp_path, n_total, bs):
    """Shards differ by one item (1025 vs 1024 at bs=1024): a per-rank range(0, n_local, bs) would issue 2 vs 1 bucketed
    all-reduces and hang.  Every rank must run the same number of steps and still cover each of its items exactly once."""
    out = str(tmp_path / "steps")
    mp.spawn(_steps_worker, args=(2, _free_port(), n_total, bs, out), nprocs=2, join=True)
    got = [torch.load(f"{out}.{r}") for r in range(2)]
    assert got[0]["steps"] == got[1]["steps"] == -(-max(g["n"] for g in got) // bs)
    for g in got:
        b = g["bounds"]
        assert len(b) == g["steps"] and b[0][0] == 0 and b[-1][1] == g["n"]
        assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert max(hi - lo for lo, hi in b) <= bs


def _rowshard_worker(rank, world, port, n_rows, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snag_amd.parallel import RowShard
        torch.set_num_threads(1)
        g = torch.Generator().manual_seed(3)
        N, R, D, B, K = n_rows, 5, 16, 24, 4
        table = torch.randn(N, D, generator=g)                    # stands for the entity-side input of the per-row fusion
        w = torch.randn(D, D, generator=g).requires_grad_(True)   # stands for the dense (replicated) fusion parameters
        rel = torch.randn(R, D, generator=g).requires_grad_(True)
        batch = (torch.randint(0, N, (B,), generator=g), torch.randint(0, R, (B,), generator=g),
                 torch.randint(0, N, (B,), generator=g), torch.randint(0, N, (B, K), generator=g))
        sh = RowShard(N, rank, world)
        assert sh.n_pad >= N and sh.n_pad - N < world and 0 <= sh.n_loc <= sh.n_shard
        # this rank "fuses" ITS rows only: a per-row function of the table and the dense parameters
        mine = table[sh.lo:sh.lo + sh.n_loc].clone().requires_grad_(True)
        local = torch.zeros(sh.n_shard, D)
        local[:sh.n_loc] = torch.tanh(mine @ w)
        full = torch.zeros(sh.n_pad, D)
        sh.all_gather(full, local.detach())
        # score this rank's share of the batch against the WHOLE gathered table; its gradient lands on all rows
        shard_b = shard_batch(batch, rank, world)
        tab = full.clone().requires_grad_(True)
        loss, _, _ = O.score_margin(tab[:N], rel, *shard_b, 2.0)
        loss.backward()
        d_full = torch.zeros(sh.n_pad, D)
        d_full[:N] = tab.grad[:N]
        d_local = torch.zeros(sh.n_shard, D)
        sh.reduce_scatter(d_local, d_full)
        # owners back-propagate the summed gradient of their rows through their part of the fusion
        local.backward(d_local)
        gw, grel = w.grad.clone(), rel.grad.clone()
        dist.all_reduce(gw)                                       # dense parameters: summed partial gradients
        dist.all_reduce(grel)
        d_table = torch.zeros(sh.n_pad, D)                        # table gradient: every rank holds its own rows -> gathered
        d_rows = torch.zeros(sh.n_shard, D)
        d_rows[:sh.n_loc] = mine.grad
        sh.all_gather(d_table, d_rows)
        lt = torch.tensor([float(loss)], dtype=torch.float64)
        dist.all_reduce(lt)
        if rank == 0:
            torch.save({"gw": gw, "grel": grel, "d_table": d_table[:N], "loss": float(lt), "full": full[:N]}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [60, 61])
def test_row_sharded_fusion_algebra(tmp_path, n_rows):
    """Row ownership (SURVEY par. 8e "row-sharded N/G + all-gather of [N,D] read-outs"): fuse own rows -> all-gather -> score the local
    batch shard on the whole table -> reduce-scatter the table gradient -> own rows' backward; dense gradients all-reduced.  Equals the
    single-process computation, also when N does not divide over the ranks (padded shards)."""
    out = str(tmp_path / "rs.pt")
    mp.spawn(_rowshard_worker, args=(2, _free_port(), n_rows, out), nprocs=2, join=True)
    got = torch.load(out)
    g = torch.Generator().manual_seed(3)
    N, R, D, B, K = n_rows, 5, 16, 24, 4
    table = torch.randn(N, D, generator=g).requires_grad_(True)
    w = torch.randn(D, D, generator=g).requires_grad_(True)
    rel = torch.randn(R, D, generator=g).requires_grad_(True)
    batch = (torch.randint(0, N, (B,), generator=g), torch.randint(0, R, (B,), generator=g),
             torch.randint(0, N, (B,), generator=g), torch.randint(0, N, (B, K), generator=g))
    full = torch.tanh(table @ w)
    loss, _, _ = O.score_margin(full, rel, *batch, 2.0)
    loss.backward()
    assert torch.allclose(got["full"], full.detach(), atol=1e-6)
    assert abs(got["loss"] - float(loss)) < 1e-4 * max(1.0, abs(float(loss)))
    for a, b in ((got["gw"], w.grad), (got["grel"], rel.grad), (got["d_table"], table.grad)):
        assert torch.allclose(a, b, atol=1e-4, rtol=1e-4)


def _zero_worker(rank, world, port, n_rows, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snag_amd.parallel import RowShard
        torch.set_num_threads(1)
        g = torch.Generator().manual_seed(8)
        D = 8
        table = torch.randn(n_rows, D, generator=g)
        sh = RowShard(n_rows, rank, world)
        full16 = torch.zeros(sh.n_pad, D, dtype=torch.bfloat16)
        full16[:n_rows] = table.bfloat16()
        master = torch.zeros(sh.n_shard, D)
        master[:sh.n_loc] = table[sh.lo:sh.lo + sh.n_loc]
        p = torch.nn.Parameter(master.clone())
        opt = torch.optim.AdamW([p], lr=1e-2, weight_decay=1e-2)
        for step in range(3):
            gl = torch.Generator().manual_seed(100 * step + rank)
            d_full = torch.zeros(sh.n_pad, D)
            d_full[:n_rows] = torch.randn(n_rows, D, generator=gl)        # this rank's scatter of table gradients
            d_mine = torch.zeros(sh.n_shard, D)
            sh.reduce_scatter(d_mine, d_full)
            p.grad = d_mine
            opt.step()
            full16[sh.lo:sh.lo + sh.n_shard] = p.detach().bfloat16()
            sh.all_gather(full16, full16[sh.lo:sh.lo + sh.n_shard].clone())
        torch.save({"full16": full16[:n_rows].float(), "rows": p.detach()[:sh.n_loc].clone(), "lo": sh.lo}, f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [40, 41])
def test_row_owned_optimizer_state_equals_dense_adamw(tmp_path, n_rows):
    """Config-5 partition logic (SURVEY par. 8e option 2): reduce-scatter of the table gradient to the row owners, AdamW on the owned rows
    with moments sharded the same way, all-gather of the updated bf16 rows == dense AdamW on the summed gradient of the whole table."""
    out = str(tmp_path / "z")
    mp.spawn(_zero_worker, args=(2, _free_port(), n_rows, out), nprocs=2, join=True)
    got = [torch.load(f"{out}.{r}") for r in range(2)]
    g = torch.Generator().manual_seed(8)
    table = torch.randn(n_rows, 8, generator=g)
    p = torch.nn.Parameter(table.clone())
    opt = torch.optim.AdamW([p], lr=1e-2, weight_decay=1e-2)
    for step in range(3):
        p.grad = sum(torch.randn(n_rows, 8, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(2))
        opt.step()
    assert torch.allclose(torch.cat([x["rows"] for x in got], 0), p.detach(), atol=1e-6)
    assert torch.equal(got[0]["full16"], got[1]["full16"]) and torch.equal(got[0]["full16"], p.detach().bfloat16().float())
