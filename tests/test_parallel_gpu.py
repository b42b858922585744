"""GPU, two processes sharing the one device, gloo as the transport: the REAL engine's data-parallel step (batch sharded across ranks,
gradient buckets exchanged as the backward finishes them, AdamW on the summed gradient) equals the single-process step on the global
batch.  On an 8-GPU node the only difference is the backend string ("nccl" = RCCL over xGMI) and one device per rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _engine(joint_way, shard=None, n_ent=900, dropout=0.0, add_noise=0, feat_dims=(160, 96)):
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    cfg = MKGCConfig(n_ent=n_ent, n_rel=13, dim=128, feat_dims=list(feat_dims), num_proj=2, joint_way=joint_way, margin=6.0, neg_num=8, dropout=dropout,
                     add_noise=add_noise, seed=5, lr=1e-3, proj_dropout=0.2 if dropout > 0 else 0.0)
    g = torch.Generator().manual_seed(17)
    feats = [torch.rand(n_ent, feat_dims[0], generator=g), torch.randn(n_ent, feat_dims[1], generator=g)]
    eng = MKGCHotPath(cfg, feats, device="cuda", shard=shard)
    eng.p["ent"].copy_(torch.randn(n_ent, 128, generator=g).cuda())
    B = 64
    batch = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).cuda() for hi, sz in ((n_ent, (B,)), (13, (B,)), (n_ent, (B,)), (n_ent, (B, 8)))]
    return eng, batch


def _named(eng):
    """parameters / gradients by name: a row-sharded engine pads the entity region of its flat buffers to whole shards, so flat offsets
    differ between the 2-rank and the 1-rank engine"""
    return ({k: v.detach().cpu().clone() for k, v in eng.p.items()}, {k: v.detach().cpu().clone() for k, v in eng.g.items()})


def _flat(d):
    return torch.cat([d[k].reshape(-1) for k in sorted(d)])


def _worker(rank, world, port, joint_way, out, row_sharded=False, n_ent=900, steps=1, callable_reducer=False, dropout=0.0, add_noise=0, feat_dims=(160, 96)):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snag_amd.parallel import GradAllReduce, RowShard, shard_batch
        eng, batch = _engine(joint_way, RowShard(n_ent, rank, world) if row_sharded else None, n_ent, dropout, add_noise, feat_dims)
        local = shard_batch(batch, rank, world)
        red = GradAllReduce(world)
        if callable_reducer:                     # the plain-callable form: allreduce(gflat) after the whole backward

            def red(flat, _r=GradAllReduce(world)):
                host = flat.cpu()
                dist.all_reduce(host)
                flat.copy_(host)
        if add_noise:
            eng.update_noise()
        losses = []
        for _ in range(steps):
            loss = eng.train_step(*local, allreduce=red)
            torch.cuda.synchronize()
            lt = loss.detach().double().cpu().reshape(1)
            dist.all_reduce(lt)
            losses.append(float(lt))
        if rank == 0:
            p, g = _named(eng)
            torch.save({"grad": _flat(g), "param": _flat(p), "loss": losses[0], "losses": losses}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("joint_way,row_sharded,n_ent", [("Mformer_hd_graph", False, 900), ("Mformer_hd_mean", False, 900),
                                                         ("Mformer_hd_graph", True, 900), ("Mformer_hd_mean", True, 900),
                                                         ("Mformer_hd_graph", True, 901),
                                                         ("atten_weight", False, 900), ("atten_weight", True, 901)])
def test_two_rank_engine_step_equals_global_batch_step(tmp_path, joint_way, row_sharded, n_ent):
    """row_sharded: each rank projects and fuses ITS half of the entity rows only (the read-out table is all-gathered before scoring,
    its gradient reduce-scattered back, the entity-table gradient gathered); 901 rows: padded shards, the in-place gather of the
    entity gradient falls back to the summed all-reduce."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), joint_way, out, row_sharded, n_ent), nprocs=2, join=True)
    got = torch.load(out)
    eng, batch = _engine(joint_way, None, n_ent)
    loss = eng.train_step(*batch)
    torch.cuda.synchronize()
    ref_p, ref_g = (_flat(d) for d in _named(eng))
    assert abs(got["loss"] - float(loss)) <= 1e-4 * max(1.0, abs(float(loss)))
    scale = float(ref_g.abs().max())
    assert float((got["grad"] - ref_g).abs().max()) <= 2e-5 * scale            # summation order only (atomics, split partial sums)
    # one AdamW step on the summed gradient.  The first Adam update is lr * g / (|g| + eps): it is only well conditioned where the
    # gradient is clear of its own summation noise, so compare there
    solid = ref_g.abs() > 1e-3 * scale
    assert int(solid.sum()) > 1000
    assert float((got["param"] - ref_p)[solid].abs().max()) <= 5e-6


@pytest.mark.parametrize("row_sharded,n_ent", [(False, 1100), (True, 2200)])
def test_two_rank_step_with_the_grouped_weight_gradient_launch(tmp_path, row_sharded, n_ent):
    """Feature tables wide enough (>= 1,024 columns, >= 512 rows per rank) for the engine's ONE grouped launch of both first-projection weight gradients:
    with a reducer (round 5) that launch is kept and everything behind the tables -- projection weights + folded fusion parameters -- travels as ONE
    collective behind its reduce (before: one launch and one collective per modality).  Two ranks = the one-rank step on the global batch."""
    out = str(tmp_path / "r0.pt")
    fd = (1280, 1024)
    mp.spawn(_worker, args=(2, _free_port(), "Mformer_hd_graph", out, row_sharded, n_ent, 2, False, 0.0, 0, fd), nprocs=2, join=True)
    got = torch.load(out)
    eng, batch = _engine("Mformer_hd_graph", None, n_ent, feat_dims=fd)
    losses = [float(eng.train_step(*batch)) for _ in range(2)]
    torch.cuda.synchronize()
    ref_p, ref_g = (_flat(d) for d in _named(eng))
    for a, b in zip(got["losses"], losses):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b))
    scale = float(ref_g.abs().max())
    assert float((got["grad"] - ref_g).abs().max()) <= 5e-5 * scale
    assert float((got["param"] - ref_p).abs().max()) <= 2.5e-3            # two AdamW steps of lr 1e-3 (sign flips where the gradient is noise)


@pytest.mark.parametrize("n_ent,callable_reducer", [(901, False), (901, True), (900, True)])
def test_row_sharded_steps_do_not_accumulate_stale_entity_gradient_rows(tmp_path, n_ent, callable_reducer):
    """Three steps in a row.  Every rank writes only ITS rows of the entity-table gradient; whatever an exchange left in the foreign
    rows must not be summed into the next step's exchange (901 rows: padded shards; the plain-callable reducer sums the whole flat
    buffer).  Parameters after three AdamW steps against the single-rank run on the global batch."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), "Mformer_hd_graph", out, True, n_ent, 3, callable_reducer), nprocs=2, join=True)
    got = torch.load(out)
    eng, batch = _engine("Mformer_hd_graph", None, n_ent)
    losses = []
    for _ in range(3):
        losses.append(float(eng.train_step(*batch)))
    torch.cuda.synchronize()
    ref_p, ref_g = (_flat(d) for d in _named(eng))
    for a, b in zip(got["losses"], losses):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (got["losses"], losses)
    scale = float(ref_g.abs().max())
    assert float((got["grad"] - ref_g).abs().max()) <= 5e-5 * scale          # (a stale copy of the previous gradient would be O(scale))
    # Adam turns last-bit differences of near-zero gradient entries into +-lr: compare the parameters where the gradient is solid
    solid = ref_g.abs() > 1e-3 * scale
    assert float((got["param"] - ref_p)[solid].abs().max()) <= 3 * 1e-3 * 0.05


@pytest.mark.parametrize("row_sharded,n_ent", [(False, 900), (True, 900), (True, 901)])
def test_two_rank_step_with_dropout_and_noise_equals_the_one_rank_step(tmp_path, row_sharded, n_ent):
    """SURVEY par. 8(e): the G-rank step equals the 1-rank step on the same global batch -- with the Bert dropouts (0.1), ProjectionHead's
    dropout and the feature / entity noise (0.2 / 0.7) ON.  Every random stream that covers entity rows is keyed on the GLOBAL row
    index, so a rank that owns rows [lo, hi) draws exactly what the single rank draws for those rows."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), "Mformer_hd_graph", out, row_sharded, n_ent, 2, False, 0.1, 1), nprocs=2, join=True)
    got = torch.load(out)
    eng, batch = _engine("Mformer_hd_graph", None, n_ent, 0.1, 1)
    eng.update_noise()
    losses = [float(eng.train_step(*batch)) for _ in range(2)]
    torch.cuda.synchronize()
    ref_p, ref_g = (_flat(d) for d in _named(eng))
    for a, b in zip(got["losses"], losses):
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (got["losses"], losses)
    scale = float(ref_g.abs().max())
    assert float((got["grad"] - ref_g).abs().max()) <= 5e-5 * scale
