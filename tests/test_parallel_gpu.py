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


def _engine(joint_way, shard=None, n_ent=900):
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    cfg = MKGCConfig(n_ent=n_ent, n_rel=13, dim=128, feat_dims=[160, 96], num_proj=2, joint_way=joint_way, margin=6.0, neg_num=8, dropout=0.0,
                     add_noise=0, seed=5, lr=1e-3)
    g = torch.Generator().manual_seed(17)
    feats = [torch.rand(n_ent, 160, generator=g), torch.randn(n_ent, 96, generator=g)]
    eng = MKGCHotPath(cfg, feats, device="cuda", shard=shard)
    eng.p["ent"].copy_(torch.randn(n_ent, 128, generator=g).cuda())
    B = 64
    batch = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).cuda() for hi, sz in ((n_ent, (B,)), (13, (B,)), (n_ent, (B,)), (n_ent, (B, 8)))]
    return eng, batch


def _worker(rank, world, port, joint_way, out, row_sharded=False, n_ent=900):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snag_amd.parallel import GradAllReduce, RowShard, shard_batch
        eng, batch = _engine(joint_way, RowShard(n_ent, rank, world) if row_sharded else None, n_ent)
        local = shard_batch(batch, rank, world)
        loss = eng.train_step(*local, allreduce=GradAllReduce(world))
        torch.cuda.synchronize()
        lt = loss.detach().double().cpu().reshape(1)
        dist.all_reduce(lt)
        if rank == 0:
            torch.save({"grad": eng.gflat.cpu(), "param": eng.flat.cpu(), "loss": float(lt)}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("joint_way,row_sharded,n_ent", [("Mformer_hd_graph", False, 900), ("Mformer_hd_mean", False, 900),
                                                         ("Mformer_hd_graph", True, 900), ("Mformer_hd_mean", True, 900),
                                                         ("Mformer_hd_graph", True, 901)])
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
    ref_g, ref_p = eng.gflat.cpu(), eng.flat.cpu()
    assert abs(got["loss"] - float(loss)) <= 1e-4 * max(1.0, abs(float(loss)))
    scale = float(ref_g.abs().max())
    assert float((got["grad"] - ref_g).abs().max()) <= 2e-5 * scale            # summation order only (atomics, split partial sums)
    # one AdamW step on the summed gradient.  The first Adam update is lr * g / (|g| + eps): it is only well conditioned where the
    # gradient is clear of its own summation noise, so compare there
    solid = ref_g.abs() > 1e-3 * scale
    assert int(solid.sum()) > 1000
    assert float((got["param"] - ref_p)[solid].abs().max()) <= 5e-6
