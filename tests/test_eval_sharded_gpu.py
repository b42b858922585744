"""GPU, two processes sharing the one device, gloo as the transport: the row-sharded similarity evaluation (SURVEY.md par. 8e, config 4:
each rank computes 1/G of the distance matrix; column top-k candidates, the adjusted diagonal and the column counts are exchanged)
gives the ranks, Hits@k, MR and MRR of the single-process ``rank_metrics`` exactly."""
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


def _data(n, d, seed):
    g = torch.Generator().manual_seed(seed)
    left = torch.randn(n, d, generator=g)
    right = left + 4.0 * torch.randn(n, d, generator=g)
    nrm = lambda x: x / x.norm(dim=1, keepdim=True)
    return nrm(left), nrm(right)


def _worker(rank, world, port, n, d, k, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from snag_amd import evalsim as E
        xl, xr = _data(n, d, 11)
        res = E.rank_metrics_sharded(xl.cuda(), xr.cuda(), csls_k=k)
        torch.cuda.synchronize()
        if rank == world - 1:           # any rank holds the full answer
            torch.save({key: (v.cpu() if torch.is_tensor(v) else v) for key, v in res.items()}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,d,k,world", [(1001, 96, 3, 2), (2500, 300, 0, 2), (777, 64, 10, 3)])
def test_sharded_rank_metrics_equal_single_process(tmp_path, n, d, k, world):
    """1001 / 777 rows: ragged shards (the last rank's is shorter); k = 0: no CSLS; three ranks on one device."""
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), n, d, k, out), nprocs=world, join=True)
    got = torch.load(out)
    from snag_amd import evalsim as E
    xl, xr = _data(n, d, 11)
    ref = E.rank_metrics(xl.cuda(), xr.cuda(), csls_k=k)
    assert torch.equal(got["rank_l2r"], ref["rank_l2r"].cpu()) and torch.equal(got["rank_r2l"], ref["rank_r2l"].cpu())
    for key in ("hits_l2r", "hits_r2l", "mr_l2r", "mr_r2l", "mrr_l2r", "mrr_r2l"):
        assert got[key] == ref[key], key
    assert 0.02 < ref["hits_l2r"][0] < 1.0
