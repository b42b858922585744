"""Child process of tests/test_rccl_gpu.py: brings up backend "nccl" (= RCCL) with ONE rank on cuda:0 and runs the engine's training
step through GradAllReduce(force=True) -- the bucketed, overlapped start()/finish() path that ships for N > 1 -- next to the same
steps without a reducer.  Prints one JSON line.  Always started as a fresh process (never re-exec a process that touched the GPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29571")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    real_stdout = os.dup(1)
    os.dup2(2, 1)                        # RCCL prints its banner on stdout
    import torch
    import torch.distributed as dist
    from snag_amd.mkgc import MKGCConfig, MKGCHotPath
    from snag_amd.parallel import GradAllReduce, common_step_count
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend()}

    def engine(joint_way, deterministic, shard=None):
        cfg = MKGCConfig(n_ent=2000, n_rel=13, dim=128, feat_dims=[160, 96], num_proj=2, joint_way=joint_way, margin=12.0, neg_num=16, dropout=0.1,
                         add_noise=1, seed=5, lr=1e-3, deterministic=deterministic)
        g = torch.Generator().manual_seed(17)
        feats = [torch.rand(2000, 160, generator=g), torch.randn(2000, 96, generator=g)]
        eng = MKGCHotPath(cfg, feats, device=dev, shard=shard)
        eng.p["ent"].copy_(torch.randn(2000, 128, generator=g).to(dev))
        B = 256
        batch = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).to(dev) for hi, sz in ((2000, (B,)), (13, (B,)), (2000, (B,)), (2000, (B, 16)))]
        return eng, batch

    for joint_way in ("Mformer_hd_graph", "Mformer_hd_mean"):
        res = {}
        from snag_amd.parallel import RowShard
        for name, reducer, shard in (("plain", None, None), ("rccl", GradAllReduce(1, force=True, bucket_bytes=1 << 20), None),
                                     ("plain_rows", None, None), ("rccl_rows", GradAllReduce(1, force=True), RowShard(2000, 0, 1, force=True)),
                                     ("plain_one_launch", None, None)):
            # "rccl_rows": the row-sharded data path too -- all-gather of the read-outs, reduce-scatter of their gradient, in-place
            # gather of the entity-table gradient -- as RCCL collectives of one rank (identities, so bit-equality still holds).  A
            # row-sharded scorer cannot carry the gradient through the read-out rows' LayerNorm (the rows it scatters to belong to
            # other ranks too), so its one-rank twin "plain_rows" is the plain engine with that fusion off: the LayerNorm's weight
            # gradient is then summed in the same order.
            eng, batch = engine(joint_way, deterministic=True, shard=shard)       # fixed-point scatter: steps are bit-reproducible, so "equal" means bit-equal
            if name == "plain_rows":
                eng._ln_fuse = False
            if name == "plain_one_launch":      # the DEFAULT single-rank engine in the form it takes above 10,240 rows: one-launch read-out forward + fused backward
                eng.force_one_launch = True
            eng.update_noise()
            losses = [float(eng.train_step(*batch, allreduce=reducer))]
            g1 = eng.gflat.clone()              # the first step's gradient: same parameters in every variant, so only summation orders differ
            losses += [float(eng.train_step(*batch, allreduce=reducer)) for _ in range(3)]
            torch.cuda.synchronize()
            res[name] = (eng.flat.clone(), eng.gflat.clone(), eng.exp_avg_sq.clone(), losses, g1)
        a, b, c = res["plain"], res["rccl"], res["rccl_rows"]
        ar = res["plain_rows"]
        out[joint_way] = {"params_bit_equal": bool(torch.equal(a[0], b[0])), "grads_bit_equal": bool(torch.equal(a[1], b[1])),
                          "moments_bit_equal": bool(torch.equal(a[2], b[2])), "losses_equal": a[3] == b[3], "loss0": a[3][0], "loss3": a[3][3],
                          "rows_params_bit_equal": bool(torch.equal(ar[0], c[0])), "rows_grads_bit_equal": bool(torch.equal(ar[1], c[1])),
                          "rows_losses_equal": ar[3] == c[3], "ln_tail_losses": [a[3], ar[3]],
                          # the DEFAULT single-rank engine (LayerNorm backward inside the scorer; and its one-launch form) against the row-sharded
                          # run over RCCL: dgamma / dbeta / dbo / dbqkv are summed in another order, nothing else differs -- bounded here
                          "default_vs_rows_grad_rel": float((a[4] - c[4]).abs().max() / c[4].abs().max()),
                          "one_launch_vs_rows_grad_rel": float((res["plain_one_launch"][4] - c[4]).abs().max() / c[4].abs().max()),
                          "one_launch_losses": res["plain_one_launch"][3], "rows_losses": c[3]}
    # the default (fp32-atomic, fused scorer) step through the same collectives: equal up to the atomics' arrival order
    eng_a, batch = engine("Mformer_hd_graph", deterministic=False)
    eng_b, _ = engine("Mformer_hd_graph", deterministic=False)
    red = GradAllReduce(1, force=True)
    la, lb = [], []
    for _ in range(3):
        la.append(float(eng_a.train_step(*batch)))
        lb.append(float(eng_b.train_step(*batch, allreduce=red)))
    torch.cuda.synchronize()
    # (parameters are not compared here: Adam's first steps are lr * g / (|g| + eps), which turns last-bit differences of near-zero
    # gradient entries into +-lr; the bit-exact comparison above is the one that checks the ordering)
    out["fused_losses"] = [la, lb]
    # an epoch driven through train_epoch: the step count is agreed over the nccl group (device tensor), every step joins the exchange
    h, r, t = (torch.randint(0, hi, (1000,), device=dev, dtype=torch.int32) for hi in (2000, 13, 2000))
    out["steps_agreed"] = common_step_count(1000, 256)
    total = eng_b.train_epoch(h, r, t, 256, allreduce=red)
    torch.cuda.synchronize()
    out["epoch_loss_finite"] = bool(torch.isfinite(total))
    dist.destroy_process_group()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
