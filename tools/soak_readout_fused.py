"""Soak: the one-launch read-out forward and the backward's fused launch (csrc/readout_fused.hip) against the engine's separate launches on random entity counts, head
counts, dropout rates and modes -- read-out, loss, the saved q|k|v and every gradient bit for bit (the arithmetic of
tests/test_mkgc_gpu.py::test_one_launch_readout_forward_equals_the_separate_launches_bit_for_bit).
    python tools/soak_readout_fused.py [first_seed] [count]"""
import os
import random
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
import torch  # noqa: E402

from snag_amd.mkgc import MKGCConfig, MKGCHotPath  # noqa: E402

DEV = "cuda"
first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    rnd = random.Random(seed)
    N = rnd.choice([rnd.randint(1, 70), rnd.randint(1, 400), rnd.randint(400, 3000)])
    heads = rnd.choice([1, 2, 4])
    dropout = rnd.choice([0.0, 0.1, 0.3])
    train = rnd.random() < 0.8
    fd = [rnd.choice([8, 40, 72, 160]), rnd.choice([12, 40, 96])]
    R, B, K, D = 9, rnd.choice([1, 17, 128]), rnd.choice([1, 8]), 128
    g = torch.Generator().manual_seed(seed)
    feats = [torch.rand(N, fd[0], generator=g), torch.randn(N, fd[1], generator=g)]
    idx = [torch.randint(0, hi, sz, generator=g, dtype=torch.int32).to(DEV) for hi, sz in ((N, (B,)), (R, (B,)), (N, (B,)), (N, (B, K)))]
    ent0 = torch.randn(N, D, generator=g).to(DEV)
    out = {}
    for co in (True, False, "separate_bwd"):
        cfg = MKGCConfig(n_ent=N, n_rel=R, dim=D, feat_dims=fd, num_proj=2, num_heads=heads, joint_way="Mformer_hd_graph", margin=6.0, neg_num=K,
                         dropout=dropout, proj_dropout=0.1, add_noise=rnd.choice([0, 1]) if co is True else cfg.add_noise, seed=seed % 97)
        eng = MKGCHotPath(cfg, feats, device=DEV)
        eng.co_launch = bool(co)
        eng.force_one_launch = True
        eng._bdesc.separate_attention_backward = int(co == "separate_bwd")      # d ctx GEMM + attention backward as two launches
        eng.p["ent"].copy_(ent0)
        if cfg.add_noise:
            eng.update_noise()
        if not train:
            eng.eval()
        loss = eng.forward(*idx, with_backward=train)
        if train:
            eng.backward()
        torch.cuda.synchronize()
        qkv = eng.save[:N * 3 * 3 * D].view(N, 3, 3 * D)
        out[co] = (eng.entity_representation().clone(), loss.clone(), qkv[:, :, D:].clone(), qkv[:, 2, :D].clone(),
                   {k: v.clone() for k, v in eng.named_gradients().items()} if train else {})
    a, b, c = out[True], out[False], out["separate_bwd"]
    ok = all(torch.equal(a[i], b[i]) for i in range(4)) and all(torch.equal(a[4][k], b[4][k]) for k in a[4])
    for k in a[4]:          # the backward's one launch against its two: dbqkv to rounding (partial sums per 32 entities), everything else bit for bit
        if k == "layer0.bqkv":
            ok = ok and float((a[4][k] - c[4][k]).abs().max()) <= 2e-5 * max(1e-6, float(c[4][k].abs().max()))
        else:
            ok = ok and torch.equal(a[4][k], c[4][k])
    if not ok:
        bad += 1
        print(f"FAILED seed {seed}: N={N} heads={heads} dropout={dropout} train={train} feat_dims={fd} B={B} K={K}")
print(f"soak_readout_fused: seeds {first}..{first + count - 1}, {bad} failures")
sys.exit(1 if bad else 0)
