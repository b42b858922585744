"""Soak: N training steps of the headline configuration (bench.synth's DB15K-shaped inputs, on-device sampling, dropout, noise) with the backward's
reduce riding in the weight-gradient launch and as a launch of its own -- every parameter and both Adam moments must be the SAME BITS after the last
step (a rare race in the riders would be amplified by the steps that follow it), and a second riders-on run must repeat the first.
    python tools/soak_riders_determinism.py [steps]"""
import os
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
import torch  # noqa: E402

import bench  # noqa: E402
from snag_amd.mkgc import MKGCConfig, MKGCHotPath  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
DEV = "cuda"
cfg = MKGCConfig()
feats, batches, known = bench.synth(cfg, 1024, 8, cfg.seed, DEV)
res = []
for riders in (True, False, True):
    eng = MKGCHotPath(cfg, feats, device=DEV)
    eng.tail_riders = riders
    neg = torch.empty(1024, cfg.neg_num, device=DEV, dtype=torch.int32)
    eng.update_noise()
    for i in range(steps):
        h, r, t = batches[i % len(batches)]
        eng.forward(h, r, t, neg, with_backward=True, sample=(7, known))
        eng.backward()
        eng.optimizer_step()
        if i % 40 == 39:
            eng.update_noise()
    torch.cuda.synchronize()
    res.append((eng.flat.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()))
ok = all(torch.equal(a, b) for a, b in zip(res[0], res[1])) and all(torch.equal(a, b) for a, b in zip(res[0], res[2]))
print(f"riders determinism over {steps} steps: {'OK (same bits: riders on = off = on again)' if ok else 'MISMATCH'}")
sys.exit(0 if ok else 1)
