"""Host-side cost of one training step: wall time of the Python loop that ENQUEUES `steps` steps (no synchronisation inside), beside
the device time of the same steps.  When the device shrinks its share (row-sharded fusion on 8 GPUs: 1/8 of the entity rows per
rank), the step is bounded by whichever is larger.  `--rows-div R` emulates one rank's share of the projections / Mformer by
building the engine over n_ent / R entities (no collectives)."""
import argparse
import time

import torch

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from snag_amd.mkgc import MKGCConfig, MKGCHotPath  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--rows-div", type=int, default=1)
ap.add_argument("--graph", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = MKGCConfig()
cfg.n_ent //= a.rows_div
feats, batches, known = bench.synth(cfg, 1024, 8, cfg.seed, dev)
eng = MKGCHotPath(cfg, feats, device=dev)
if a.rows_div > 1:
    eng.tail_riders = False           # a sharded rank runs with a reducer: the backward's reduce stays a launch of its own (mkgc.backward)
neg = torch.empty(1024, cfg.neg_num, device=dev, dtype=torch.int32)
eng.update_noise()
if a.graph:
    eng.capture_step(1024, cfg.neg_num, sample=(1, known))


def step(i):
    h, r, t = batches[i % len(batches)]
    if a.graph:
        eng.replay_step(h, r, t)
        return
    eng.forward(h, r, t, neg, with_backward=True, grad_scale=1.0, sample=(1, known))
    eng.backward(1.0)
    eng.optimizer_step()


for i in range(20):
    step(i)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n_ent={cfg.n_ent} graph={a.graph}: host enqueue {1e6 * (t1 - t0) / a.steps:.1f} us/step, device-complete {1e6 * (t2 - t0) / a.steps:.1f} us/step")
