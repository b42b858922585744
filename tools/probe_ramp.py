"""Per-step device time of the first steps after engine construction (clock ramp / cold caches): python tools/probe_ramp.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from snag_amd.mkgc import MKGCConfig, MKGCHotPath
dev = torch.device("cuda", 0)
cfg = MKGCConfig()
feats, batches, known = bench.synth(cfg, 1024, 8, cfg.seed, dev)
eng = MKGCHotPath(cfg, feats, device=dev)
neg = torch.empty(1024, cfg.neg_num, device=dev, dtype=torch.int32)
eng.update_noise()
n = 80
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 0         # forward-only passes straight before the timed steps
for i in range(pre):
    eng.forward(*batches[i % len(batches)], neg)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    t0 = time.perf_counter()
    h, r, t = batches[i % len(batches)]
    eng.forward(h, r, t, neg, with_backward=True, grad_scale=1.0, sample=(1, known))
    eng.backward(1.0)
    eng.optimizer_step()
    host.append(1e6 * (time.perf_counter() - t0))
    ev[i + 1].record()
torch.cuda.synchronize()
d = [1e3 * ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for i in range(0, n, 10):
    print(f"steps {i:2d}-{i+9:2d}: device {' '.join(f'{x:6.0f}' for x in d[i:i+10])}")
    print(f"             host   {' '.join(f'{x:6.0f}' for x in host[i:i+10])}")
