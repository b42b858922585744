"""Host-side enqueue time per training step (no sync inside the loop) vs GPU time."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from snag_amd.mkgc import MKGCConfig, MKGCHotPath
cfg = MKGCConfig()
feats, batches, known = bench.synth(cfg, 1024, 4, cfg.seed, torch.device('cuda'))
neg = torch.randint(0, cfg.n_ent, (1024, 32), device='cuda', dtype=torch.int32)
batches = [b + (neg,) for b in batches]
eng = MKGCHotPath(cfg, feats)
eng.update_noise()
for i in range(10): eng.train_step(*batches[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50): eng.train_step(*batches[i % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / 50:.3f} ms/step, wall incl. drain {1e3 * (t2 - t0) / 50:.3f} ms/step")
