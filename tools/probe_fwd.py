import sys, time, torch
sys.path.insert(0, '.')
import bench
from snag_amd.mkgc import MKGCConfig, MKGCHotPath
cfg = MKGCConfig()
feats, batches, known = bench.synth(cfg, 1024, 4, cfg.seed, torch.device('cuda'))
neg = torch.randint(0, cfg.n_ent, (1024, 32), device='cuda', dtype=torch.int32)
eng = MKGCHotPath(cfg, feats)
eng.update_noise()
for mode in ("train", "eval"):
    eng.train(mode == "train")
    for i in range(10): eng.forward(*batches[i % 4], neg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50): eng.forward(*batches[i % 4], neg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode}: host enqueue {1e3 * (t1 - t0) / 50:.3f} ms/fwd, wall {1e3 * (t2 - t0) / 50:.3f} ms/fwd")
