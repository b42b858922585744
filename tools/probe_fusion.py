"""The drop-in MformerFusion (full layer, four outputs) at N = 15,000, M = 3, D = 128: bench.py's roofline_fusion loop alone, for a rocprofv3 kernel table
(tools/prof_fusion.sh)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
torch.cuda.set_device(0)
rec = bench.roofline_fusion(torch.device("cuda", 0), cpu_probe=False, reps=40)
print(json.dumps({k: rec[k] for k in ("fwd", "bwd", "loop_equals_module_forward")}))
