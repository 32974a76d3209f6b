"""GPU box: bench.py's roofline_hbm block alone (trilinear warp forward / backward at 160x192x224)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.bench_warp_hbm(torch.device("cuda", 0)), indent=1))
