"""GPU box: bench.py's roofline_hbm block alone (trilinear warp forward / backward at 160x192x224)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
r = bench.bench_warp_hbm(torch.device("cuda", 0), bench.load_pmc()[0])
print(json.dumps(r, indent=1))
print("SUMMARY lib=%s fwd %.3f (%.1f us)  bwd %.3f (%.1f us)  rough fwd %.3f bwd %.3f" % (os.environ.get("DFMIR_HIP_LIB", "default"), r["frac"], r["avg_launch_ms"] * 1e3, r["bwd"]["frac"], r["bwd"]["avg_launch_ms"] * 1e3, r["rough_field"]["fwd"]["frac"], r["rough_field"]["bwd"]["frac"]))
