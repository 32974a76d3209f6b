import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
r = bench.bench_warp_hbm("cuda", {})
print(os.environ.get("TAG", ""), "fwd cold %.3f (%.1f us) warm %.3f | bwd cold %.3f warm %.3f | rough fwd %.3f bwd %.3f" % (
    r["frac"], 1e3 * r["avg_launch_ms"], r["warm"]["frac"], r["bwd"]["frac"], r["bwd"]["warm"]["frac"],
    r["rough_field"]["fwd"]["frac"], r["rough_field"]["bwd"]["frac"]))
