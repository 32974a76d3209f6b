"""Time every conv launch of one train step by geometry (synchronising around each: for attribution only)."""
import os, sys, collections, contextlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from dfmir_amd import ops
from dfmir_amd.options import default_options
from dfmir_amd.registration_model import REGISTRATIONModel
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
B, S = 16, 256
opt = default_options(batch_size=B, crop_size=S, load_size=S, ngf=64, gpu_ids=[0], checkpoints_dir="/tmp/dfmir_bench", name="bench")
torch.manual_seed(0)
model = REGISTRATIONModel(opt)
batches = [bench.synth_pairs(B, S, S, dev, i) for i in range(2)]
paths = [""] * B
def feed(i):
    a, b = batches[i % 2]
    return {"A": a, "B": b, "A_paths": paths, "B_paths": paths}
with contextlib.redirect_stdout(sys.stderr):
    model.data_dependent_initialize(feed(0)); model.setup(opt); model.parallelize()
for i in range(2):
    model.set_input(feed(i)); model.optimize_parameters()
agg = collections.defaultdict(lambda: [0, 0.0])
def wrap(name, fn, keyf):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        e = agg[(name,) + keyf(*a, **k)]; e[0] += 1; e[1] += dt
        return r
    return w
ops.conv_raw = wrap("fwd/dgrad", ops.conv_raw, lambda x5, w, b, Cout, K, stride, p3, dil, pm, act, slope, out_sp, *r, **k: (tuple(x5.shape), Cout, K, stride, dil, tuple(out_sp)))
ops.conv_wgrad_raw = wrap("wgrad", ops.conv_wgrad_raw, lambda x5, dy5, K, stride, p3, pm, **k: (tuple(x5.shape), dy5.shape[1], K, stride))
model.set_input(feed(0)); model.optimize_parameters()
torch.cuda.synchronize()
tot = sum(v[1] for v in agg.values())
print("total conv time %.1f ms (with per-launch sync overhead)" % (tot * 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%7.3f ms %3d x %7.1f us  %s" % (v[1] * 1e3, v[0], v[1] / v[0] * 1e6, k))
