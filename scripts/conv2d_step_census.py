"""GPU box: every conv / wgrad launch of one eager 2-D train step (bench workload) with its shape-kind, FLOPs and
HIP-event time, excluding the big split kernels (conv3x3_L / wgrad3x3_L) unless ALL=1."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_pairs
from dfmir_amd import ops
from dfmir_amd.options import default_options
from dfmir_amd.registration_model import REGISTRATIONModel
B, S = 16, 256
dev = torch.device("cuda", 0)
opt = default_options(batch_size=B, crop_size=S, load_size=S, ngf=64, gpu_ids=[0], checkpoints_dir="/tmp/c", name="c")
opt.capture_step = False
torch.manual_seed(0)
model = REGISTRATIONModel(opt)
a, b = synth_pairs(B, S, S, dev, 1)
data = {"A": a, "B": b, "A_paths": [""] * B, "B_paths": [""] * B}
model.data_dependent_initialize(data); model.setup(opt); model.parallelize()
for _ in range(3):
    model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
recs = []
def prof(kind, flops, launch):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); launch(); e.record(); recs.append((kind, flops, s, e))
ops.set_conv_profiler(prof)
model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, fl, s, e in recs:
    if not os.environ.get("ALL") and kind in ("conv3x3_L", "wgrad3x3_L"):
        continue
    key = (kind, round(fl / 1e9, 2))
    ms = s.elapsed_time(e)
    c = agg.setdefault(key, [0, 0.0]); c[0] += 1; c[1] += ms
tot = 0
for (kind, gf), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print("%-14s %8.2f GF x%3d  %7.3f ms total  %6.1f TF" % (kind, gf, n, ms, gf * n / ms if ms else 0))
print("total (listed)", round(tot, 2), "ms")
