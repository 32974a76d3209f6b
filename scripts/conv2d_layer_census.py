"""GPU box: every conv forward / dgrad / weight-gradient launch of one eager 2-D train step (bench workload) by SHAPE,
with HIP-event time and rate; skips the 256-channel 64^2 layers unless ALL=1."""
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
raw, wraw = ops.conv_raw, ops.conv_wgrad_raw


def conv_raw(x5, w_tcc, bias, Cout, K, stride, pad, dil, pad_mode, act, slope, out_sp, *a_, **k_):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); y = raw(x5, w_tcc, bias, Cout, K, stride, pad, dil, pad_mode, act, slope, out_sp, *a_, **k_); e.record()
    N, Cin = x5.shape[0], x5.shape[1]
    fl = 2.0 * N * Cout * out_sp[0] * out_sp[1] * out_sp[2] * Cin * K[0] * K[1] * K[2]
    recs.append(("fwd/dgrad n%d %d->%d k%s s%d d%d in %s out %s" % (N, Cin, Cout, "x".join(map(str, K)), stride, dil,
                 "x".join(map(str, x5.shape[2:])), "x".join(map(str, out_sp))), fl, s, e))
    return y


def conv_wgrad_raw(x5, dy5, K, stride, pad, pad_mode, *a_, **k_):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); y = wraw(x5, dy5, K, stride, pad, pad_mode, *a_, **k_); e.record()
    N, Cin = x5.shape[0], x5.shape[1]
    Cout = dy5.shape[1]
    fl = 2.0 * dy5.numel() * Cin * K[0] * K[1] * K[2]
    recs.append(("wgrad     n%d %d->%d k%s s%d    in %s out %s" % (N, Cin, Cout, "x".join(map(str, K)), stride,
                 "x".join(map(str, x5.shape[2:])), "x".join(map(str, dy5.shape[2:]))), fl, s, e))
    return y


ops.conv_raw, ops.conv_wgrad_raw = conv_raw, conv_wgrad_raw
model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for kind, fl, s, e in recs:
    if not os.environ.get("ALL") and " 256->256 k1x3x3" in kind:
        continue
    c = agg.setdefault(kind, [0, 0.0, fl, []]); c[0] += 1; c[1] += s.elapsed_time(e); c[3].append(s.elapsed_time(e))
tot = 0
for kind, (n, ms, fl, each) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms
    print("%-74s %7.2f GF x%3d %7.3f ms  %6.1f TF" % (kind, fl / 1e9, n, ms, fl * n / ms / 1e9 if ms else 0)
          + ("   each: " + " ".join("%.3f" % t for t in each) if 1 < n <= 4 else ""))
print("total (listed)", round(tot, 2), "ms")
