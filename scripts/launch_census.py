"""GPU box: where do the launches of one steady-state train step come from?  torch.profiler over ONE step of the
bench workload; prints (a) device kernels by count, (b) the torch (aten) ops that launch kernels, grouped by the
Python line of dfmir_amd/ that issued them."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_pairs  # noqa: E402
from dfmir_amd.options import default_options  # noqa: E402
from dfmir_amd.registration_model import REGISTRATIONModel  # noqa: E402

B, S = int(os.environ.get("B", 16)), 256
dev = torch.device("cuda", 0)
opt = default_options(batch_size=B, crop_size=S, load_size=S, ngf=64, gpu_ids=[0], checkpoints_dir="/tmp/c", name="c")
for k in sys.argv[1:]:
    key, val = k.split("=")
    setattr(opt, key, eval(val))
torch.manual_seed(0)
model = REGISTRATIONModel(opt)
a, b = synth_pairs(B, S, S, dev, 1)
data = {"A": a, "B": b, "A_paths": [""] * B, "B_paths": [""] * B}
model.data_dependent_initialize(data)
model.setup(opt)
model.parallelize()
for _ in range(3):
    model.set_input(data)
    model.optimize_parameters()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.set_input(data)
    model.optimize_parameters()
    torch.cuda.synchronize()
evs = prof.events()
kern = collections.Counter()
ktime = collections.Counter()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:90]] += 1
        ktime[e.name[:90]] += e.device_time
print("device kernels in one step: %d  (%.1f ms of kernel time)" % (sum(kern.values()), sum(ktime.values()) / 1e3))
for n, c in kern.most_common(45):
    print("%5d  %8.1f us  %s" % (c, ktime[n] / c, n))
print()
by_line = collections.Counter()
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.kernels:
        where = "?"
        for fr in (e.stack or []):
            if "dfmir_amd" in fr or "bench.py" in fr:
                where = fr.strip()
                break
        by_line[(e.name, where[-90:])] += len(e.kernels)
print("aten ops that launch kernels, by issuing line:")
for (n, w), c in by_line.most_common(60):
    print("%5d  %-28s %s" % (c, n, w))

# ---- second pass: every aten op of one step with the dfmir_amd / autograd-node frame that issued it
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

SKIP = ("view", "reshape", "alias", "detach", "slice", "select", "as_strided", "unsqueeze", "squeeze", "expand",
        "permute", "transpose", "empty", "t.default", "_unsafe_view", "is_", "size", "stride", "numel", "dim",
        "_local_scalar_dense", "lift_fresh", "record_stream", "split", "unbind", "narrow", "view_as")
ops_by = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        short = name.replace("aten.", "")
        if not any(short.startswith(s) for s in SKIP):
            where = "(autograd engine / no python frame)"
            for fr in reversed(traceback.extract_stack()):
                if "dfmir_amd" in fr.filename:
                    where = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                    break
            ops_by[(short, where)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    model.set_input(data)
    model.optimize_parameters()
torch.cuda.synchronize()
print("\naten ops of one step by issuing frame (%d total):" % sum(ops_by.values()))
for (n, w), c in ops_by.most_common(80):
    print("%5d  %-30s %s" % (c, n, w))
