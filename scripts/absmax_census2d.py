"""GPU box: which tensors of one eager 2-D train step still get a standalone dfmir_absmax launch, and from where."""
import os, sys, traceback
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
for _ in range(2):
    model.set_input(data); model.optimize_parameters()
orig = ops.absmax
def spy(t):
    st = traceback.extract_stack(limit=8)
    print("absmax", tuple(t.shape), " <- ", " <- ".join("%s:%d" % (f.name, f.lineno) for f in st[:-1][::-1][:6]))
    return orig(t)
ops.absmax = spy
model.set_input(data); model.optimize_parameters()
torch.cuda.synchronize()
