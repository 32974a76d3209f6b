"""GPU box: does a replayed step compute what the eager kernels compute on the same weights?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfmir_amd import ops
from dfmir_amd.options import default_options
from dfmir_amd.registration_model import REGISTRATIONModel
DEV = "cuda"
B, S = 2, 64
torch.manual_seed(0)
opt = default_options(batch_size=B, crop_size=S, load_size=S, ngf=8, gpu_ids=[0], checkpoints_dir="/tmp/c", name="c", capture_step=True)
model = REGISTRATIONModel(opt)
def batch(i):
    g = torch.Generator(); g.manual_seed(i)
    return {"A": (torch.rand(B, 1, S, S, generator=g) * 2 - 1).to(DEV), "B": (torch.rand(B, 1, S, S, generator=g) * 2 - 1).to(DEV), "A_paths": [""] * B, "B_paths": [""] * B}
model.data_dependent_initialize(batch(0)); model.setup(opt); model.parallelize()
for i in range(6):
    d = batch(10 + i)
    # weights before the step
    w0 = model.optimizer_G.flat_p.clone()
    with torch.no_grad():
        ref = model.netG(torch.cat([d["A"], d["B"]]))     # eager forward with the pre-step weights
    model.set_input(d)
    model.optimize_parameters()
    torch.cuda.synchronize()
    got = model.fake
    print(i, "graph" if model._graph['graph'] is not None else "eager", "max|fake - eager fake| = %.3e (scale %.3e)" % (float((got - ref).abs().max()), float(ref.abs().max())),
          "inputs equal:", bool(torch.equal(model.real_A, d["A"])))
