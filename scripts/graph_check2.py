"""GPU box: per-parameter gradient differences between a replayed step and the same step enqueued eagerly from the
same state; plus replay-vs-replay and eager-vs-eager (the noise floor)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfmir_amd import ops
from dfmir_amd.options import default_options
from dfmir_amd.registration_model import REGISTRATIONModel
DEV = "cuda"
from tests.golden import common as C
from tests.test_gpu_models import _hip_model_from_oracle, _load
from tests.test_oracle_golden import make_step
st_, size, B = make_step()
S = size
model, opt = _hip_model_from_oracle(st_, size, B, 8)
opt.capture_step = True
ids_state = ops.seed_patch_ids(4242, DEV)
A0, B0 = C.image_pair(93, B, size, size)
base_forward = model.netF.forward
model.netF.forward = lambda feats, num_patches=64, patch_ids=None, bf=base_forward: bf(
    feats, num_patches, patch_ids if patch_ids is not None else
    [C.patch_ids(0, i, f.shape[2] * f.shape[3], 256).to(DEV) for i, f in enumerate(feats)])
model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""] * B, "B_paths": [""] * B})
del model.netF.forward
_load(model.netF, st_.netF)
model.setup(opt)
model.parallelize()
def batch(i):
    A_, B_ = C.image_pair(200 + 2 * i, B, size, size)
    return {"A": A_.to(DEV), "B": B_.to(DEV), "A_paths": [""] * B, "B_paths": [""] * B}
def snapshot():
    return ([(o.flat_p.clone(), o.exp_avg.clone(), o.exp_avg_sq.clone(), o._steps) for o in model.optimizers], ids_state.clone())
def restore(s):
    for o, (p, m, v, n) in zip(model.optimizers, s[0]):
        o.flat_p.copy_(p); o.exp_avg.copy_(m); o.exp_avg_sq.copy_(v); o._steps = n
    ids_state.copy_(s[1]); ops.bump_weights_epoch()
def grads():
    g = {}
    for nm, net in (("G", model.netG), ("F", model.netF), ("R", model.netR)):
        for k, p in net.named_parameters():
            if nm == "G" and k.endswith(".bias") and k != "model.30.bias":
                continue
            g[nm + "." + k] = p.grad.detach().clone()
    return g
def run(d, eager):
    model._graph_state()['force_eager'] = eager
    model.set_input(d); model.optimize_parameters(); torch.cuda.synchronize()
    return grads(), list(model.get_current_losses().values())
def cmp(a, b, tag):
    rows = sorted(((float((a[k] - b[k]).norm()) / (float(b[k].norm()) + 1e-30), k) for k in a), reverse=True)
    print("   %-16s" % tag, " ".join("%s %.1e" % (k.replace("model.", "m").replace("conv_block", "cb"), r) for r, k in rows[:4]))
model.keep_dbg = True
for i in range(6):
    d = batch(i)
    s = snapshot()
    e1, le1 = run(d, True); restore(s)
    g1, lg1 = run(d, False)
    l1a = float(ops.masked_l1(model.registered, model.real_B, None, -0.95)); l1b = float(ops.masked_l1(model.idt_B, model.registered, None, -0.95))
    dbg = [float(t) for t in getattr(model, "_dbg", [])]
    print(i, "graph" if model._graph['graph'] is not None else "eager", "R", lg1[2], "ref", le1[2], "recomputed l1", l1a, l1b, "graph's inputs", dbg,
          "ptrs", [t.data_ptr() % 100000 for t in getattr(model, "_dbg", [])])
