"""Diagnostic (GPU box): per-parameter gradient error of one HIP train step vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import _hip_model_from_oracle, _load

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ngf = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = 1
torch.manual_seed(7)
st = O.RegistrationStep(size, B, ngf=ngf)
with torch.no_grad():
    st.netR.flow.weight.mul_(1e5)
    st.netR.flow.bias.copy_(C.randn(8, 2) * 1.0)
st.ids_hook = lambda c, feats: [C.patch_ids(c, i, f.shape[2] * f.shape[3], 256) for i, f in enumerate(feats)]
A0, B0 = C.image_pair(9, B, size, size)
st.data_dependent_initialize(A0, B0)
with torch.no_grad():
    for p in st.netF.parameters():
        if p.dim() == 1:
            p.add_(0.01)
model, opt = _hip_model_from_oracle(st, size, B, ngf)
call = [0]
base = model.netF.forward
def netF_forward(feats, num_patches=64, patch_ids=None):
    if patch_ids is None:
        patch_ids = [C.patch_ids(call[0], i, f.shape[2] * f.shape[3], 256).to("cuda") for i, f in enumerate(feats)]
        call[0] += 1
    return base(feats, num_patches, patch_ids)
model.netF.forward = netF_forward
model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""], "B_paths": [""]})
_load(model.netF, st.netF)
model.setup(opt)
A_, B_ = C.image_pair(11, B, size, size)
ref = st.step(A_, B_)
model.set_input({"A": A_, "B": B_, "A_paths": [""], "B_paths": [""]})
model.optimize_parameters()
print("losses ref", ref)
print("losses hip", model.get_current_losses())
for nm, on, hn in (("G", st.netG, model.netG), ("F", st.netF, model.netF), ("R", st.netR, model.netR)):
    for (k, po), (k2, ph) in zip(on.named_parameters(), hn.named_parameters()):
        g1, g2 = po.grad, ph.grad.cpu()
        sc = float(g1.abs().max())
        err = float((g1 - g2).abs().max())
        cos = float((g1.flatten() @ g2.flatten()) / (g1.norm() * g2.norm() + 1e-30))
        print("%s %-36s scale %.3e  maxerr %.3e  rel %.2e  cos %.6f" % (nm, k, sc, err, err / max(sc, 1e-30), cos))
