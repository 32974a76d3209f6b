"""Diagnostic (GPU box): gradients of the step's intermediate tensors, HIP vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import _hip_model_from_oracle, _load

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ngf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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

keep_o, keep_h = {}, {}
def rg(d, k, t):
    if t.requires_grad:
        t.retain_grad()
        d[k] = t
# oracle hooks
of = st.forward
def o_forward(A, Bt):
    of(A, Bt); rg(keep_o, "fake_B", st.fake_B); rg(keep_o, "idt_B", st.idt_B)
st.forward = o_forward
onr = st.netR.forward
def o_netR(a, b, registration=False):
    out = onr(a, b, registration); rg(keep_o, "regA", out[0]); rg(keep_o, "flow", out[2]); return out
st.netR.forward = o_netR
ost = O.spatial_transform
cnt = [0]
def o_st(src, flow, mode='bilinear'):
    y = ost(src, flow, mode)
    if src.shape[1] == 1 and src.shape[-1] == size and y.requires_grad:
        rg(keep_o, "warp%d" % cnt[0], y); cnt[0] += 1
    return y
O.spatial_transform = o_st
# feats of G encoder passes
ogf = st.netG.forward
ocalls = [0]
def o_G(x, layers=(), encode_only=False):
    out = ogf(x, layers, encode_only)
    if encode_only:
        for i, f in enumerate(out):
            rg(keep_o, "enc%d_f%d" % (ocalls[0], i), f)
        ocalls[0] += 1
    return out
st.netG.forward = o_G

hf = model.forward
def h_forward():
    hf(); rg(keep_h, "fake_B", model.fake_B); rg(keep_h, "idt_B", model.idt_B)
model.forward = h_forward
hnr = model.netR.forward
def h_netR(a, b, registration=False):
    out = hnr(a, b, registration); rg(keep_h, "regA", out[0]); rg(keep_h, "flow", out[2]); return out
model.netR.forward = h_netR
hst = model.spatialTransformer.forward
hc = [0]
def h_st(src, flow):
    y = hst(src, flow)
    if y.requires_grad:
        rg(keep_h, "warp%d" % (hc[0] + 2), y); hc[0] += 1    # oracle: warp0,1 are inside netR (ys, yt)
    return y
model.spatialTransformer.forward = h_st
hgf = model.netG.forward
hcalls = [0]
def h_G(x, layers=[], encode_only=False):
    out = hgf(x, layers, encode_only)
    if encode_only:
        if torch.is_grad_enabled():
            for i, f in enumerate(out):
                rg(keep_h, "enc%d_f%d" % (hcalls[0], i), f)
        hcalls[0] += 1
    return out
model.netG.forward = h_G

A_, B_ = C.image_pair(11, B, size, size)
ref = st.step(A_, B_)
model.set_input({"A": A_, "B": B_, "A_paths": [""], "B_paths": [""]})
model.optimize_parameters()
print("oracle kept", sorted(keep_o.keys()))
print("hip kept   ", sorted(keep_h.keys()))
for k in sorted(keep_o.keys()):
    if k not in keep_h or keep_o[k].grad is None or keep_h[k].grad is None:
        continue
    g1, g2 = keep_o[k].grad, keep_h[k].grad.cpu()
    v1, v2 = keep_o[k].detach(), keep_h[k].detach().cpu()
    sc = float(g1.abs().max()); err = float((g1 - g2).abs().max())
    verr = float((v1 - v2).abs().max()) / max(float(v1.abs().max()), 1e-30)
    idx = (g1 - g2).abs().flatten().argmax().item()
    pos = []
    for s in reversed(g1.shape):
        pos.append(idx % s); idx //= s
    print("%-12s value rel %.1e | grad scale %.3e maxerr %.3e rel %.2e at %s" % (k, verr, sc, err, err / max(sc, 1e-30), list(reversed(pos))))
