"""GPU box: d(loss)/d(fake_B), d(idt_B) of ONE train step (64x64, batch 2, ngf 8), HIP vs fp32 oracle, both against the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import PinnedIds, _hip_model_from_oracle, _load
size, B = 64, 2
LN = float(os.environ.get('LAMBDA_NCE', '0.25'))
LAYERS = [int(v) for v in os.environ.get('NCE_LAYERS', '0,4,8,12,16').split(',')]
def make(double):
    torch.manual_seed(11)
    st = O.RegistrationStep(size, B, ngf=8, lambda_NCE=LN, nce_layers=LAYERS)
    with torch.no_grad():
        st.netR.flow.weight.mul_(1e5)
    st.ids_hook = lambda c, feats: [C.patch_ids(c, i, f.shape[2] * f.shape[3], 256) for i, f in enumerate(feats)]
    A0, B0 = C.image_pair(7, B, size, size)
    st.data_dependent_initialize(A0, B0)
    with torch.no_grad():
        for p in st.netF.parameters():
            if p.dim() == 1:
                p.add_(0.01)
    if double:
        for m in (st.netG, st.netF, st.netR):
            m.double()
    og = st.g_loss
    def g_loss():
        st.fake_B.retain_grad(); st.idt_B.retain_grad()
        return og()
    st.g_loss = g_loss
    return st, A0, B0
s64, _, _ = make(True)
st, A0, B0 = make(False)
model, opt = _hip_model_from_oracle(st, size, B, 8)
opt.capture_step = False
opt.lambda_NCE = LN
opt.nce_layers = ','.join(str(v) for v in LAYERS)
model.nce_layers = list(LAYERS); model.criterionNCE = model.criterionNCE[:len(LAYERS)]
src = model.patch_id_source = PinnedIds()
model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""] * B, "B_paths": [""] * B})
_load(model.netF, st.netF)
model.setup(opt)
model.parallelize()
of = model.forward
cap = {}
def wrapped():
    r = of()
    model.fake.register_hook(lambda g: cap.__setitem__('fake', g.detach().clone()))
    return r
model.forward = wrapped
A_, B_ = C.image_pair(300, B, size, size)
st.step(A_, B_); s64.step(A_.double(), B_.double())
model.set_input({"A": A_, "B": B_, "A_paths": [""] * B, "B_paths": [""] * B})
model.optimize_parameters()
torch.cuda.synchronize()
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
for nm in ("fake_B", "idt_B"):
    g64 = getattr(s64, nm).grad.double(); g32 = getattr(st, nm).grad.double()
    gh = (cap['fake'][:B] if nm == "fake_B" else cap['fake'][B:]).cpu().double()
    sc = float((gh * g64).sum() / (g64 * g64).sum())
    print("   best-fit scale - 1 = %+.3e, residual after rescaling %.2e; per image rel err %s" % (sc - 1, rel(gh, sc * g64), [round(rel(gh[i], g64[i]), 6) for i in range(B)]))
    print("d %-7s HIP %.2e  fp32 %.2e   (|g| %.3e, max %.3e)   value err HIP %.2e fp32 %.2e" % (
        nm, rel(gh, g64), rel(g32, g64), float(g64.norm()), float(g64.abs().max()),
        rel(getattr(model, nm).detach().cpu().double(), getattr(s64, nm).detach()), rel(getattr(st, nm).detach().double(), getattr(s64, nm).detach())))
