"""GPU box: ONE NCE term in isolation through the model's own calculate_NCE_loss: d(term)/d(query image), HIP vs fp32 oracle,
both against the fp64 oracle.  python scripts/diag/diag_nce_term.py   (NCE_LAYERS=16, SRC=B|A, TGT=idt|fake|rand)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import PinnedIds, _hip_model_from_oracle, _load
size, B = 64, 2
LAYERS = [int(v) for v in os.environ.get('NCE_LAYERS', '16').split(',')]
def make(double):
    torch.manual_seed(11)
    st = O.RegistrationStep(size, B, ngf=8, lambda_NCE=1.0, nce_layers=LAYERS)
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
    return st, A0, B0
s64, _, _ = make(True)
st, A0, B0 = make(False)
model, opt = _hip_model_from_oracle(st, size, B, 8)
opt.capture_step = False
opt.lambda_NCE = 1.0
model.nce_layers = list(LAYERS); model.criterionNCE = model.criterionNCE[:len(LAYERS)]
src_ = model.patch_id_source = PinnedIds()
model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""] * B, "B_paths": [""] * B})
_load(model.netF, st.netF)
model.setup(opt)
A_, B_ = C.image_pair(300, B, size, size)
with torch.no_grad():
    fake64 = s64.netG(torch.cat((A_, B_), 0).double())
tgt_kind = os.environ.get("TGT", "idt")
tgt0 = {"idt": fake64[B:], "fake": fake64[:B], "rand": C.randn(5, B, 1, size, size).double().tanh()}[tgt_kind].float()
src0 = B_ if os.environ.get("SRC", "B") == "B" else A_
call = 100
def run_oracle(s, dt):
    s._nce_calls = call
    t = tgt0.clone().to(dt).requires_grad_()
    l = s.nce(src0.to(dt), t)
    l.backward()
    return float(l), t.grad.double()
l64, g64 = run_oracle(s64, torch.float64)
l32, g32 = run_oracle(st, torch.float32)
src_.call = call
model._key_feats = None
th = tgt0.clone().cuda().requires_grad_()
lh = model.calculate_NCE_loss(src0.cuda(), th)
lh.backward()
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
print("layers %s src %s tgt %s: loss fp64 %.8f fp32 %.8f HIP %.8f | d tgt: HIP %.2e fp32 %.2e  (|g| %.3e)" % (
    LAYERS, os.environ.get("SRC", "B"), tgt_kind, l64, l32, float(lh), rel(th.grad.cpu().double(), g64), rel(g32, g64), float(g64.norm())))
