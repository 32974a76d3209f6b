"""GPU box: after ONE train step (64x64, batch 2, ngf 8, flow head x 1e5) -- per parameter tensor, how many elements received a
different Adam update (|dw| > lr / 2, i.e. the sign of a noise-level gradient flipped) in the HIP path / in the fp32 CPU oracle,
both against the fp64 CPU oracle, and the relative L2 error of the gradient itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import PinnedIds, _hip_model_from_oracle, _load, nce_sizes
size, B = 64, 2
def make(double):
    torch.manual_seed(11)
    st = O.RegistrationStep(size, B, ngf=8, lambda_NCE=float(os.environ.get('LAMBDA_NCE', '0.25')))
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
    return st, A0, B0
s64, _, _ = make(True)
st, A0, B0 = make(False)
model, opt = _hip_model_from_oracle(st, size, B, 8)
opt.capture_step = False
opt.lambda_NCE = float(os.environ.get('LAMBDA_NCE', '0.25'))
for kv in os.environ.get('OPTS', '').split(','):
    if '=' in kv:
        k_, v_ = kv.split('='); setattr(opt, k_, v_ not in ('0', 'False'))
print('opts:', os.environ.get('OPTS', ''), {k: v for k, v in os.environ.items() if k.startswith('DFMIR_') and k != 'DFMIR_HIP_LIB'})
src = model.patch_id_source = PinnedIds()
model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""] * B, "B_paths": [""] * B})
_load(model.netF, st.netF)
model.setup(opt)
model.parallelize()
w0 = {("G." + k): v.detach().clone().double() for k, v in s64.netG.named_parameters()}
w0.update({("R." + k): v.detach().clone().double() for k, v in s64.netR.named_parameters()})
w0.update({("F." + k): v.detach().clone().double() for k, v in s64.netF.named_parameters()})
A_, B_ = C.image_pair(300, B, size, size)
st.step(A_, B_); s64.step(A_.double(), B_.double())
# (eager: the id source generates fresh tensors per call)
model.set_input({"A": A_, "B": B_, "A_paths": [""] * B, "B_paths": [""] * B})
model.optimize_parameters()
torch.cuda.synchronize()
lr = 2e-4
print("%-44s %9s %12s %12s %12s %12s" % ("parameter", "numel", "flips HIP", "flips fp32", "grad err HIP", "grad err fp32"))
tot = [0, 0, 0]
for tag, mh, m32, m64 in (("G.", model.netG, st.netG, s64.netG), ("R.", model.netR, st.netR, s64.netR), ("F.", model.netF, st.netF, s64.netF)):
    ph, p32, p64 = dict(mh.named_parameters()), dict(m32.named_parameters()), dict(m64.named_parameters())
    for k, v64 in p64.items():
        if k not in ph:
            continue
        u64 = v64.detach().double() - w0[tag + k]
        uh = ph[k].detach().cpu().double() - w0[tag + k]
        u32 = p32[k].detach().double() - w0[tag + k]
        fh = int(((uh - u64).abs() > lr / 2).sum()); f32 = int(((u32 - u64).abs() > lr / 2).sum())
        g64 = v64.grad.double() if v64.grad is not None else None
        gh = ph[k].grad.detach().cpu().double() if ph[k].grad is not None else None
        g32 = p32[k].grad.double() if p32[k].grad is not None else None
        eh = float((gh - g64).norm() / (g64.norm() + 1e-300)) if gh is not None and g64 is not None else float("nan")
        e32 = float((g32 - g64).norm() / (g64.norm() + 1e-300)) if g32 is not None and g64 is not None else float("nan")
        tot[0] += v64.numel(); tot[1] += fh; tot[2] += f32
        if (os.environ.get('ALL') or k in ('model.26.weight', 'model.12.conv_block.1.weight', 'model.30.weight', 'mlp_4.0.weight', 'mlp_3.0.weight', 'mlp_0.0.weight', 'flow.weight')) and not k.endswith('bias'):
            sc = float((gh * g64).sum() / (g64 * g64).sum()); res = float((gh - sc * g64).norm() / (g64.norm() + 1e-300))
            print("%-44s %9d %12d %12d %12.2e %12.2e   best-fit scale - 1 = %+.2e, residual after rescaling %.2e" % (tag + k, v64.numel(), fh, f32, eh, e32, sc - 1, res))
print("total elements %d, updates that differ from the fp64 oracle's by more than lr/2: HIP %d, fp32 oracle %d" % tuple(tot))
