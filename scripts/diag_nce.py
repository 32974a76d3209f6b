"""Diagnostic (GPU box): one calculate_NCE_loss backward, HIP vs oracle, gradient at every stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import _load
from tests.test_oracle_golden import make_tiny_generator
from dfmir_amd import networks as N, ops
from dfmir_amd.patchnce import PatchNCELoss
from dfmir_amd.options import default_options

B, size = 1, 64
og = make_tiny_generator()
hg = N.define_G(1, 1, 8, 'resnet_9blocks', 'instance', False, 'xavier', 0.02, False, False, [0], None)
_load(hg, og)
x = C.image_pair(52, B, size, size)[0]
src = C.image_pair(53, B, size, size)[1]
layers = [0, 4, 8, 12, 16]
# oracle
xo = x.clone().requires_grad_()
fq = og(xo, layers, encode_only=True)
for f in fq: f.retain_grad()
fk = [f.detach() for f in og(src, layers, encode_only=True)]
opf = O.PatchSampler(32, True); torch.manual_seed(5); opf.create_mlp(fk)
with torch.no_grad():
    for p in opf.parameters():
        p.mul_(20.0)
        if p.dim() == 1: p.add_(0.05)
ids = [C.patch_ids(0, i, f.shape[2] * f.shape[3], 256) for i, f in enumerate(fk)]
kp, _ = opf(fk, 256, ids)
qp, _ = opf(fq, 256, ids)
for q in qp: q.retain_grad()
tot = 0
for q, k in zip(qp, kp):
    tot = tot + (O.patchnce_loss(q, k, B, 0.07) * 0.25).mean()
(tot / 5).backward()
# hip
xh = x.clone().cuda().requires_grad_()
hq = hg(xh, layers, encode_only=True)
for f in hq: f.retain_grad()
with torch.no_grad():
    hk = hg(src.cuda(), layers, encode_only=True)
hpf = N.PatchSampleF(use_mlp=True, init_type='xavier', init_gain=0.02, nc=32, gpu_ids=[0])
hpf.create_mlp(hk); _load(hpf, opf)
hids = [i.cuda() for i in ids]
with torch.no_grad():
    hkp, _ = hpf(hk, 256, hids)
hqp, _ = hpf(hq, 256, hids)
for q in hqp: q.retain_grad()
crit = PatchNCELoss(default_options(batch_size=B))
htot = 0
for q, k in zip(hqp, hkp):
    htot = htot + ops.mean(crit(q, k)) * 0.25
(htot / 5).backward()
print("loss", float(tot / 5), float(htot / 5))
def rep(name, a, b):
    b = b.cpu()
    sc = float(a.abs().max()); err = float((a - b).abs().max())
    cos = float((a.flatten() @ b.flatten()) / (a.norm() * b.norm() + 1e-30))
    print("%-10s scale %.3e maxerr %.3e rel %.2e cos %.5f norm_ratio %.4f" % (name, sc, err, err / max(sc, 1e-30), cos, float(b.norm() / (a.norm() + 1e-30))))
for i in range(5):
    rep("q%d.grad" % i, qp[i].grad, hqp[i].grad)
for i in range(5):
    rep("f%d.grad" % i, fq[i].grad, hq[i].grad)
rep("x.grad", xo.grad, xh.grad)
for (k, po), (k2, ph) in zip(og.named_parameters(), hg.named_parameters()):
    if po.grad is not None and k.endswith("weight"):
        rep(k[-28:], po.grad, ph.grad)
