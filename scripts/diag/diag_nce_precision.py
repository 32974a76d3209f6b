"""GPU box: PatchSampleF + PatchNCELoss ALONE on given feature maps (the five tapped layers' shapes at ngf 8, 64x64, batch 2):
relative L2 error of d(feature) and of the MLP gradients against an fp64 run of the oracle, HIP vs fp32 oracle."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import _load
from dfmir_amd import networks as N, ops
from dfmir_amd.patchnce import PatchNCELoss
from dfmir_amd.options import default_options
B = 2
shapes = [(B, 1, 70, 70), (B, 16, 64, 64), (B, 32, 32, 32), (B, 32, 16, 16), (B, 32, 16, 16)]
scale = float(os.environ.get("FSCALE", "1.0"))
fq0 = [C.randn(10 + i, *s) * scale for i, s in enumerate(shapes)]
fk0 = [C.randn(20 + i, *s) * scale for i, s in enumerate(shapes)]
if os.environ.get("REAL") == "1":
    # the features a freshly initialised generator really produces (ngf 8, 64 x 64): query = G's output fed back, key = the input
    torch.manual_seed(11)
    og = O.Generator(ngf=8); O.init_weights_xavier(og)
    A0 = C.image_pair(7, B, 64, 64)[0]
    with torch.no_grad():
        fk0 = [f.clone() for f in og(A0, [0, 4, 8, 12, 16], encode_only=True)]
        fq0 = [f.clone() for f in og(og(A0), [0, 4, 8, 12, 16], encode_only=True)]
    shapes = [tuple(f.shape) for f in fk0]
    print("real features: per layer mean |f| / std over positions of the channel means:", [(round(float(f.abs().mean()), 3), round(float(f.flatten(2).std(2).mean()), 4)) for f in fq0])
if os.environ.get("RELU") == "1":
    fq0 = [f.relu() for f in fq0]; fk0 = [f.relu() for f in fk0]
opf = O.PatchSampler(32, True); torch.manual_seed(5); opf.create_mlp(fk0)
with torch.no_grad():
    for p in opf.parameters():
        if p.dim() == 1: p.add_(0.01)
o64 = copy.deepcopy(opf).double()
ids = [C.patch_ids(0, i, s[2] * s[3], 256) for i, s in enumerate(shapes)]
def run_oracle(pf, dt):
    fq = [f.clone().to(dt).requires_grad_() for f in fq0]
    fk = [f.clone().to(dt) for f in fk0]
    kp, _ = pf(fk, 256, ids)
    qp, _ = pf(fq, 256, ids)
    tot = 0
    for q, k in zip(qp, kp):
        tot = tot + O.patchnce_loss(q, k.detach(), B, 0.07).mean()
    (tot / 5).backward()
    return float(tot / 5), [f.grad.double() for f in fq], {k: p.grad.double() for k, p in pf.named_parameters()}
l64, df64, g64 = run_oracle(o64, torch.float64)
l32, df32, g32 = run_oracle(opf, torch.float32)
hpf = N.PatchSampleF(use_mlp=True, init_type='xavier', init_gain=0.02, nc=32, gpu_ids=[0])
hk = [f.cuda() for f in fk0]
hpf.create_mlp(hk); _load(hpf, opf)
hq = [f.clone().cuda().requires_grad_() for f in fq0]
hids = [i.cuda() for i in ids]
with torch.no_grad():
    hkp, _ = hpf(hk, 256, hids)
hqp, _ = hpf(hq, 256, hids)
crit = PatchNCELoss(default_options(batch_size=B))
htot = 0
for q, k in zip(hqp, hkp):
    htot = htot + ops.mean(crit(q, k))
(htot / 5).backward()
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
print("loss fp64 %.9f  fp32 %.9f  HIP %.9f" % (l64, l32, float(htot / 5)))
print("%-22s %12s %12s %8s" % ("tensor", "HIP", "fp32 oracle", "ratio"))
for i in range(5):
    a, b = rel(hq[i].grad.cpu().double(), df64[i]), rel(df32[i], df64[i])
    print("%-22s %12.2e %12.2e %8.1f" % ("d feature %d" % i, a, b, a / max(b, 1e-30)))
hg = {k: p.grad.detach().cpu().double() for k, p in hpf.named_parameters()}
for k in g64:
    a, b = rel(hg[k], g64[k]), rel(g32[k], g64[k])
    print("%-22s %12.2e %12.2e %8.1f" % (k, a, b, a / max(b, 1e-30)))
