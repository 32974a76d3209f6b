"""GPU box: the generator ALONE (forward + backward with a fixed cotangent on the output and on the five tapped features):
relative L2 error of the output, the input gradient and every parameter gradient against an fp64 run of the oracle's generator,
for the HIP path and for the fp32 oracle.  python scripts/diag/diag_gen_grads.py [ngf] [size] [batch]"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import _load
from dfmir_amd import networks as N
ngf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
only_out = os.environ.get("ONLY_OUT") == "1"
torch.manual_seed(5)
og = O.Generator(ngf=ngf)
O.init_weights_xavier(og)
o64 = copy.deepcopy(og).double()
hg = N.define_G(1, 1, ngf, 'resnet_9blocks', 'instance', False, 'xavier', 0.02, False, False, [0], None)
_load(hg, og)
x = C.image_pair(52, B, size, size)[0]
def run(g, x, dt, dev):
    x = x.detach().clone().to(dev, dt).requires_grad_()
    if os.environ.get("ENC_ONLY"):
        lay = [int(v) for v in os.environ["ENC_ONLY"].split(",")]
        feats = g(x, lay, encode_only=True)
        y = feats[-1]
        # a cotangent with the structure an NCE head produces: tiny, zero-sum over positions
        loss = sum((f * (C.randn(54 + i, *f.shape) - C.randn(54 + i, *f.shape).mean((2, 3), keepdim=True)).to(dev, dt) * 1e-3).sum() for i, f in enumerate(feats))
    else:
        y, feats = g(x, [0, 4, 8, 12, 16], encode_only=False)
        cy = C.randn(53, *y.shape).to(dev, dt)
        loss = (y * cy).sum()
        if not only_out:
            loss = loss + sum((f * C.randn(54 + i, *f.shape).to(dev, dt)).sum() for i, f in enumerate(feats))
    loss.backward()
    return y.detach().cpu().double(), x.grad.detach().cpu().double(), {k: p.grad.detach().cpu().double() for k, p in g.named_parameters() if p.grad is not None}
y64, dx64, g64 = run(o64, x, torch.float64, "cpu")
y32, dx32, g32 = run(og, x, torch.float32, "cpu")
yh, dxh, gh = run(hg, x, torch.float32, "cuda")
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
print("ngf %d size %d batch %d%s" % (ngf, size, B, "  (cotangent on the output only)" if only_out else ""))
print("%-34s %12s %12s %8s" % ("tensor", "HIP", "fp32 oracle", "ratio"))
print("%-34s %12.2e %12.2e %8.1f" % ("output", rel(yh, y64), rel(y32, y64), rel(yh, y64) / rel(y32, y64)))
print("%-34s %12.2e %12.2e %8.1f" % ("d input", rel(dxh, dx64), rel(dx32, dx64), rel(dxh, dx64) / rel(dx32, dx64)))
for k in g64:
    if k.endswith(".bias") and k != "model.30.bias":
        continue
    a, b = rel(gh[k], g64[k]), rel(g32[k], g64[k])
    print("%-34s %12.2e %12.2e %8.1f" % (k, a, b, a / max(b, 1e-30)))
