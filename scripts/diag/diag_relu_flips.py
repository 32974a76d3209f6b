"""Diagnostic (GPU box): how many ReLU decisions of one generator pass differ from an fp64 run, for the HIP path and
for plain fp32 PyTorch, layer by layer -- and how far the normalised activations are from fp64 where both are positive.

Why: one flipped ReLU behind InstanceNorm switches that element's whole gradient on or off.  In a plane of 4096
elements x 256 channels x 2 images a single flip moves the weight gradient of the conv in front of it by ~1/sqrt(2 M)
= 7e-4 relative -- the size of the per-layer HIP-vs-fp64 errors test_full_size_gradients_vs_fp64_oracle reports.  If
the per-layer "HIP : fp32 PyTorch" ratios follow the flip counts, they are a lottery of rounding at |x| ~ 1e-7, not a
property of the convolution kernels' arithmetic.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import _full_size_hip, _full_size_oracle

B = 1
st32, size, A0, B0 = _full_size_oracle(O, B)
st64, _, _, _ = _full_size_oracle(O, B, double=True)
model = _full_size_hip(st32, size, B, A0, B0)
A_, B_ = C.image_pair(11, B, size, size)
x = torch.cat((A_, B_), 0)

acts = {"r32": {}, "r64": {}, "hip": {}}


def tap(store, name):
    def hook(mod, inp, out):
        store[name] = out.detach().cpu().double()
    return hook


def relu_sites(net, hip):
    sites = []
    for i, m in enumerate(net.model):
        if hasattr(m, "conv_block"):
            # oracle: the ReLU module's output; HIP: the fused InstanceNorm(relu=True) module's output
            sites.append(("model.%d.relu" % i, m.conv_block[2] if hip else m.conv_block[3]))
    return sites


hs = []
for tag, net, hip in (("r32", st32.netG, False), ("r64", st64.netG, False), ("hip", model.netG, True)):
    for name, mod in relu_sites(net, hip):
        hs.append(mod.register_forward_hook(tap(acts[tag], name)))
with torch.no_grad():
    st32.netG(x)
    st64.netG(x.double())
    model.netG(x.cuda())
    torch.cuda.synchronize()
for h in hs:
    h.remove()
print("switches:", {k: v for k, v in os.environ.items() if k.startswith("DFMIR_")})
print("%-18s %12s %12s   %12s %12s" % ("ReLU behind", "flips HIP", "flips fp32", "rms err HIP", "rms err fp32"))
tot = [0, 0]
for name in acts["r64"]:
    r64, r32, hp = acts["r64"][name], acts["r32"][name], acts["hip"][name]
    f_h = int(((hp > 0) != (r64 > 0)).sum())
    f_c = int(((r32 > 0) != (r64 > 0)).sum())
    both = (r64 > 0)
    e_h = float(((hp - r64)[both] ** 2).mean().sqrt())
    e_c = float(((r32 - r64)[both] ** 2).mean().sqrt())
    tot[0] += f_h
    tot[1] += f_c
    print("%-18s %12d %12d   %12.3e %12.3e" % (name, f_h, f_c, e_h, e_c))
print("total flips: HIP %d, fp32 PyTorch %d (of %d decisions per layer)" % (tot[0], tot[1], acts["r64"][name].numel()))
