"""Diagnostic (GPU box): where the generator's forward pass picks up its distance to an fp64 run, module by module, for
the HIP path and for plain fp32 PyTorch (the oracle) on the same weights and input -- rel. L2 error of every module's
output (reference module indices, models/networks.py:982-1024)."""
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
idx = [1, 3, 4, 6, 7, 8, 10, 11] + list(range(12, 21)) + [21, 22, 24, 25, 26, 28, 31]
names = {1: "conv7x7 1->64", 3: "IN+ReLU", 4: "conv 64->128", 6: "IN+ReLU", 7: "blur-down", 8: "conv 128->256", 10: "IN+ReLU",
         11: "blur-down", 21: "blur-up", 22: "conv 256->128", 24: "IN+ReLU", 25: "blur-up", 26: "conv 128->64", 28: "IN+ReLU",
         31: "conv7x7 64->1 + tanh"}
with torch.no_grad():
    _, f32 = st32.netG(x, idx, encode_only=False)
    _, f64 = st64.netG(x.double(), idx, encode_only=False)
    _, fh = model.netG(x.cuda(), list(idx), encode_only=False)
    torch.cuda.synchronize()
print("switches:", {k: v for k, v in os.environ.items() if k.startswith("DFMIR_")})
print("%-4s %-22s %12s %12s %7s" % ("idx", "module", "HIP", "fp32 CPU", "ratio"))
for i, a, b, c in zip(idx, fh, f32, f64):
    eh = float((a.cpu().double() - c).norm() / c.norm())
    ec = float((b.double() - c).norm() / c.norm())
    print("%-4d %-22s %12.3e %12.3e %7.2f" % (i, names.get(i, "ResnetBlock"), eh, ec, eh / ec))

# one layer in isolation on identical inputs: conv 256->256 reflect, InstanceNorm
import torch.nn.functional as F
from dfmir_amd import ops
torch.manual_seed(0)
xx = torch.randn(2, 256, 64, 64).relu()
w = torch.randn(256, 256, 3, 3) * 0.02
ref = F.conv2d(F.pad(xx.double(), (1, 1, 1, 1), mode="reflect"), w.double())
cpu = F.conv2d(F.pad(xx, (1, 1, 1, 1), mode="reflect"), w)
xd = xx.cuda().unsqueeze(2).contiguous()
hip = ops.conv_raw(xd, ops.weight_pack(w.cuda(), 0), None, 256, (1, 3, 3), 1, (0, 1, 1), 1, 1, 0, 0.0, (1, 64, 64), ops.absmax(xd))[:, :, 0].cpu()
print("conv 256->256 alone: HIP %.3e  fp32 CPU %.3e" % (float((hip.double() - ref).norm() / ref.norm()), float((cpu.double() - ref).norm() / ref.norm())))
y = ref.float()
inr = F.instance_norm(y.double())
inc = F.instance_norm(y)
inh = ops.instance_norm(y.cuda(), None, False, 1e-5).cpu()
print("InstanceNorm alone: fp32 CPU %.3e" % float((inc.double() - inr).norm() / inr.norm()), "" if inh is None else " HIP %.3e" % float((inh.double() - inr).norm() / inr.norm()))
