"""GPU box: the ConvBlock over cat(nearest_up2(a), b) at the top of the 3-D U-Net (a [1,32,80,96,112], b = the two input
volumes [1,2,160,192,224], 34 -> 32 channels; torchvoxelmorph/networks.py:64,97-100,1506-1521) -- the parity-class
kernels against the materialising path (DFMIR_CONV3D_NO_UPPHASE=1): forward, forward + backward, error vs fp64 on a crop.
Also the run rocprofv3 --pmc profiles (scripts/prof_upconv3d.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
from dfmir_amd.networks import Conv3d

dev = "cuda"
Ca, Cb, Cout = 32, 2, 32
lo = (80, 96, 112)
hi = tuple(2 * s for s in lo)
g = torch.Generator(device=dev); g.manual_seed(1)
a = torch.randn(1, Ca, *lo, device=dev, generator=g)
b = torch.randn(1, Cb, *hi, device=dev, generator=g)
conv = Conv3d(Ca + Cb, Cout, 3, 1, 1).to(dev)
cot = torch.randn(1, Cout, *hi, device=dev, generator=g)

def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

def fwd():
    with torch.no_grad():
        return conv.forward_upcat(a, b, act=1, slope=0.2)

def fwd_bwd():
    ar = a.detach().requires_grad_()
    y = conv.forward_upcat(ar, b, act=1, slope=0.2)
    conv.zero_grad()
    y.backward(cot)
    return ar.grad

ref_gf = 2.0 * Cout * hi[0] * hi[1] * hi[2] * (Ca + Cb) * 27 / 1e9
ms_f = timeit(fwd)
ms_fb = timeit(fwd_bwd, 3)
tag = "materialised" if os.environ.get("DFMIR_CONV3D_NO_UPPHASE") else "parity-class"
print("%s: forward %.3f ms (%.0f TF reference-equivalent)   forward + backward %.3f ms (%.0f TF)" % (
    tag, ms_f, ref_gf / ms_f, ms_fb, 3 * ref_gf / ms_fb))
# error vs fp64 on a crop (interior outputs depend on the crop only)
c = 12
y = fwd()
ac = a[:, :, :c + 1, :c + 1, :c + 1].double().cpu()
bc = b[:, :, :2 * c + 2, :2 * c + 2, :2 * c + 2].double().cpu()
xc = torch.cat([torch.nn.functional.interpolate(ac, scale_factor=2, mode="nearest"), bc], 1)
ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv3d(xc, conv.weight.double().cpu(), conv.bias.double().cpu(), padding=1), 0.2)
got = y[:, :, :2 * c, :2 * c, :2 * c].double().cpu()
ref = ref[:, :, :2 * c, :2 * c, :2 * c]
print("forward rel-L2 error vs fp64: %.2e" % float((got - ref).norm() / ref.norm()))
