"""Micro-benchmark (GPU box): the generator's conv shapes through dfmir_conv_fwd / dfmir_conv_wgrad.
Usage: python scripts/bench_conv.py [batch]   (set DFMIR_CONV_GENERIC=1 for the generic kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SHAPES = [  # Cin, Cout, H, reflect
    (256, 256, 64, True), (64, 128, 256, False), (128, 256, 128, False), (256, 128, 128, False), (128, 64, 256, False)]
print("generic" if os.environ.get("DFMIR_CONV_GENERIC") == "1" else "specialised", "kernels, n =", n,
      "| 3x3 form:", "fp32 MFMA" if os.environ.get("DFMIR_CONV_FP32") else os.environ.get("DFMIR_CONV_SPLIT", "fp16x2"))
for Cin, Cout, H, refl in SHAPES:
    x = torch.randn(n, Cin, 1, H, H, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
    b = torch.zeros(Cout, device="cuda")
    wt = ops.weight_pack(w, 0)
    dy = torch.randn(n, Cout, 1, H, H, device="cuda") * 1e-4
    xa, da = ops.absmax(x), ops.absmax(dy)     # range probes of the fp16x2 split (reused, as in ConvFn)
    fl = 2.0 * n * Cout * H * H * Cin * 9
    def fwd():
        return ops.conv_raw(x, wt, b, Cout, (1, 3, 3), 1, (0, 1, 1), 1, 1 if refl else 0, 0, 0.0, (1, H, H), xa)
    def wgrad():
        return ops.conv_wgrad_raw(x, dy, (1, 3, 3), 1, (0, 1, 1), 1 if refl else 0, x_amax=xa, dy_amax=da)
    for name, fn in (("fwd", fwd), ("wgrad", wgrad)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        print("%-5s %3d->%3d @%3d^2 %s  %7.3f ms  %6.1f TFLOP/s" % (name, Cin, Cout, H, "refl" if refl else "zero", ms, fl / ms / 1e9))

# ---- accuracy of the same kernels against an fp64 convolution (relative L2 and max error)
import torch.nn.functional as F
print("accuracy vs fp64 (n = 2):")
for Cin, Cout, H, refl in SHAPES:
    Hh = min(H, 128)
    x = torch.randn(2, Cin, 1, Hh, Hh, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
    wt = ops.weight_pack(w, 0)
    y = ops.conv_raw(x, wt, None, Cout, (1, 3, 3), 1, (0, 1, 1), 1, 1 if refl else 0, 0, 0.0, (1, Hh, Hh), ops.absmax(x))
    xp = F.pad(x[:, :, 0].double(), (1, 1, 1, 1), mode="reflect" if refl else "constant")
    ref = F.conv2d(xp, w.double())
    d = (y[:, :, 0].double() - ref)
    dyy = torch.randn(2, Cout, 1, Hh, Hh, device="cuda") * 1e-4 * torch.rand(2, Cout, 1, 1, 1, device="cuda") ** 4
    dwt = ops.conv_wgrad_raw(x, dyy, (1, 3, 3), 1, (0, 1, 1), 1 if refl else 0, x_amax=ops.absmax(x), dy_amax=ops.absmax(dyy))
    dw = ops.weight_unpack(dwt, (Cout, Cin, 3, 3)).double()
    dwr = torch.nn.grad.conv2d_weight(xp, (Cout, Cin, 3, 3), dyy[:, :, 0].double())
    dd = dw - dwr
    print("  %3d->%3d @%3d^2  fwd rel L2 %.2e max %.2e   wgrad rel L2 %.2e max %.2e" % (
        Cin, Cout, Hh, float(d.norm() / ref.norm()), float(d.abs().max() / ref.abs().max()),
        float(dd.norm() / dwr.norm()), float(dd.abs().max() / dwr.abs().max())))
