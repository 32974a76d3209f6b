"""GPU box: the 3-D stride-1 3x3x3 layer shapes of VxmDense (default features, 160x192x224) -- forward / dgrad /
wgrad rates and the error of each against an fp64 convolution on a crop.  DFMIR_CONV3D_FP32=1 selects the fp32-MFMA
kernels for an A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

dev = "cuda"
def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

print("split3d" if not os.environ.get("DFMIR_CONV3D_FP32") else "fp32 MFMA")
ONLY = os.environ.get("ONLY")          # e.g. ONLY=34-32,16-16
for Cin, Cout, sp in ((34, 32, (160, 192, 224)), (32, 16, (160, 192, 224)), (16, 16, (160, 192, 224)), (48, 32, (80, 96, 112)),
                      (64, 32, (40, 48, 56)), (32, 34, (160, 192, 224)), (16, 32, (160, 192, 224)), (16, 3, (160, 192, 224))):
    if ONLY and "%d-%d" % (Cin, Cout) not in ONLY.split(","):
        continue
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(1, Cin, *sp, device=dev, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, 3, device=dev, generator=g) / (Cin * 27) ** 0.5)
    b = torch.randn(Cout, device=dev, generator=g)
    dy = torch.randn(1, Cout, *sp, device=dev, generator=g)
    fl = 2.0 * Cout * sp[0] * sp[1] * sp[2] * Cin * 27
    with torch.no_grad():
        xa = ops.absmax(x)
        wt = ops.weight_pack(w, 0)
        K = (3, 3, 3)
        ms = timeit(lambda: ops.conv_raw(x, wt, b, Cout, K, 1, (1, 1, 1), 1, 0, 1, 0.2, sp, xa))
        y = ops.conv_raw(x, wt, b, Cout, K, 1, (1, 1, 1), 1, 0, 0, 0.0, sp, xa)
        # error vs fp64 on a crop (interior voxels only depend on the crop + 1 halo)
        c = 24
        xr = x[:, :, :c + 2, :c + 2, :c + 2].double().cpu()
        ref = torch.nn.functional.conv3d(xr, w.double().cpu(), b.double().cpu())      # valid conv: output = voxels 1..c
        got = y[:, :, 1:c + 1, 1:c + 1, 1:c + 1].double().cpu()
        err = float((got - ref).norm() / ref.norm())
        da = ops.absmax(dy)
        msw = timeit(lambda: ops.conv_wgrad_raw(x, dy, K, 1, (1, 1, 1), 0, x_amax=xa, dy_amax=da), 3)
    print("%2d->%2d @%-12s fwd %7.3f ms %6.1f TF (rel-L2 err vs fp64 %.1e)   wgrad %7.3f ms %6.1f TF" % (
        Cin, Cout, "x".join(map(str, sp)), ms, fl / ms / 1e9, err, msw, fl / msw / 1e9))
    del x, dy, y
    torch.cuda.empty_cache()
