"""GPU box: the z-marching 3-D conv kernel (csrc/conv3dm.hip) on the full-resolution layer shapes -- forward with LeakyReLU
and the input-gradient form with the folded LeakyReLU derivative; HIP-event time per launch.  DFMIR_HIP_LIB selects a
knock-out build (scripts/build_ko_march.sh), DFMIR_CONV3D_NO_MARCH=1 the tiled kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

dev = "cuda"
sp = tuple(int(v) for v in os.environ.get("SP", "160,192,224").split(","))
def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
out = []
for Cin, Cout in ((32, 16), (16, 16), (16, 32)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(1, Cin, *sp, device=dev, generator=g)
    src = torch.randn(1, Cout, *sp, device=dev, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, 3, device=dev, generator=g) / (Cin * 27) ** 0.5)
    b = torch.randn(Cout, device=dev, generator=g)
    fl = 2.0 * Cout * sp[0] * sp[1] * sp[2] * Cin * 27
    with torch.no_grad():
        xa = ops.absmax(x)
        wt = ops.weight_pack(w, 0)
        ms = timeit(lambda: ops.conv_raw(x, wt, b, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 1, 0.2, sp, xa))
        ms2 = timeit(lambda: ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, sp, xa, act_src=src, act_slope=0.2))
    out.append("%d->%d %.3f / %.3f ms (%.0f TF)" % (Cin, Cout, ms, ms2, fl / ms / 1e9))
    del x, src
    torch.cuda.empty_cache()
print(os.environ.get("TAG", "-"), " | ".join(out))
