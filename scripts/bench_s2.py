"""GPU box: the stride-2 encoder levels of the 160x192x224 U-Net (csrc/conv3ds2.hip) one by one: forward, data gradient,
weight gradient (DFMIR_CONV3D_NO_S2=1: the generic gather kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

dev = "cuda"
def timeit(fn, reps=20):
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

print("DFMIR_CONV3D_NO_S2 =", os.environ.get("DFMIR_CONV3D_NO_S2"), " DFMIR_S2W_MINP =", os.environ.get("DFMIR_S2W_MINP"))
for Cin, Cout, sp in ((16, 32, (80, 96, 112)), (32, 32, (40, 48, 56)), (32, 32, (20, 24, 28)), (16, 32, (64, 64, 64)), (32, 32, (32, 32, 32))):
    g = torch.Generator(device=dev); g.manual_seed(1)
    osp = tuple((s + 1) // 2 for s in sp)
    x = torch.randn(1, Cin, *sp, device=dev, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev, generator=g) / (Cin * 27) ** 0.5
    b = torch.randn(Cout, device=dev, generator=g)
    dy = torch.randn(1, Cout, *osp, device=dev, generator=g)
    fl = 2.0 * Cout * osp[0] * osp[1] * osp[2] * Cin * 27
    with torch.no_grad():
        wt, wd = ops.weight_pack(w, 0), ops.weight_pack(w, 1)
        K = (3, 3, 3)
        mf = timeit(lambda: ops.conv_raw(x, wt, b, Cout, K, 2, (1, 1, 1), 1, 0, 1, 0.2, osp))
        md = timeit(lambda: ops.conv_raw(dy, wd, None, Cin, K, 1, (1, 1, 1), 2, 0, 0, 0.0, sp))
        dw = ops.zeros((27, Cin, Cout), dev)
        mw = timeit(lambda: ops.conv_wgrad_raw(x, dy, K, 2, (1, 1, 1), 0, out=dw))
    print("%2d->%2d @%-10s -> %-10s %5.2f GF   fwd %6.3f ms  dgrad %6.3f ms  wgrad %6.3f ms" % (
        Cin, Cout, "x".join(map(str, sp)), "x".join(map(str, osp)), fl / 1e9, mf, md, mw))
