"""GPU box: weight gradient of the ConvBlocks over cat(nearest_up2(a), b) at the decoder's levels -- the parity-class kernel
(csrc/conv3duw.hip, up-sampled channels; skip channels on the direct kernel) against the direct kernel over both parts
(DFMIR_UPWGRAD_DIRECT=1 inside this script): HIP-event time per call, and the difference of the two gradients.
DFMIR_HIP_LIB selects a knock-out build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
from dfmir_amd._lib import set_option

dev = "cuda"
sp = tuple(int(v) for v in os.environ.get("SP", "160,192,224").split(","))
levels = [(32, 2, 32, 1), (32, 16, 32, 2), (32, 32, 32, 4)]      # (the 1/8 level: W = 28, materialised)
if os.environ.get("LEVELS"):
    levels = levels[:int(os.environ["LEVELS"])]


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
for Ca, Cb, Cout, div in levels:
    full = tuple(v // div for v in sp)
    low = tuple(v // 2 for v in full)
    g = torch.Generator(device=dev); g.manual_seed(1)
    a = torch.randn(1, Ca, *low, device=dev, generator=g)
    b = torch.randn(1, Cb, *full, device=dev, generator=g)
    dy = torch.randn(1, Cout, *full, device=dev, generator=g) * 1e-3
    with torch.no_grad():
        xa = ops.absmax(torch.cat([a.flatten(), b.flatten()])).clone()
        da = ops.absmax(dy).clone()
        res = {}
        for tag, off in (("par", ""), ("direct", "1")):
            set_option("DFMIR_UPWGRAD_DIRECT", off or None)
            db = torch.zeros(Cout, device=dev)
            dw = ops.conv_wgrad_raw(None, dy, (3, 3, 3), 1, (1, 1, 1), 0, x_amax=xa, dy_amax=da, db=db, parts=(a, b))
            out_ = torch.zeros_like(dw)
            ms = timeit(lambda: ops.conv_wgrad_raw(None, dy, (3, 3, 3), 1, (1, 1, 1), 0, out=out_, x_amax=xa, dy_amax=da,
                                                   db=db, parts=(a, b)))
            res[tag] = (dw, ms)
        set_option("DFMIR_UPWGRAD_DIRECT", None)
        d = (res["par"][0] - res["direct"][0]).double().norm() / res["direct"][0].double().norm()
    out.append("%d+%d->%d @%s par %.3f direct %.3f ms rel %.1e" % (Ca, Cb, Cout, "x".join(map(str, full)), res["par"][1],
                                                                  res["direct"][1], float(d)))
    del a, b, dy
    torch.cuda.empty_cache()
print(os.environ.get("TAG", "-"), " | ".join(out))
