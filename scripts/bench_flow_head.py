"""GPU box: (1) the flow head Conv3d(16, 3, 3, padding=1) forward on conv3d_march_k's FLOW form against the fp32-FMA kernel
(DFMIR_CONV3D_NO_FLOW_MARCH), error of each against fp64 on a crop; (2) the memory-bound helpers of the 3-D step at its
shapes: trilinear resize x0.5 / x2 of the flow (forward + adjoint), nearest_up2 + cat (forward + adjoint), LeakyReLU
backward.  Run it under DFMIR_HIP_LIB=<other build> for an A/B of (2) between libraries."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

dev = "cuda"


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


print("library:", os.environ.get("DFMIR_HIP_LIB", "in-tree"))
for sp in ((160, 192, 224), (128, 128, 128)):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(1, 16, *sp, device=dev, generator=g)
    w = torch.randn(3, 16, 3, 3, 3, device=dev, generator=g) / (16 * 27) ** 0.5
    b = torch.randn(3, device=dev, generator=g)
    S = sp[0] * sp[1] * sp[2]
    byt = 4.0 * (16 + 3) * S
    with torch.no_grad():
        xa = ops.absmax(x)
        wt = ops.weight_pack(w, 0)
        c = 24
        ref = torch.nn.functional.conv3d(x[:, :, :c + 2, :c + 2, :c + 2].double().cpu(), w.double().cpu(), b.double().cpu())
        for off in (False, True):
            if off and not hasattr(ops, "_NO_FLOW_MARCH"):
                continue
            if hasattr(ops, "_NO_FLOW_MARCH"):
                ops._NO_FLOW_MARCH = off
            fn = lambda: ops.conv_raw(x, wt, b, 3, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, sp, xa)
            ms = timeit(fn)
            y = fn()
            err = float((y[:, :, 1:c + 1, 1:c + 1, 1:c + 1].double().cpu() - ref).norm() / ref.norm())
            print("flow head 16->3 @%-12s %-22s %7.3f ms  %5.2f TB/s algorithmic  rel-L2 err vs fp64 %.1e" % (
                "x".join(map(str, sp)), "fp32-FMA kernel" if off else "march FLOW form", ms, byt / ms / 1e9, err))
        if hasattr(ops, "_NO_FLOW_MARCH"):
            ops._NO_FLOW_MARCH = False
    del x
    if os.environ.get("ONLY_FLOW"):
        continue
    # the flow at full resolution -> half (x0.5) and the integrated flow back (x2)
    half = [s // 2 for s in sp]
    f = torch.randn(1, 3, *sp, device=dev, generator=g).requires_grad_()
    h = torch.randn(1, 3, *half, device=dev, generator=g).requires_grad_()
    for name, src, out_sp, mult in (("resize x0.5", f, half, 0.5), ("resize x2", h, list(sp), 2.0)):
        with torch.no_grad():
            ms = timeit(lambda: ops.resize_linear(src, out_sp, mult))
        y = ops.resize_linear(src, out_sp, mult)
        cot = torch.randn_like(y)
        msb = timeit(lambda: torch.autograd.grad(y, src, cot, retain_graph=True))
        byt = 4.0 * (src.numel() + y.numel())
        print("%-12s @%-12s fwd %7.3f ms %5.2f TB/s   adjoint %7.3f ms %5.2f TB/s" % (
            name, "x".join(map(str, sp)), ms, byt / ms / 1e9, msb, byt / msb / 1e9))
        del y, cot
    del f, h
    # nearest_up2 + cat of the second decoder level (32 up-sampled + 32 skip channels at half resolution) and its adjoint
    q = [s // 4 for s in sp]
    a = torch.randn(1, 32, *q, device=dev, generator=g).requires_grad_()
    bb = torch.randn(1, 32, *half, device=dev, generator=g).requires_grad_()
    with torch.no_grad():
        ms = timeit(lambda: ops.UpCatFn.apply(a, bb))
    y = ops.UpCatFn.apply(a, bb)
    cot = torch.randn_like(y)
    msb = timeit(lambda: torch.autograd.grad(y, (a, bb), cot, retain_graph=True))
    byt = 4.0 * (a.numel() + bb.numel() + y.numel())
    print("upcat 32+32  @%-12s fwd %7.3f ms %5.2f TB/s   adjoint %7.3f ms %5.2f TB/s" % (
        "x".join(map(str, half)), ms, byt / ms / 1e9, msb, byt / msb / 1e9))
    del y, cot, a, bb
    # LeakyReLU backward as its own pass (16 channels at full resolution)
    yv = torch.randn(1, 16, *sp, device=dev, generator=g)
    dy = torch.randn(1, 16, *sp, device=dev, generator=g)
    dx = torch.empty_like(dy)
    ms = timeit(lambda: ops.check(ops.lib().dfmir_act_bwd(ops._p(dy), ops._p(yv), ops._p(dx), dy.numel(), 1, 0.2, ops._st())))
    print("LeakyReLU backward 16 ch @%-12s %7.3f ms %5.2f TB/s" % ("x".join(map(str, sp)), ms, 12.0 * dy.numel() / ms / 1e9))
    del yv, dy, dx
    torch.cuda.empty_cache()
