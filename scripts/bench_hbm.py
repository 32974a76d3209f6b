"""GPU box: achieved HBM bandwidth (ALGORITHMIC bytes / time) of the memory-bound kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

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

def report(name, nbytes, ms):
    print("%-46s %9.1f MB  %8.3f ms  %7.1f GB/s  (%.0f%% of 8 TB/s)" % (name, nbytes / 1e6, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80.0))

dev = "cuda"
# ---- warps (fwd 4*(C+nd+C)*N, bwd 4*(C + 2C + 2nd)*N)
# field regimes: "smooth" = a regularised registration field (control points every 32 voxels, ~1 voxel
# rms, what VoxelMorph's smoothness loss produces); "rough" = control points every 16 voxels, 3 voxels rms
# (local variation beyond the LDS window: exercises the per-voxel fallback of warp_win.hip)
for name, shp, C, cell, amp in (("warp3d 160x192x224 C=1 smooth", (1, 160, 192, 224), 1, 32, 1.0),
                                ("warp3d 160x192x224 C=1 rough", (1, 160, 192, 224), 1, 16, 3.0),
                                ("warp3d half-res self C=3 smooth", (1, 80, 96, 112), 3, 16, 1.0),
                                ("warp2d 256^2 B=16 C=1 smooth", (16, 256, 256), 1, 32, 1.0),
                                ("warp2d 128^2 B=16 self C=2 smooth", (16, 128, 128), 2, 16, 1.0)):
    B, sp = shp[0], shp[1:]
    nd = len(sp)
    src = torch.randn(B, C, *sp, device=dev)
    coarse = torch.randn(B, nd, *[max(2, s_ // cell) for s_ in sp], device=dev) * amp
    flow = torch.nn.functional.interpolate(coarse, size=sp, mode='trilinear' if nd == 3 else 'bilinear', align_corners=True).contiguous()
    nv = src.numel() // C
    ms = timeit(lambda: ops._warp_fwd(src, flow, 0, 0))
    report(name + " fwd", 4 * (C + nd + C) * nv, ms)
    dout = torch.randn_like(src)
    dsrc = torch.zeros_like(src); dflow = torch.empty_like(flow)
    ms = timeit(lambda: ops._warp_bwd(dout, src, flow, None, dflow, 0, 0))
    report(name + " bwd d(flow) only", 4 * (C + C + 2 * nd) * nv, ms)
    ms = timeit(lambda: ops._warp_bwd_dsrc(dout, src, flow, dflow, 0, 0))
    report(name + " bwd d(src)+d(flow), owner-gather", 4 * (C + 2 * C + 2 * nd) * nv, ms)

    def bwd_atomic():
        ops.zero_(dsrc)
        ops._warp_bwd(dout, src, flow, dsrc, dflow, 0, 0)
    ms = timeit(bwd_atomic)
    report(name + " bwd d(src)+d(flow), atomic form", 4 * (C + 2 * C + 2 * nd) * nv, ms)
# ---- InstanceNorm
for name, shp in (("instnorm [32,128,256,256]", (32, 128, 256, 256)), ("instnorm [32,256,64,64]", (32, 256, 64, 64))):
    x = torch.randn(*shp, device=dev)
    y = torch.empty_like(x); mean = torch.empty(shp[0] * shp[1], device=dev); rstd = torch.empty_like(mean)
    from dfmir_amd._lib import lib, check
    S = shp[2] * shp[3]
    ms = timeit(lambda: check(lib().dfmir_instnorm_fwd(ops._p(x), None, ops._p(y), ops._p(mean), ops._p(rstd), shp[0] * shp[1], S, 1e-5, 1, None, ops._st())), 10)
    report(name + " fwd+relu", 8 * x.numel(), ms)
    dy = torch.randn_like(x); dx = torch.empty_like(x)
    ms = timeit(lambda: check(lib().dfmir_instnorm_bwd(ops._p(dy), ops._p(x), ops._p(mean), ops._p(rstd), ops._p(dx), shp[0] * shp[1], S, 1, None, ops._st())), 10)
    report(name + " bwd", 12 * x.numel(), ms)
# ---- blur / pad / adam
x = torch.randn(32, 128, 256, 256, device=dev)
ms = timeit(lambda: ops.blur_down(x), 10); report("blur_down [32,128,256,256] fwd", 4 * x.numel() * 1.25, ms)
n = 11365633
p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
v.abs_()
ms = timeit(lambda: ops.adam_step(p, g, m, v, 2e-4, 0.5, 0.999, 1e-8, 3)); report("adam 11.37 M params", 28 * n, ms)
