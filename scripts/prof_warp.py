"""GPU box: a short loop of the 3-D warp forward/backward at the BASELINE volume size, to be run
under rocprofv3 (--kernel-trace --stats, or one --pmc pass at a time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

dev = "cuda"
sp = (160, 192, 224)
B, C, nd = 1, 1, 3
torch.manual_seed(0)
src = torch.randn(B, C, *sp, device=dev)
cell = int(os.environ.get('CELL', '16')); amp = float(os.environ.get('AMP', '3.0'))
coarse = torch.randn(B, nd, *[max(2, s_ // cell) for s_ in sp], device=dev) * amp
flow = torch.nn.functional.interpolate(coarse, size=sp, mode='trilinear', align_corners=True).contiguous()
dout = torch.randn_like(src)
dsrc = torch.zeros_like(src); dflow = torch.empty_like(flow)
reps = int(os.environ.get("REPS", "10"))
for _ in range(reps):
    ops._warp_fwd(src, flow, 0, 0)
    ops._warp_bwd_dsrc(dout, src, flow, dflow, 0, 0)      # owner-gather kernels (no device-scope atomics)
torch.cuda.synchronize()
def tm(fn, reps=20):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
print("cell %d amp %.1f: fwd %.1f us  bwd(dflow only) %.1f us  bwd(dsrc+dflow) %.1f us" % (
    cell, amp, tm(lambda: ops._warp_fwd(src, flow, 0, 0)),
    tm(lambda: ops._warp_bwd(dout, src, flow, None, dflow, 0, 0)),
    tm(lambda: ops._warp_bwd_dsrc(dout, src, flow, dflow, 0, 0))))
