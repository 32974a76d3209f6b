"""Micro-experiment (GPU box): does an HBM-bound InstanceNorm pass hide under the power-capped split conv when the two
run on different HIP streams over independent halves of the batch?  Prints conv alone, IN alone, back to back, overlapped."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = 100
C, H = 256, 64
x = torch.randn(n, C, 1, H, H, device="cuda")
w = torch.randn(C, C, 3, 3, device="cuda") * 0.02
wt = ops.weight_pack(w, 0)
xa = ops.absmax(x)
z = torch.randn(n, C, H, H, device="cuda")
dy = torch.randn(n, C, 1, H, H, device="cuda") * 1e-4
da = ops.absmax(dy)


def conv():
    return ops.conv_raw(x, wt, None, C, (1, 3, 3), 1, (0, 1, 1), 1, 1, 0, 0.0, (1, H, H), xa)


def wgrad():
    return ops.conv_wgrad_raw(x, dy, (1, 3, 3), 1, (0, 1, 1), 1, x_amax=xa, dy_amax=da)


def inorm():
    with torch.no_grad():
        return ops.instance_norm(z, relu=True)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fa, fb, overlap):
    for _ in range(3):
        fa(); fb()
        with torch.cuda.stream(s1):
            fa()
        with torch.cuda.stream(s2):
            fb()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    if overlap:
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        for _ in range(reps):
            with torch.cuda.stream(s1):
                fa()
            with torch.cuda.stream(s2):
                fb()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    else:
        for _ in range(reps):
            fa(); fb()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


nop = lambda: None
for name, fa in (("conv fwd", conv), ("wgrad", wgrad)):
    a = timed(fa, nop, False)
    b = timed(nop, inorm, False)
    ab = timed(fa, inorm, False)
    ov = timed(fa, inorm, True)
    print("n=%d %-8s alone %.3f ms | IN alone %.3f | back to back %.3f | two streams %.3f  (hidden %.0f %% of IN)" % (
        n, name, a, b, ab, ov, 100 * (ab - ov) / b))
