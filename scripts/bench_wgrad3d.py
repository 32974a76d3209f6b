"""GPU box: the 3-D weight-gradient kernel alone on the VxmDense layer shapes (ONLY=34-32,... selects; REPS=n)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

dev = "cuda"
ONLY = os.environ.get("ONLY")
REPS = int(os.environ.get("REPS", "5"))
out = []
for Cin, Cout, sp in ((34, 32, (160, 192, 224)), (32, 16, (160, 192, 224)), (16, 16, (160, 192, 224)), (48, 32, (80, 96, 112)),
                      (16, 32, (160, 192, 224)), (16, 3, (160, 192, 224))):
    if ONLY and "%d-%d" % (Cin, Cout) not in ONLY.split(","):
        continue
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(1, Cin, *sp, device=dev, generator=g)
    dy = torch.randn(1, Cout, *sp, device=dev, generator=g)
    fl = 2.0 * Cout * sp[0] * sp[1] * sp[2] * Cin * 27
    with torch.no_grad():
        xa, da = ops.absmax(x), ops.absmax(dy)
        f = lambda: ops.conv_wgrad_raw(x, dy, (3, 3, 3), 1, (1, 1, 1), 0, x_amax=xa, dy_amax=da)
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(REPS):
            f()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / REPS
    out.append("%d->%d %.3f ms %.0f TF" % (Cin, Cout, ms, fl / ms / 1e9))
    del x, dy
print(os.environ.get("DFMIR_HIP_LIB", "default")[-22:], " | ".join(out))
