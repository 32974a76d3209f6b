"""GPU box: soak of the two register-resident weight-gradient kernels at full size -- SOAK (400) launches each, every result compared
with the first one (the float atomics of the epilogues reorder sums: 1e-6 of the norm; a hazard or a race would be a large,
sporadic difference).  conv3d_upwgrad4_k<true> issues two of its MFMA chains from inline asm (csrc/conv3duw.hip, mma_v)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dfmir_amd import ops
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(3)
sp = (160, 192, 224)
low = tuple(v // 2 for v in sp)
a = torch.randn(1, 32, *low, device=dev, generator=g)
b = torch.randn(1, 2, *sp, device=dev, generator=g)
dy = torch.randn(1, 32, *sp, device=dev, generator=g) * 1e-3
xa = ops.absmax(torch.cat([a.flatten(), b.flatten()])).clone(); da = ops.absmax(dy).clone()
def run_up():
    db = torch.zeros(32, device=dev)
    dw = ops.conv_wgrad_raw(None, dy, (3, 3, 3), 1, (1, 1, 1), 0, x_amax=xa, dy_amax=da, db=db, parts=(a, b))
    return dw, db
x = torch.randn(1, 32, *sp, device=dev, generator=g)
dy16 = torch.randn(1, 16, *sp, device=dev, generator=g) * 1e-3
xa2 = ops.absmax(x).clone(); da2 = ops.absmax(dy16).clone()
def run_wm():
    db = torch.zeros(16, device=dev)
    dw = ops.conv_wgrad_raw(x, dy16, (3, 3, 3), 1, (1, 1, 1), 0, x_amax=xa2, dy_amax=da2, db=db)
    return dw, db
for name, fn in (("upwgrad4 fused", run_up), ("wgrad_march", run_wm)):
    with torch.no_grad():
        ref = fn()
        worst = [0.0, 0.0]
        for i in range(int(os.environ.get("SOAK", "400"))):
            got = fn()
            for j in range(2):
                e = float((got[j] - ref[j]).abs().max() / ref[j].abs().max())
                worst[j] = max(worst[j], e)
        torch.cuda.synchronize()
    print("%s: %d launches, worst max-abs difference to the first / max |.|: dW %.2e  db %.2e" % (name, int(os.environ.get("SOAK", "400")), worst[0], worst[1]))
    assert worst[0] < 1e-5 and worst[1] < 1e-5
