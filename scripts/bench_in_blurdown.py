import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfmir_amd import ops
x = torch.randn(32, 128, 256, 256, device="cuda")
cot = torch.randn(32, 128, 128, 128, device="cuda")
xa = x.clone().requires_grad_()
for _ in range(3):
    z = ops.instance_norm_relu_blur_down(xa); z.backward(cot); xa.grad = None
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(10):
    ev[0].record(); z = ops.instance_norm_relu_blur_down(xa); ev[1].record(); z.backward(cot); ev[2].record()
    torch.cuda.synchronize(); tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2]); xa.grad = None
print("banded" if os.environ.get("DFMIR_IN_BLUR_BANDED") else "ring-free", "fwd %.3f ms (%.2f TB/s)  bwd %.3f ms (%.2f TB/s)" % (
    tf / 10, 1.342e9 / (tf / 10) / 1e9, tb / 10, 2.416e9 / (tb / 10) / 1e9))
