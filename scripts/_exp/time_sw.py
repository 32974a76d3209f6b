import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dfmir_amd import ops
n, Cin, Cout, H = 16, 34, 16, 256
x = torch.randn(n, Cin, 1, H, H, device="cuda"); dy = torch.randn(n, Cout, 1, H, H, device="cuda")
def run():
    ops.conv_wgrad_raw(x, dy, (1, 3, 3), 1, (0, 1, 1), 0)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("%-28s %.1f us (incl. the zero-fill of dW)" % (os.environ.get("TAG", "?"), e0.elapsed_time(e1) / 10 * 1e3))
