"""One conv shape, a few launches (for rocprofv3 --pmc).  args: kind(fwd|wgrad) Cin Cout H n reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
kind, Cin, Cout, H, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
x = torch.randn(n, Cin, 1, H, H, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
wt = ops.weight_pack(w, 0)
dy = torch.randn(n, Cout, 1, H, H, device="cuda")
xa, da = ops.absmax(x), ops.absmax(dy)
for _ in range(reps):
    if kind == "fwd":
        ops.conv_raw(x, wt, None, Cout, (1, 3, 3), 1, (0, 1, 1), 1, 1, 0, 0.0, (1, H, H), xa)
    else:
        ops.conv_wgrad_raw(x, dy, (1, 3, 3), 1, (0, 1, 1), 1, x_amax=xa, dy_amax=da)
torch.cuda.synchronize()
