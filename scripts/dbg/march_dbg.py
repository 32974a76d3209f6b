import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from dfmir_amd import ops, _lib
from tests.golden import common as C
DEV = "cuda"
def run(cfg, nseg=None):
    Cin, Cout, N, D, H, W = cfg
    x = C.randn(301, N, Cin, D, H, W).to(DEV)
    w = (C.randn(302, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5).to(DEV)
    wt = ops.weight_pack(w, 0)
    if nseg: _lib.set_option("DFMIR_MARCH_NSEG", str(nseg))
    y1 = ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, (D, H, W), x_amax=ops.absmax(x))
    _lib.set_option("DFMIR_MARCH_NSEG", None)
    _lib.set_option("DFMIR_CONV3D_NO_MARCH", "1")
    y0 = ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, (D, H, W), x_amax=ops.absmax(x))
    _lib.set_option("DFMIR_CONV3D_NO_MARCH", None)
    yr = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    torch.cuda.synchronize()
    e1 = (y1.double() - yr).abs(); e0 = (y0.double() - yr).abs()
    print(cfg, "nseg", nseg, "march max err %.3e  tiled max err %.3e  scale %.3f" % (e1.max(), e0.max(), yr.abs().max()))
    bad = (e1 > 1e-4 * yr.abs().max()).nonzero()
    print("  bad count", len(bad), "of", e1.numel())
    if len(bad):
        b = bad.cpu().numpy()
        for d, nm in enumerate("n c z y x".split()):
            u, cnt = np.unique(b[:, d], return_counts=True)
            print("   ", nm, dict(zip(u.tolist(), cnt.tolist())))
for cfg in [(16, 32, 2, 9, 37, 72), (16, 32, 1, 9, 37, 72), (16, 32, 2, 9, 16, 32), (16, 32, 1, 6, 18, 36), (32, 16, 2, 9, 37, 72), (16, 16, 2, 9, 37, 72)]:
    run(cfg)
run((16, 32, 2, 9, 37, 72), 1)
