import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from dfmir_amd import ops, _lib
from tests.golden import common as C
DEV = "cuda"
def run(cfg, nseg, actg):
    Cin, Cout, N, D, H, W = cfg
    x = C.randn(301, N, Cin, D, H, W).to(DEV)
    w = (C.randn(302, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5).to(DEV)
    src = C.randn(305, N, Cout, D, H, W).to(DEV)
    wt = ops.weight_pack(w, 0)
    _lib.set_option("DFMIR_MARCH_NSEG", str(nseg))
    y1 = ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, (D, H, W), x_amax=ops.absmax(x),
                      act_src=src if actg else None, act_slope=0.2)
    _lib.set_option("DFMIR_MARCH_NSEG", None)
    yr = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    if actg: yr = yr * torch.where(src.double() > 0, 1.0, 0.2)
    torch.cuda.synchronize()
    e1 = (y1.double() - yr).abs()
    print(cfg, "nseg", nseg, "actg", actg, "max err %.3e scale %.3f" % (e1.max(), yr.abs().max()),
          "per-z max err:", ["%.1e" % float(e1[:, :, z].max()) for z in range(D)])
    bad = (e1 > 1e-4 * yr.abs().max()).nonzero()
    if len(bad):
        b = bad.cpu().numpy()
        for d, nm in enumerate("n c z y x".split()):
            u, cnt = np.unique(b[:, d], return_counts=True)
            print("   ", nm, dict(zip(u.tolist(), cnt.tolist())))
for actg in (False, True):
    for nseg in (1, 2, 3):
        run((16, 32, 1, 11, 24, 64), nseg, actg)
run((16, 32, 1, 11, 16, 32), 1, True)
run((16, 16, 1, 11, 24, 64), 1, True)
run((32, 16, 1, 11, 24, 64), 1, True)
