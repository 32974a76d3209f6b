import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from dfmir_amd import ops, _lib
from dfmir_amd.ops import _p, _st, check, lib, DfConvGeom
from tests.golden import common as C
DEV = "cuda"
def run(cfg, nseg, reps=int(os.environ.get("REPS", "40"))):
    Cin, Cout, N, D, H, W = cfg
    x = C.randn(301, N, Cin, D, H, W).to(DEV)
    w = (C.randn(302, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5).to(DEV)
    wt = ops.weight_pack(w, 0)
    xa = ops.absmax(x)
    g = DfConvGeom(N, Cin, Cout, D, H, W, D, H, W, 3, 3, 3, 1, 1, 1, 1, 1, 0, 0, 0.0)
    SRC = C.randn(305, N, Cout, D, H, W).to(DEV) if os.environ.get("ACTG") else None
    _lib.set_option("DFMIR_MARCH_NSEG", str(nseg))
    ref = None
    for r in range(reps):
        y = torch.full((N, Cout, D, H, W), float("nan"), device=DEV)
        slot = ops.amax_slot(x.device, 64)
        check(lib().dfmir_conv3d_march_fwd(ctypes.byref(g), _p(x), _p(xa), 1, _p(wt), None, _p(y), _p(slot),
                                           _p(SRC) if SRC is not None else None, 0.2, _st()))
        nn = int(torch.isnan(y).sum())
        if ref is None: ref = y.clone()
        d = (y - ref).abs()
        nd = int((d > 0).sum())
        if nn or nd:
            b = (torch.isnan(y) | (d > 0)).nonzero().cpu().numpy()
            print(cfg, "run", r, "nan", nn, "diff", nd)
            for dd, nm in enumerate("n c z y x".split()):
                u, cnt = np.unique(b[:, dd], return_counts=True)
                print("   ", nm, dict(zip(u.tolist(), cnt.tolist())))
            break
    _lib.set_option("DFMIR_MARCH_NSEG", None)
    torch.cuda.synchronize()
    print(cfg, "nseg", nseg, "done")
for rep in range(3):
    run((32, 16, 2, 20, 80, 96), 1)
    run((16, 32, 2, 20, 80, 96), 1)
    run((16, 16, 2, 20, 80, 96), 1)
    run((16, 32, 2, 20, 80, 96), 3)
