import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from dfmir_amd import ops, _lib
from tests.golden import common as C
DEV = "cuda"
def run(cfg, nseg, actg, reps=30):
    Cin, Cout, N, D, H, W = cfg
    x = C.randn(301, N, Cin, D, H, W).to(DEV)
    w = (C.randn(302, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5).to(DEV)
    src = C.randn(305, N, Cout, D, H, W).to(DEV)
    wt = ops.weight_pack(w, 0)
    xa = ops.absmax(x)
    _lib.set_option("DFMIR_CONV3D_NO_MARCH", "1")
    good = ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, (D, H, W), x_amax=xa,
                        act_src=src if actg else None, act_slope=0.2)
    _lib.set_option("DFMIR_CONV3D_NO_MARCH", None)
    _lib.set_option("DFMIR_MARCH_NSEG", str(nseg))
    shown = 0
    for r in range(reps):
        y1 = ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, (D, H, W), x_amax=xa,
                          act_src=src if actg else None, act_slope=0.2)
        bad = ((y1 - good).abs() > 1e-4 * good.abs().max()).nonzero()
        if len(bad) and shown < 3:
            shown += 1
            b = bad.cpu().numpy()
            print(cfg, "nseg", nseg, "actg", actg, "run", r, "bad", len(b), "of", y1.numel())
            for d, nm in enumerate("n c z y x".split()):
                u, cnt = np.unique(b[:, d], return_counts=True)
                print("   ", nm, dict(zip(u.tolist(), cnt.tolist())))
        junk = torch.randn(1 << 20, device=DEV)
    _lib.set_option("DFMIR_MARCH_NSEG", None)
    torch.cuda.synchronize()
    print(cfg, "nseg", nseg, "actg", actg, "bad runs shown", shown)
run((16, 32, 2, 20, 80, 96), 1, False)
run((16, 32, 1, 20, 80, 96), 1, False)
run((16, 32, 1, 20, 32, 64), 1, False)
run((16, 16, 2, 20, 80, 96), 1, True)
