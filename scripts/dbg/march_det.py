import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from dfmir_amd import ops, _lib
from tests.golden import common as C
DEV = "cuda"
def run(cfg, nseg, actg, reps=200):
    Cin, Cout, N, D, H, W = cfg
    x = C.randn(301, N, Cin, D, H, W).to(DEV)
    w = (C.randn(302, Cout, Cin, 3, 3, 3) / (Cin * 27) ** 0.5).to(DEV)
    src = C.randn(305, N, Cout, D, H, W).to(DEV)
    wt = ops.weight_pack(w, 0)
    xa = ops.absmax(x)
    _lib.set_option("DFMIR_MARCH_NSEG", str(nseg))
    ref = None; nbad = 0; worst = 0.0
    for r in range(reps):
        y1 = ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, (D, H, W), x_amax=xa,
                          act_src=src if actg else None, act_slope=0.2)
        if ref is None: ref = y1.clone()
        elif not torch.equal(ref, y1):
            nbad += 1; worst = max(worst, float((ref - y1).abs().max()))
        # disturb the allocator / caches between runs
        junk = torch.randn(1 << 20, device=DEV)
    _lib.set_option("DFMIR_MARCH_NSEG", None)
    torch.cuda.synchronize()
    print(cfg, "nseg", nseg, "actg", actg, "non-identical runs:", nbad, "of", reps, "worst diff %.3e" % worst)
for cfg in [(32, 16, 1, 11, 24, 64), (16, 32, 1, 11, 24, 64), (16, 16, 1, 11, 24, 64), (32, 16, 2, 20, 80, 96), (16, 32, 2, 20, 80, 96), (16, 16, 2, 20, 80, 96)]:
    for actg in (False, True):
        run(cfg, 1, actg)
        run(cfg, 3, actg)
