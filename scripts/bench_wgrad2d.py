"""Micro-benchmark (GPU box): the dominant 2-D shapes of the step -- 256 -> 256 3x3 reflect @64^2 at n = 32 (the full
generator pass) and n = 48 (the stacked query pass) -- forward and weight gradient, REPS launches back to back (long
enough to reach the power-capped clock).  DFMIR_HIP_LIB selects a knock-out / variant build (scripts/build_ko.sh).
With a W2_TRACE build: prints the per-phase cycle stamps of one workgroup of the weight-gradient kernel."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dfmir_amd
from dfmir_amd import ops

REPS = int(os.environ.get("REPS", "60"))
which = sys.argv[1] if len(sys.argv) > 1 else "both"
print("lib:", os.path.basename(dfmir_amd.LIB_PATH), {k: v for k, v in os.environ.items() if k.startswith("DFMIR_") and k != "DFMIR_HIP_LIB"})
for n in (32, 48):
    Cin = Cout = 256
    H = 64
    x = torch.randn(n, Cin, 1, H, H, device="cuda").relu_()
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.02
    b = torch.zeros(Cout, device="cuda")
    wt = ops.weight_pack(w, 0)
    dy = torch.randn(n, Cout, 1, H, H, device="cuda") * 1e-4
    xa, da = ops.absmax(x), ops.absmax(dy)
    fl = 2.0 * n * Cout * H * H * Cin * 9
    acc = ops.zeros((9, Cin, Cout), x.device)

    def fwd():
        return ops.conv_raw(x, wt, b, Cout, (1, 3, 3), 1, (0, 1, 1), 1, 1, 0, 0.0, (1, H, H), xa)

    def wgrad():
        return ops.conv_wgrad_raw(x, dy, (1, 3, 3), 1, (0, 1, 1), 1, out=acc, x_amax=xa, dy_amax=da)

    for name, fn in (("fwd", fwd), ("wgrad", wgrad)):
        if which not in ("both", name):
            continue
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(REPS):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / REPS
        print("%-5s 256->256 @64^2 n=%d  %7.4f ms  %6.1f TF algorithmic  issued frac %.3f" % (name, n, ms, fl / ms / 1e9, 3 * fl / ms / 1e9 / 2500e3 * 1e3))

h = ctypes.CDLL(dfmir_amd.LIB_PATH)
if hasattr(h, "dfmir_w2_trace_dump"):
    import numpy as np
    buf = np.zeros(8 * 64 * 8, dtype=np.uint32)
    h.dfmir_w2_trace_dump(ctypes.c_void_p(buf.ctypes.data))
    t = buf.reshape(8, 64, 8).astype(np.int64)
    print("wgrad trace (cycles; wave: mean over runs 8..55 of [first phase, second phase, barrier wait, iteration])")
    for wv in range(8):
        a = t[wv, 8:56]
        d1 = (a[:, 1] - a[:, 0]) & 0xffffffff
        d2 = (a[:, 2] - a[:, 1]) & 0xffffffff
        d3 = (a[:, 3] - a[:, 2]) & 0xffffffff
        it = (t[wv, 9:57, 0] - t[wv, 8:56, 0]) & 0xffffffff
        print("  wave %d (group %d): %7.0f %7.0f %7.0f | %7.0f" % (wv, wv >> 2, d1.mean(), d2.mean(), d3.mean(), it.mean()))
    for wv in (0, 4):
        it = (t[wv, 1:64, 0] - t[wv, 0:63, 0]) & 0xffffffff
        print("  wave %d iteration lengths, runs 0..62:" % wv, " ".join(str(int(v)) for v in it))
        first = 0 if wv < 4 else 1          # the stamp the convert phase starts at
        lw = (t[wv, 8:56, 4] - t[wv, 8:56, first]) & 0xffffffff
        print("  wave %d: wait for the run's global loads at the start of its convert phase: mean %.0f max %d" % (wv, lw.mean(), lw.max()))
    # skew of the phases between the two waves of SIMD 0 (waves 0 and 4), runs 8..15
    for r in range(8, 12):
        base = int(t[0, r, 0])
        print("  run %d: " % r + "  ".join("w%d %s" % (wv, [int((int(v) - base) & 0xffffffff) if ((int(v) - base) & 0xffffffff) < 1 << 30 else int((int(v) - base) & 0xffffffff) - (1 << 32) for v in t[wv, r]]) for wv in (0, 4)))
