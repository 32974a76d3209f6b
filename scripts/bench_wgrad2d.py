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
        print("%-13s 256->256 @64^2 n=%d  %7.4f ms  %6.1f TF algorithmic  issued frac %.3f" % (name, n, ms, fl / ms / 1e9, 3 * fl / ms / 1e9 / 2500e3 * 1e3))

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

if hasattr(h, "dfmir_cs_trace_dump"):
    import numpy as np
    buf = np.zeros(8 * 32 * 8 + 8, dtype=np.uint32)
    h.dfmir_cs_trace_dump(ctypes.c_void_p(buf.ctypes.data))
    t = buf[:2048].reshape(8, 32, 8).astype(np.int64)          # [group*4 + wave][half-step & 31][slot]
    d = lambda a, b: (a - b) & 0xffffffff
    print("forward trace (cycles), one workgroup, half-steps 2..29: [0] start [1] work done [2] after barrier [3] loads arrived (store side) [4] weights stored")
    for g in range(2):
        for wv in range(4):
            w = g * 4 + wv
            comp = [hh for hh in range(2, 30) if (hh & 1) == g]
            stor = [hh for hh in range(2, 30) if (hh & 1) != g]
            c_work = np.mean([d(t[w, hh, 1], t[w, hh, 0]) for hh in comp])
            c_bar = np.mean([d(t[w, hh, 2], t[w, hh, 1]) for hh in comp])
            s_wait = np.mean([d(t[w, hh, 3], t[w, hh, 0]) for hh in stor])
            s_work = np.mean([d(t[w, hh, 1], t[w, hh, 3]) for hh in stor])
            s_bar = np.mean([d(t[w, hh, 2], t[w, hh, 1]) for hh in stor])
            print("  group %d wave %d: compute half-step: work %6.0f barrier %5.0f | store half-step: load wait %5.0f convert+store %5.0f barrier %6.0f"
                  % (g, wv, c_work, c_bar, s_wait, s_work, s_bar))
    hs = [int(d(t[0, hh + 1, 0], t[0, hh, 0])) for hh in range(0, 31)]
    print("  half-step lengths (wave 0):", hs)
    print("  whole kernel of that workgroup: %d cycles, %d wall-clock ticks (100 MHz)" % (int(d(buf[2049], buf[2048])), int(d(buf[2051], buf[2050]))))

if hasattr(h, "dfmir_cs_wg_dump"):
    import numpy as np
    nwg = 1536 if which != "fwd32" else 1024                     # the last launch was n = 48: 48 * 16 tiles * 2 cout halves
    buf = np.zeros(2048 * 6, dtype=np.uint32)
    h.dfmir_cs_wg_dump(ctypes.c_void_p(buf.ctypes.data))
    w = buf.reshape(2048, 6)[:nwg].astype(np.int64)
    t0 = w[:, 0].min()
    st, pro, lp, en = [(w[:, i] - t0) / 100.0 for i in range(4)]   # microseconds
    cu = (w[:, 5] & 15) * 65536 + (w[:, 4] & 0xffff00)           # XCC id + (SE, SH, CU) bits of HW_ID (wave / SIMD bits dropped)
    print("forward kernel, per-workgroup wall clock of the last launch (%d workgroups, %d distinct CUs):" % (nwg, len(set(cu.tolist()))))
    print("  kernel span %.1f us; workgroup body mean %.1f us (prologue %.1f, main loop %.1f, epilogue %.1f)"
          % (en.max(), (en - st).mean(), (pro - st).mean(), (lp - pro).mean(), (en - lp).mean()))
    gaps, per_cu, idle_tail = [], [], []
    for c in set(cu.tolist()):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(st[idx])]
        per_cu.append(len(o))
        for a, b in zip(o[:-1], o[1:]):
            gaps.append(st[b] - en[a])
        idle_tail.append(en.max() - en[o[-1]])
    gaps = np.array(gaps)
    print("  workgroups per CU: min %d max %d; gap between consecutive workgroups of a CU: mean %.2f us, median %.2f, max %.2f; "
          "first start spread %.2f us; idle at the end: mean %.1f us max %.1f"
          % (min(per_cu), max(per_cu), gaps.mean(), np.median(gaps), gaps.max(), st[[np.where(cu == c)[0][np.argmin(st[np.where(cu == c)[0]])] for c in set(cu.tolist())]].max(), np.mean(idle_tail), np.max(idle_tail)))

if hasattr(h, "dfmir_w1_trace_dump"):
    import numpy as np
    buf = np.zeros(4 * 64 * 4, dtype=np.uint32)
    h.dfmir_w1_trace_dump(ctypes.c_void_p(buf.ctypes.data))
    t = buf.reshape(4, 64, 4).astype(np.int64)
    print("w1 trace (cycles), stages 3..44: [first half, vmcnt wait, barrier wait, second half + copy] | stage")
    for wv in range(4):
        a = t[wv, 3:45]
        nxt = t[wv, 4:46, 0]
        d = [((a[:, 1] - a[:, 0]) & 0xffffffff).mean(), ((a[:, 2] - a[:, 1]) & 0xffffffff).mean(), ((a[:, 3] - a[:, 2]) & 0xffffffff).mean(), ((nxt - a[:, 3]) & 0xffffffff).mean(), ((nxt - a[:, 0]) & 0xffffffff).mean()]
        print("  wave %d: %7.0f %7.0f %7.0f %7.0f | %7.0f" % ((wv,) + tuple(d)))
    for kx in range(3):
        a = t[0, 3 + kx:45:3]; nxt = t[0, 4 + kx:46:3, 0]
        print("  wave 0, kx %d: first half %.0f  vmcnt %.0f  barrier %.0f  second half %.0f" % (kx, ((a[:, 1] - a[:, 0]) & 0xffffffff).mean(), ((a[:, 2] - a[:, 1]) & 0xffffffff).mean(), ((a[:, 3] - a[:, 2]) & 0xffffffff).mean(), ((nxt[:len(a)] - a[:len(nxt), 3]) & 0xffffffff).mean()))
