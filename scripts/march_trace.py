"""GPU box: per-phase cycle sums of one workgroup of conv3d_march_k (trace build: scripts/build_ko_march.sh trace;
DFMIR_HIP_LIB=build/ko/libdfmir_hip_m3trace.so).  Slots per wave: 0 convert + LDS store (incl. the wait for the loads),
1 issue of the next plane's loads, 2 MFMA phase, 3 epilogue, 4 barrier at the end of a plane, 5 second barrier (one-slot
form), 6 prologue."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
dev = "cuda"
sp = (160, 192, 224)
L = ops.lib()
buf = (ctypes.c_ulonglong * 64)()
names = ["cvt+st", "gload", "mfma", "epi", "bar", "bar2", "prolog", "-"]
SHAPES = ((16, 3),) if os.environ.get("ONLY_FLOW") else ((32, 16), (16, 16), (16, 32), (16, 3))
for Cin, Cout in SHAPES:
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(1, Cin, *sp, device=dev, generator=g)
    src = torch.randn(1, Cout, *sp, device=dev, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, 3, device=dev, generator=g) / (Cin * 27) ** 0.5)
    xa = ops.absmax(x); wt = ops.weight_pack(w, 0)
    for actg in ((False,) if Cout == 3 else (False, True)):
        if Cout == 3:                                    # the flow head: no activation
            run = lambda: ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0, 0.0, sp, xa)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            L.dfmir_m3_trace(buf, 1)
            run()
            torch.cuda.synchronize()
            L.dfmir_m3_trace(buf, 1)
            print("16->3 (FLOW form)   (cycles per wave, one workgroup, whole segment)")
            for w_ in range(8):
                row = [buf[w_ * 8 + i] for i in range(8)]
                tot = sum(row)
                print("  wave %d: " % w_ + "  ".join("%s %7d (%4.1f%%)" % (names[i], row[i], 100.0 * row[i] / max(tot, 1)) for i in range(7)) + "   total %d" % tot)
            continue
        for _ in range(3):
            ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0 if actg else 1, 0.2, sp, xa, act_src=src if actg else None, act_slope=0.2)
        torch.cuda.synchronize()
        L.dfmir_m3_trace(buf, 1)
        ops.conv_raw(x, wt, None, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 0 if actg else 1, 0.2, sp, xa, act_src=src if actg else None, act_slope=0.2)
        torch.cuda.synchronize()
        L.dfmir_m3_trace(buf, 1)
        print("%d->%d actg=%d   (cycles per wave, one workgroup, whole segment)" % (Cin, Cout, actg))
        for w_ in range(8):
            row = [buf[w_ * 8 + i] for i in range(8)]
            tot = sum(row)
            print("  wave %d: " % w_ + "  ".join("%s %7d (%4.1f%%)" % (names[i], row[i], 100.0 * row[i] / max(tot, 1)) for i in range(7)) + "   total %d" % tot)
    del x, src
