import os, sys, ctypes
sys.path.insert(0, os.getcwd())
import torch
from dfmir_amd import ops, _lib
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
names = ["gload0 issue", "lstore0", "sync0", "mfma loop", "sync A", "lstore", "sync B", "epilogue", "publish"]
dev = "cuda"
for Cin, Cout in ((34, 32), (16, 16), (16, 32)):
    sp = (160, 192, 224)
    x = torch.randn(1, Cin, *sp, device=dev); w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (Cin * 27) ** 0.5
    b = torch.randn(Cout, device=dev)
    with torch.no_grad():
        xa = ops.absmax(x); wt = ops.weight_pack(w, 0)
        for _ in range(2):
            ops.conv_raw(x, wt, b, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 1, 0.2, sp, xa)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        raw.dfmir_c3s_trace(buf, 1)
        ops.conv_raw(x, wt, b, Cout, (3, 3, 3), 1, (1, 1, 1), 1, 0, 1, 0.2, sp, xa)
        torch.cuda.synchronize()
        raw.dfmir_c3s_trace(buf, 1)
    tot = sum(buf[:9]); ntile = 40 * 24 * 14
    print("%d->%d  per-tile memtime ticks (100 MHz => x10 ns): total %.0f" % (Cin, Cout, tot / ntile))
    for n, v in zip(names, buf[:9]):
        print("   %-14s %8.1f  %5.1f%%" % (n, v / ntile, 100.0 * v / tot))
    del x
