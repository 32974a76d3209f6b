"""Micro-benchmark (GPU box): the 7x7 stem 1 -> 64 at 256^2 (forward, weight gradient) through dfmir_conv7x7_c1_*."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
from dfmir_amd.ops import lib, _p, _st, check
for n in (32, 48):
    x = torch.randn(n, 1, 256, 256, device="cuda")
    w = torch.randn(64, 1, 7, 7, device="cuda") * 0.1
    b = torch.zeros(64, device="cuda")
    y = torch.empty(n, 64, 256, 256, device="cuda")
    dy = torch.randn(n, 64, 256, 256, device="cuda")
    dw = torch.zeros(64 * 49, device="cuda"); db = torch.zeros(64, device="cuda")
    def fwd():
        check(lib().dfmir_conv7x7_c1_fwd(_p(x), _p(w), _p(b), _p(y), n, 256, 256, 64, 1, _st()))
    def wg():
        check(lib().dfmir_conv7x7_c1_wgrad(_p(x), _p(dy), _p(dw), _p(db), n, 256, 256, 64, 1, _st()))
    for name, fn, mb in (("fwd", fwd, n * 65 * 65536 * 4 / 1e6), ("wgrad", wg, n * 65 * 65536 * 4 / 1e6)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print("%s n=%d %-5s %.3f ms  %.2f TB/s  %.1f TF" % (os.environ.get("DFMIR_HIP_LIB", "default")[-12:], n, name, ms, mb / ms / 1e3, 2.0 * n * 64 * 49 * 65536 / ms / 1e9))
