"""GPU box: the 1x1 weight gradients of the 2-D step on conv1x1_wgrad_k (DFMIR_NO_1X1_WGRAD=1: the generic gather kernel) --
the 64 -> 49 tap GEMM of the generator's 7x7 head at n = 32 (13.2 GFLOP, 0.95 GB) and PatchSampleF's 256 -> 256 Linear over
12 288 sampled rows (1.6 GFLOP)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops

for name, n, cin, cout, sp in (("head tap GEMM 64->49 @256^2 n=32", 32, 64, 49, (1, 256, 256)),
                               ("stem input gradient 64->49 @256^2 n=48", 48, 64, 49, (1, 256, 256)),
                               ("PatchSampleF Linear 256->256, 12288 rows", 1, 256, 256, (1, 1, 12288))):
    x = torch.randn(n, cin, *sp, device="cuda")
    dy = torch.randn(n, cout, *sp, device="cuda") * 1e-3
    fn = lambda: ops.conv_wgrad_raw(x, dy, (1, 1, 1), 1, (0, 0, 0), 0)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    fl = 2.0 * n * cin * cout * sp[0] * sp[1] * sp[2]
    by = 4.0 * n * (cin + cout) * sp[0] * sp[1] * sp[2]
    ref = torch.einsum("ncp,ndp->cd", x.flatten(2).double(), dy.flatten(2).double())
    got = fn().reshape(cin, cout).double()
    print("wgrad %-44s %7.3f ms  %6.1f TFLOP/s  %5.2f TB/s algorithmic   rel-L2 err vs fp64 %.1e" % (
        name, ms, fl / ms / 1e9, by / ms / 1e9, float((got - ref).norm() / ref.norm())))
    # the same GEMM forward (conv_mfma_k: the generic kernel; a streaming form measured slower, profiles/r06_ab_1x1_fwd.txt)
    w = torch.randn(cout, cin, 1, 1, 1, device="cuda") / cin ** 0.5
    wt = ops.weight_pack(w.view(cout, cin, 1), 0)
    f2 = lambda: ops.conv_raw(x, wt, None, cout, (1, 1, 1), 1, (0, 0, 0), 1, 0, 0, 0.0, sp)
    for _ in range(3):
        f2()
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        f2()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    yref = torch.einsum("dc,ncp->ndp", w.view(cout, cin).double(), x[:1].flatten(2).double())
    err = float((f2()[:1].flatten(2).double() - yref).norm() / yref.norm())
    print("fwd   %-44s %7.3f ms  %6.1f TFLOP/s  %5.2f TB/s algorithmic   rel-L2 err vs fp64 %.1e" % (name, ms, fl / ms / 1e9, by / ms / 1e9, err))
