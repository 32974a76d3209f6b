"""GPU box: what would a fused first InstanceNorm of a ResnetBlock save?  conv (256 -> 256 @64^2, the producer) followed by
(a) the InstanceNorm + ReLU pass that writes the normalised tensor, (b) the statistics-only pass (dfmir_instnorm_stats: the
consumer would normalise while it stages its operand) -- the conv's output is cache-warm in both, as in the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfmir_amd import ops
from dfmir_amd.ops import _p, _st, check, lib

for n in (32, 48):
    x = torch.randn(n, 256, 1, 64, 64, device="cuda")
    w = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
    b = torch.zeros(256, device="cuda")
    wt = ops.weight_pack(w, 0)
    xa = ops.absmax(x)
    planes = n * 256
    y = torch.empty(n, 256, 64, 64, device="cuda")
    mean, rstd = torch.empty(planes, device="cuda"), torch.empty(planes, device="cuda")
    slot = torch.zeros(64, device="cuda")

    def conv():
        return ops.conv_raw(x, wt, b, 256, (1, 3, 3), 1, (0, 1, 1), 1, 1, 0, 0.0, (1, 64, 64), xa)

    def full():
        c1 = conv()
        check(lib().dfmir_instnorm_fwd(_p(c1), None, _p(y), _p(mean), _p(rstd), planes, 4096, 1e-5, 1, _p(slot), _st()))

    def stats():
        c1 = conv()
        check(lib().dfmir_instnorm_stats(_p(c1), _p(mean), _p(rstd), planes, 4096, 1e-5, 1, _p(slot), _st()))

    res = {}
    for rep in range(2):
        for name, fn in (("conv", conv), ("conv+IN", full), ("conv+stats", stats)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            res[name] = s.elapsed_time(e) / 20
        print("n = %d: conv %.3f ms | + InstanceNorm + ReLU pass %.1f us | + statistics-only pass %.1f us" % (
            n, res["conv"], 1e3 * (res["conv+IN"] - res["conv"]), 1e3 * (res["conv+stats"] - res["conv"])))
