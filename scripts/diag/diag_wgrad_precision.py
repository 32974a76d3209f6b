"""Diagnostic (GPU box): per-parameter error of the HIP step's gradients against an fp64 run of the oracle, beside the
fp32 CPU oracle's own error -- the table `test_full_size_gradients_vs_fp64_oracle` bounds -- under whatever DFMIR_*
switches the environment carries.  The two CPU runs are cached in a file so that several switch settings can be
compared in one gpurun call:

    python scripts/diag/diag_wgrad_precision.py /tmp/g.pt            # default path
    DFMIR_CONV_FP32=1 python scripts/diag/diag_wgrad_precision.py /tmp/g.pt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import _full_size_hip, _full_size_oracle

cache = sys.argv[1] if len(sys.argv) > 1 else "/tmp/diag_wgrad_precision.pt"
B = 1
st32, size, A0, B0 = _full_size_oracle(O, B)
A_, B_ = C.image_pair(11, B, size, size)
if os.path.exists(cache):
    ref = torch.load(cache)
else:
    st64, _, _, _ = _full_size_oracle(O, B, double=True)
    model_state = {t: {k: v.clone() for k, v in n.state_dict().items()} for t, n in
                   (("G", st32.netG), ("R", st32.netR), ("F", st32.netF))}
    st32.step(A_, B_)
    st64.step(A_.double(), B_.double())
    ref = {"g32": {}, "g64": {}, "state": model_state}
    for tag, n32, n64 in (("G", st32.netG, st64.netG), ("R", st32.netR, st64.netR), ("F", st32.netF, st64.netF)):
        for (k, p32), (_, p64) in zip(n32.named_parameters(), n64.named_parameters()):
            ref["g32"][tag + "." + k] = p32.grad.clone()
            ref["g64"][tag + "." + k] = p64.grad.clone()
    torch.save(ref, cache)
    # the HIP model below must start from the PRE-step weights
    for t, n in (("G", st32.netG), ("R", st32.netR), ("F", st32.netF)):
        n.load_state_dict(model_state[t])
model = _full_size_hip(st32, size, B, A0, B0)
model.set_input({"A": A_, "B": B_, "A_paths": [""], "B_paths": [""]})
model.optimize_parameters()
torch.cuda.synchronize()
sw = {k: v for k, v in os.environ.items() if k.startswith("DFMIR_")}
print("switches:", sw)
print("%-40s %10s %10s %7s" % ("parameter", "HIP", "fp32 CPU", "ratio"))
worst = 0.0
for tag, nh in (("G", model.netG), ("R", model.netR), ("F", model.netF)):
    for k, ph in nh.named_parameters():
        if tag == "G" and k.endswith(".bias") and k != "model.30.bias":
            continue
        name = tag + "." + k
        g64 = ref["g64"][name].flatten()
        e_cpu = float((ref["g32"][name].double().flatten() - g64).norm() / g64.norm())
        e_hip = float((ph.grad.detach().cpu().double().flatten() - g64).norm() / g64.norm())
        r = e_hip / (e_cpu + 1e-12)
        if tag != "F":
            worst = max(worst, (e_hip - 2e-5) / (e_cpu + 1e-12))
        if tag == "G" and k.endswith(".weight") or tag == "F":
            print("%-40s %10.2e %10.2e %7.2f" % (name, e_hip, e_cpu, r))
print("worst (HIP - 2e-5) / CPU over G, R: %.2f" % worst)
