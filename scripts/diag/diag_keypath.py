import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.golden import common as C
from tests.test_gpu_models import _hip_model_from_oracle, _load
from tests.test_oracle_golden import make_step
DEV="cuda"
res = []
for mode in ("batched", "seq", "batched"):
    batched = mode == "batched"
    st, size, B = make_step()
    model, opt = _hip_model_from_oracle(st, size, B, 8)
    A0, B0 = C.image_pair(93, B, size, size)
    call = [0]
    base_forward = model.netF.forward
    def pinned(feats, num_patches=64, patch_ids=None, base_forward=base_forward, call=call):
        if patch_ids is None:
            patch_ids = [C.patch_ids(call[0], i, f.shape[2] * f.shape[3], 256).to(DEV) for i, f in enumerate(feats)]
            call[0] += 1
        return base_forward(feats, num_patches, patch_ids)
    model.netF.forward = pinned
    model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""] * B, "B_paths": [""] * B})
    _load(model.netF, st.netF)
    model.setup(opt); model.parallelize()
    if batched:
        del model.netF.forward
        model.patch_id_source = lambda sizes, n_sets, P: torch.stack(
            [torch.stack([C.patch_ids(2 + t, l, S, P) for t in range(n_sets)]) for l, S in enumerate(sizes)])
    A_, B_ = C.image_pair(100, B, size, size)
    model.set_input({"A": A_, "B": B_, "A_paths": [""] * B, "B_paths": [""] * B})
    model.optimize_parameters()
    g = {}
    for nm, net in (("G", model.netG), ("F", model.netF), ("R", model.netR)):
        for k, p in net.named_parameters():
            g[nm + "." + k] = p.grad.detach().clone()
    res.append(g)
for a, b, tag in ((res[0], res[1], "batched vs seq"), (res[0], res[2], "batched vs batched")):
    print(tag)
    rows = []
    for k in a:
        if k.startswith('G.') and k.endswith('.bias') and k != 'G.model.30.bias':
            continue
        d = float((a[k] - b[k]).norm()); n = float(b[k].norm())
        rows.append((d / (n + 1e-30), k, n))
    for r in sorted(rows, reverse=True)[:12]:
        print("  %.3e  %-40s |g| %.3e" % r)
