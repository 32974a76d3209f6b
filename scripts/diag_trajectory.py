"""GPU box: the twelve-step trajectory of tests/test_gpu_models.py::test_twelve_step_trajectory_vs_oracle, losses of both sides
per step, eager and captured (python scripts/diag_trajectory.py [capture=0|1] [steps])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dfmir_oracle as O
from tests.golden import common as C
from tests.test_gpu_models import PinnedIds, _hip_model_from_oracle, _load, nce_sizes
capture = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
size, B = 64, 2
torch.manual_seed(11)
st = O.RegistrationStep(size, B, ngf=8)
with torch.no_grad():
    st.netR.flow.weight.mul_(float(os.environ.get("FLOWMUL", "1e5")))
st.ids_hook = lambda c, feats: [C.patch_ids(c, i, f.shape[2] * f.shape[3], 256) for i, f in enumerate(feats)]
A0, B0 = C.image_pair(7, B, size, size)
st.data_dependent_initialize(A0, B0)
with torch.no_grad():
    for p in st.netF.parameters():
        if p.dim() == 1:
            p.add_(0.01)
model, opt = _hip_model_from_oracle(st, size, B, 8)
opt.capture_step = bool(capture)
src = model.patch_id_source = PinnedIds()
model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": [""] * B, "B_paths": [""] * B})
_load(model.netF, st.netF)
model.setup(opt)
model.parallelize()
for it in range(steps):
    A_, B_ = C.image_pair(300 + 2 * it, B, size, size)
    ref = st.step(A_, B_)
    src.prefill(nce_sizes(size), 3, opt.num_patches)
    model.set_input({"A": A_, "B": B_, "A_paths": [""] * B, "B_paths": [""] * B})
    model.optimize_parameters()
    got = model.get_current_losses()
    fl = float(model.flow.abs().max()) if hasattr(model, "flow") else -1
    print("step %2d " % it + "  ".join("%s %.6f/%.6f" % (k, got[k], v) for k, v in ref.items()) + "  |flow|max %.2f  oracle flow max %.2f" % (fl, float(st.flow.abs().max()) if hasattr(st, "flow") else -1))
