"""Generator of tests/golden/edges.npz -- the option / argument branches INSIDE hot-path rows A8, A9 and A13 that the
main fixtures (make_golden.py) do not reach.  Runs only where /root/reference exists; imports the reference itself with
the shims of make_golden.py and records INPUTS-BY-SEED + the reference's own outputs:

  E1  util.losses.NCC_Loss.forward(prediction, target, mask)        (util/losses.py:257-261), 2-D and 3-D, and the empty mask
  E2  torchvoxelmorph.losses.NCC(win).loss = -mean(cc)                (models/voxelmorph/torchvoxelmorph/losses.py:7-67;
      its hard-coded `.to("cuda")` at :29 is redirected to the CPU for the duration of the call)
  E3  torchvoxelmorph.losses.Grad('l1' | 'l2', loss_mult).loss        (:93-117)
  E4  util.losses.Grad_Loss(penalty='l1'), with and without `mask=`   (util/losses.py:81-130)
  E5  PatchNCELoss with nce_includes_all_negatives_from_minibatch     (models/patchnce.py:32-38)
  E6  PatchSampleF(use_mlp=False), i.e. --netF sample                 (models/networks.py:280-281,602-619), and what the
      reference's REGISTRATIONModel does with it (data_dependent_initialize builds Adam over no parameters)

    python tests/golden/make_golden_edges.py
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from tests.golden import common as C                      # noqa: E402
from tests.golden import make_golden as MG                # noqa: E402


def main():
    MG.install_shims()
    import models.networks as RN
    from models.patchnce import PatchNCELoss as RefNCE
    from models.voxelmorph.torchvoxelmorph import losses as RVL
    from util.losses import Grad_Loss as RefGrad
    from util.losses import NCC_Loss as RefNCC
    npy = MG.npy
    out = {}
    d = C.edge_inputs()

    # ---- E1 masked NCC_Loss
    for tag, kv in (("2d", [9, 9]), ("3d", [9, 9, 9])):
        I = d["I" + tag].clone().requires_grad_()
        l = RefNCC('cpu', kernel_var=kv, kernel_type='mean')(I, d["J" + tag], mask=d["mask" + tag])
        l.backward()
        out.update({"ncc_masked_" + tag: npy(l), "dncc_masked_" + tag: npy(I.grad)})
    l = RefNCC('cpu', kernel_var=[9, 9], kernel_type='mean')(d["I2d"], d["J2d"], mask=torch.zeros_like(d["I2d"], dtype=torch.bool))
    out["ncc_empty_mask"] = np.array(float(l))

    # ---- E2 vxm NCC: -mean(cc); `.to("cuda")` -> CPU while it runs
    real_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a)
        return real_to(self, *a, **k)

    torch.Tensor.to = to_cpu
    try:
        for tag, win in (("2d", [5, 5]), ("3d", None)):
            yp = d["I" + tag].clone().requires_grad_()
            l = RVL.NCC(win).loss(d["J" + tag], yp)
            l.backward()
            out.update({"vxm_ncc_" + tag: npy(l), "dvxm_ncc_" + tag: npy(yp.grad)})
    finally:
        torch.Tensor.to = real_to

    # ---- E3 vxm Grad
    for tag, pen, mult in (("l1", 'l1', None), ("l2m", 'l2', 2.5)):
        f = d["field3"].clone().requires_grad_()
        l = RVL.Grad(pen, loss_mult=mult).loss(None, f)
        l.backward()
        out.update({"vxm_grad_" + tag: npy(l), "dvxm_grad_" + tag: npy(f.grad)})

    # ---- E4 Grad_Loss l1 / masked
    f = d["field3"].clone().requires_grad_()
    l = RefGrad(dim=3, penalty='l1')(f)
    l.backward()
    out.update(grad3d_l1=npy(l), dgrad3d_l1=npy(f.grad))
    f = d["field2"].clone().requires_grad_()
    l = RefGrad(dim=2, penalty='l1', loss_mult=0.5)(f, mask=d["fmask2"])
    l.backward()
    out.update(grad2d_l1_masked=npy(l), dgrad2d_l1_masked=npy(f.grad))
    f = d["field2"].clone().requires_grad_()
    l = RefGrad(dim=2, penalty='l2')(f, mask=d["fmask2"])
    l.backward()
    out.update(grad2d_l2_masked=npy(l), dgrad2d_l2_masked=npy(f.grad))

    # ---- E5 / E6: PatchSampleF without the MLP (--netF sample) feeding PatchNCELoss with all negatives of the minibatch
    feats = C.edge_sample_feats()
    rpf = RN.PatchSampleF(use_mlp=False, init_type='xavier', init_gain=0.02, nc=32, gpu_ids=[])
    assert len(list(rpf.parameters())) == 0
    ids = [C.patch_ids(40, i, f.shape[2] * f.shape[3], 48) for i, f in enumerate(feats)]
    fq = [f.clone().requires_grad_() for f in feats]
    fk = [C.randn(165 + i, *f.shape) for i, f in enumerate(feats)]
    kpool, _ = rpf(fk, 48, ids)
    qpool, _ = rpf(fq, 48, ids)
    for name, allneg in (("all", True), ("own", False)):
        crit = RefNCE(argparse.Namespace(nce_includes_all_negatives_from_minibatch=allneg, batch_size=2, nce_T=0.07))
        for f in fq:
            f.grad = None
        tot = 0
        for i, (q, k) in enumerate(zip(qpool, kpool)):
            l = crit(q, k)
            out["sample_%s_loss%d" % (name, i)] = npy(l)
            if name == "all":
                out["sample_q%d" % i] = npy(q)
            tot = tot + l.mean()
        tot.backward(retain_graph=True)
        for i, f in enumerate(fq):
            out["sample_%s_dfeat%d" % (name, i)] = npy(f.grad)

    # what the reference's model does with --netF sample: torch.optim.Adam over PatchSampleF's (empty) parameter list
    try:
        torch.optim.Adam(rpf.parameters(), lr=2e-4, betas=(0.5, 0.999))     # registration_model.py:134-135
        msg = ""
    except Exception as exc:                                                # noqa: BLE001
        msg = "%s: %s" % (type(exc).__name__, exc)
    out["netF_sample_optimizer_error"] = np.array(msg)
    MG.HERE = HERE
    MG.save("edges.npz", **out)
    print("netF=sample in data_dependent_initialize ->", msg)


if __name__ == "__main__":
    main()
