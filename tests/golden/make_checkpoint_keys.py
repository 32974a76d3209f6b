"""Generate tests/golden/checkpoint_keys.json by letting the REFERENCE write its own checkpoints (row N2).

Runs ONLY in the build container (imports /root/reference through the shims of make_golden.py).  The reference's
REGISTRATIONModel is built at config-1 geometry (64x64, batch 2, ngf 8) and at the full generator width (ngf 64), run
through data_dependent_initialize (netF's MLPs exist only from then on, registration_model.py:131-136), and its own
`save_networks('latest')` (models/base_model.py:164-180) writes `latest_net_{G,F,R}.pth`; the fixture records, per file,
the ordered key list with each tensor's shape and dtype -- names and geometry only, no weights, no reference source.

    python tests/golden/make_checkpoint_keys.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import make_golden as MG  # noqa: E402
from tests.golden import common as C  # noqa: E402


def main():
    MG.install_shims()
    import models.registration_model as RM
    out = {}
    for ngf in (8, 64):
        size, B = 64, 2
        opt = MG.ref_options(size, B, ngf)
        orig_open = RM.open_image_to_torch
        RM.open_image_to_torch = lambda path, sz: orig_open(path, sz)[:, :, :size, :size].expand(B, -1, -1, -1)
        torch.manual_seed(3)
        model = RM.REGISTRATIONModel(opt)
        A0, B0 = C.image_pair(93, B, size, size)
        model.data_dependent_initialize({"A": A0, "B": B0, "A_paths": ["a"] * B, "B_paths": ["b"] * B})
        model.save_networks('latest')
        RM.open_image_to_torch = orig_open
        entry = {}
        for nm in ("G", "F", "R"):
            path = os.path.join(opt.checkpoints_dir, opt.name, "latest_net_%s.pth" % nm)
            sd = torch.load(path, map_location="cpu")
            entry[nm] = [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]
        out["size%d_ngf%d" % (size, ngf)] = entry
    path = os.path.join(HERE, "checkpoint_keys.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: {n: len(v) for n, v in e.items()} for k, e in out.items()})


if __name__ == "__main__":
    main()
