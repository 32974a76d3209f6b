"""the reference's own plugin loader finds the HIP class when INTEGRATION.md section 1 is
applied verbatim (models/__init__.py:25-67, options/base_options.py:75-110).  CPU; needs /root/reference (skipped on
the GPU box, where it does not exist)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_reference_loader_discovers_the_hip_plugin():
    env = dict(os.environ, PYTHONPATH=REPO, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dropin_probe.py")], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DROPIN_RESULT ")][-1]
    out = json.loads(line[len("DROPIN_RESULT "):])
    assert out["found_hip_class"], "models.find_model_using_name('registration') did not return dfmir_amd's class"
    assert out["option_setter_is_hip"]
    assert out["opt_model"] == "registration"
    # the model's option setter ran inside the reference's two-pass parse: CUT defaults (registration_model.py:61-62)
    assert out["nce_idt"] is True and out["lambda_NCE"] == 0.25 and out["pool_size"] == 0
    assert out["missing_fields"] == [], "opt fields read without a getattr default: %s" % out["missing_fields"]
    for f in ("nce_layers", "ngf", "crop_size", "lr", "beta1", "batch_size", "nce_T", "lr_policy", "direction"):
        assert f in out["fields_read"]
    for f in ("define_G", "define_F", "SpatialTransformer", "VxmDense"):
        assert f in out["factories"]
