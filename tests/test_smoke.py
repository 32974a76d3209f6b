"""The driver's end-to-end check, kept green by the GPU suite: __graft_entry__.smoke() wraps netF.forward with the
reference's own signature `(feats, num_patches=64, patch_ids=None)` (models/networks.py:602) and runs the DEFAULT
train step (stacked query passes, key-feature reuse) against the oracle."""
import inspect

import pytest


@pytest.mark.gpu
def test_driver_smoke():
    import __graft_entry__ as entry
    entry.smoke()


def test_patch_sampler_keeps_reference_signature():
    from dfmir_amd.networks import PatchSampleF
    sig = inspect.signature(PatchSampleF.forward)
    assert list(sig.parameters) == ["self", "feats", "num_patches", "patch_ids"]
    assert sig.parameters["num_patches"].default == 64 and sig.parameters["patch_ids"].default is None
