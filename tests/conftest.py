import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(REPO, "tests", "golden")

    def load(name):
        return np.load(os.path.join(d, name), allow_pickle=False)
    return load


def pytest_sessionfinish(session, exitstatus):
    """DFMIR_MARGINS_OUT=<file>: every norm-wise comparison of the GPU tests (tests/test_gpu_ops.py::close) as
    `measured / bound` -- the table the tolerances are set from (profiles/rNN_parity_margins.txt)."""
    out = os.environ.get("DFMIR_MARGINS_OUT")
    mod = sys.modules.get("tests.test_gpu_ops")
    if not out or mod is None or not getattr(mod, "MARGINS", None):
        return
    worst = {}
    for test, what, err, bound in mod.MARGINS:
        bound = max(bound, 1e-300)                       # (bit-exact comparisons: rtol = atol = 0)
        k = (test.split("[")[0], what)
        if k not in worst or err / bound > worst[k][0] / worst[k][1]:
            worst[k] = (err, bound, test)
    with open(out, "w") as f:
        f.write("# worst measured relative error / its bound, per (test, tensor), over %d close() calls\n" % len(mod.MARGINS))
        for (test, what), (err, bound, full) in sorted(worst.items(), key=lambda kv: -kv[1][0] / kv[1][1]):
            f.write("%6.3f  %.3e / %.3e  %s :: %s\n" % (err / bound, err, bound, full, what))
