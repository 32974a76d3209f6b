"""CPU: the C-ABI library loads and exports every symbol include/dfmir_hip.h declares, and the
ctypes table covers exactly that set (no compute calls here -- there is no GPU in this tier)."""
import ctypes
import os
import re

import dfmir_amd
from dfmir_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(REPO, "include", "dfmir_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfmir_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    syms = header_symbols()
    assert len(syms) >= 40
    h = ctypes.CDLL(dfmir_amd.LIB_PATH)
    for s in syms:
        assert hasattr(h, s), "libdfmir_hip.so does not export %s" % s


def test_ctypes_table_matches_header():
    assert header_symbols() == _lib.exported_symbols()


def test_abi_version_and_error_string():
    h = dfmir_amd.lib()
    assert h.dfmir_abi_version() == 14
    # a bad-argument call must fail loudly without touching a device
    rc = h.dfmir_scale(None, None, 0, 1.0, None)
    assert rc != 0
    assert b"invalid argument" in h.dfmir_last_error()


def test_geom_struct_layout():
    assert ctypes.sizeof(_lib.DfConvGeom) == 20 * 4


def test_options_table_is_settable_and_falls_back_to_the_environment(monkeypatch):
    """dfmir_set_option / dfmir_get_option (include/dfmir_hip.h "Options"): the one process-global table in front of
    the environment that every kernel-selection switch of the library reads."""
    name = "DFMIR_TEST_ONLY_OPTION"
    monkeypatch.delenv(name, raising=False)
    assert _lib.get_option(name) is None
    monkeypatch.setenv(name, "from-env")
    assert _lib.get_option(name) == "from-env"            # never set through the API: the environment answers
    _lib.set_option(name, "41")
    assert _lib.get_option(name) == "41"
    _lib.set_option(name, None)                           # explicitly unset hides the environment variable too
    assert _lib.get_option(name) is None
    h = dfmir_amd.lib()
    assert h.dfmir_set_option(b"NOT_OURS", b"1") != 0     # names outside the DFMIR_ namespace are refused
    assert b"invalid argument" in h.dfmir_last_error()


def test_library_is_loaded_after_pytorch_rocm():
    """dfmir_amd.lib() imports torch before it dlopens libdfmir_hip.so: PyTorch-ROCm carries its own HIP runtime, and a
    process that loaded this library first ended up with two runtimes -- the kernels here were then launched on one that had
    enumerated no device (found with `python __graft_entry__.py --smoke`, which builds and smokes in ONE process)."""
    import subprocess, sys, os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import dfmir_amd; assert 'torch' not in sys.modules; "
            "dfmir_amd.lib(); assert 'torch' in sys.modules; print('ok')" % repo)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
