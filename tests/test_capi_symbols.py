"""The C-ABI library loads without a GPU and exports every symbol include/jtprop.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from junctiontree_amd import _capi, engine


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "jtprop.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jtp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 19
    handle = ctypes.CDLL(_capi.LIB_PATH)
    for name in names:
        assert hasattr(handle, name), "libjtprop.so does not export %s" % name
        assert name in _capi.SYMBOLS, "the ctypes binding does not cover %s" % name
    assert sorted(_capi.SYMBOLS) == names


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_capi.TreeDesc) == 152 and ctypes.sizeof(_capi.Stats) == 760
    lib = _capi.lib()
    assert b"gfx950" in lib.jtp_version()
    assert lib.jtp_kernel_name(0) == b"jt_collect<T, 0>"
    assert lib.jtp_kernel_name(_capi.N_VARIANTS) is None


def test_plan_only_needs_no_gpu_and_device_plans_fail_loudly_without_one():
    plan = engine.Plan([0, (2, [1])], [[1, 2], [2, 3], [2]], {1: 2, 2: 3, 3: 2}, plan_only=True)
    d = plan.describe()
    assert d["n_cliques"] == 2 and d["n_messages"] == 2
    with pytest.raises(_capi.JtpError):                      # no device work on a host-only plan
        plan.propagate()
    plan.close()
    if _capi.device_count() == 0:
        with pytest.raises(_capi.JtpError, match="no CPU fallback"):
            engine.Plan([0, (2, [1])], [[1, 2], [2, 3], [2]], {1: 2, 2: 3, 3: 2})


def test_package_version_is_the_librarys():
    """`junctiontree_amd.__version__` is read from `jtp_version()` (round 6: it said 0.1.0 beside a 0.5.0 library)."""
    import junctiontree_amd as jt
    from junctiontree_amd import _capi
    text = _capi.lib().jtp_version().decode()
    assert text.startswith("jtprop ") and jt.__version__ == text.split()[1]
