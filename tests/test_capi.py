"""C-ABI checks that need no GPU: the shared library loads, exports exactly the
symbols include/randomfield_hip.h declares, and the product path fails LOUDLY
(no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import randomfield_amd
from randomfield_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "randomfield_hip.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", text)))


def _have_gpu():
    try:
        return _hip.device_count() > 0
    except RuntimeError:
        return False


def test_library_is_built_in_tree():
    assert os.path.exists(_hip.LIB_PATH), "run `make -C randomfield_amd/csrc` (or __graft_entry__.build())"
    assert os.path.dirname(_hip.LIB_PATH) == os.path.dirname(randomfield_amd.__file__)


def test_exports_every_declared_symbol():
    lib = ctypes.CDLL(_hip.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    # and the Python binding covers the same set
    assert sorted(_hip.SIGNATURES) == declared


def test_version_and_error_string():
    lib = _hip.load()
    assert lib.rf_version() >= 1
    assert isinstance(_hip.last_error(), str)


def test_shape_support_query_needs_no_gpu():
    assert _hip.shape_supported(1024, 1024, 1024) and _hip.shape_supported(16, 32, 64)
    assert _hip.shape_supported(2048, 2048, 2048) and _hip.shape_supported(8, 8, 16)
    lib = _hip.load()
    assert lib.rf_shape_supported(1024, 1024, 1024) == 1          # tiled power-of-two kernels
    # any other even shape up to 8192 per axis (4096 on complex128 plans: rf_plan_create checks the dtype): the generic mixed-radix
    # kernels (the reference's own test shapes; axes beyond the tiled kernels' 2048)
    assert lib.rf_shape_supported(4, 6, 8) == 2 and lib.rf_shape_supported(40, 60, 80) == 2
    assert lib.rf_shape_supported(16, 16, 18) == 2 and lib.rf_shape_supported(8, 8, 8) == 2
    assert lib.rf_shape_supported(4096, 16, 16) == 2 and lib.rf_shape_supported(8, 6000, 8192) == 2
    assert not _hip.shape_supported(16384, 16, 16) and not _hip.shape_supported(5, 6, 8) and not _hip.shape_supported(4, 6, 7)


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_hip_backend_fails_loudly_without_gpu():
    from randomfield_amd import Generator
    from randomfield_amd.transform import Plan
    with pytest.raises(RuntimeError):
        Plan(shape=(16, 16, 16), dtype_in=np.complex64)            # default backend is 'hip'
    with pytest.raises(RuntimeError):
        Generator(16, 16, 16, 2.5)
    with pytest.raises(RuntimeError):
        _hip.DevicePlan(16, 16, 16)


def test_product_never_imports_oracle():
    """The product package must not import, call or link anything under oracle/."""
    pkg = os.path.dirname(randomfield_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "cpu_ref" not in text and "import oracle" not in text and "from oracle" not in text, f
