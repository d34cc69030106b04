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
DIAG_HEADER = os.path.join(ROOT, "include", "randomfield_hip_diag.h")


def _declared_symbols(header=HEADER):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rf_[a-z0-9_]+)\s*\(", text)))


def _header_macro(name):
    m = re.search(r"#define\s+%s\s+(\d+)" % name, open(HEADER).read())
    return int(m.group(1))


def _have_gpu():
    try:
        return _hip.device_count() > 0
    except RuntimeError:
        return False


def test_library_is_built_in_tree():
    assert os.path.exists(_hip.LIB_PATH), "run `make -C randomfield_amd/csrc` (or __graft_entry__.build())"
    assert os.path.dirname(_hip.LIB_PATH) == os.path.dirname(randomfield_amd.__file__)


def test_exports_every_declared_symbol():
    """Both headers against the library's dynamic symbol table and against the binding: the consumer surface
    (randomfield_hip.h) and the diagnostics (randomfield_hip_diag.h) are disjoint, each is exported completely, the library
    exports no rf_ symbol that neither declares, and _hip.py binds exactly the two sets."""
    import subprocess
    lib = ctypes.CDLL(_hip.LIB_PATH)
    declared, diag = _declared_symbols(), _declared_symbols(DIAG_HEADER)
    assert len(declared) >= 30 and len(diag) >= 10
    assert not set(declared) & set(diag)
    for name in declared + diag:
        assert hasattr(lib, name), "library does not export %s" % name
    assert sorted(_hip.SIGNATURES) == declared
    assert sorted(_hip.DIAG_SIGNATURES) == diag
    nm = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = sorted(set(re.findall(r"\sT\s+(rf_[a-z0-9_]+)$", nm, flags=re.M)))
    assert exported == sorted(declared + diag), set(exported) ^ set(declared + diag)
    # ... and nothing else at all: the C++ helpers the translation units share (namespace rfc) are not an interface (rf_exports.map)
    every = [line.split()[-1] for line in nm.splitlines() if line.strip()]
    assert sorted(every) == exported, [n for n in every if n not in exported][:5]
    # the knobs and virtual-rank steps a consumer should not bind live in the diagnostics header only
    for name in ("rf_kernel_ms", "rf_set_merged_yz", "rf_merged_yz_ms", "rf_slab_exchange_local", "rf_slab_exchange_local_reverse",
                 "rf_mt_share_exchange_local", "rf_slab_forward", "rf_slab_backward", "rf_slab_set_exchange_standin"):
        assert name in diag and name not in declared


def test_version_and_error_string():
    lib = _hip.load()
    major, minor = _header_macro("RF_ABI_MAJOR"), _header_macro("RF_ABI_MINOR")
    assert lib.rf_version() == (major << 16) | minor           # the library was built from THIS header
    assert _hip.abi_version() == (major, minor) == (_hip.ABI_MAJOR, _hip.ABI_MINOR)
    assert major >= 5                                          # (rounds 1-4 answered 1 whatever the surface was)
    # every feature bit the header names is set in this build and known to the binding
    text = open(HEADER).read()
    bits = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"RF_FEATURE_([A-Z0-9_]+)\s*=\s*1\s*<<\s*(\d+)", text)}
    assert len(bits) >= 12 and {k: 1 << v for k, v in bits.items()} == _hip.FEATURES
    assert lib.rf_abi_features() == sum(_hip.FEATURES.values())
    assert _hip.abi_features() == sorted(_hip.FEATURES)
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
    assert not _hip.shape_supported(16418, 16, 16) and not _hip.shape_supported(5, 6, 8) and not _hip.shape_supported(4, 6, 7)     # 16418 = 2 x 8209 (prime)
    # longer axes split into two factors that each fit one line (the four-step form, rf_generic.h generic_split)
    assert lib.rf_shape_supported(16384, 16, 16) == 2 and lib.rf_shape_supported(4, 6, 2 * 12000) == 2
    # the dtype-aware query is what rf_plan_create accepts: a complex128 line holds 4096 points, so 2 x 6000 x 8192 ... nz / 2 = 4096 fits, 8192-point y lines split
    assert lib.rf_shape_supported_dtype(8, 6000, 8192, _hip.RF_F32) == 2 and lib.rf_shape_supported_dtype(8, 8198, 8, _hip.RF_F64) == 0      # 8198 = 2 x 4099 (prime)
    assert lib.rf_shape_supported_dtype(4096, 4, 8, _hip.RF_F64) == 2 and lib.rf_shape_supported_dtype(1024, 1024, 1024, _hip.RF_F64) == 1
    assert lib.rf_shape_supported_dtype(16, 16, 16, 7) == 0
    assert _hip.shape_supported(4096, 16, 16, np.complex128) and _hip.shape_supported(8192, 16, 16, np.complex128) and not _hip.shape_supported(8198, 16, 16, np.complex128)
    assert _hip.shape_supported(8192, 16, 16, np.complex64)


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
