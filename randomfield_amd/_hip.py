"""
ctypes binding of ``librandomfield_hip.so`` (the C-ABI declared in
``include/randomfield_hip.h``).

There is NO CPU fallback here: if the shared library has not been built, cannot
be loaded, or no GPU is visible, every entry point raises ``RuntimeError`` with
the reason.  Build the library in-tree with ``make -C randomfield_amd/csrc`` (or
``python -c "import __graft_entry__ as g; g.build()"``).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librandomfield_hip.so")

RF_F32, RF_F64 = 0, 1
DIRECT_RECORD_BYTES = 192                 # RF_DIRECT_RECORD_BYTES (randomfield_hip_diag.h)
NOISE_NATIVE, NOISE_EXTERNAL, NOISE_RESIDENT = 0, 1, 2
LAYOUT_DENSE, LAYOUT_PADDED = 0, 1

_c_void_pp = ctypes.POINTER(ctypes.c_void_p)
_c_dp = ctypes.POINTER(ctypes.c_double)

# the ABI this binding was written against (include/randomfield_hip.h RF_ABI_MAJOR / RF_ABI_MINOR): load() refuses a library of
# another major version or an older minor one
ABI_MAJOR, ABI_MINOR = 5, 3
FEATURES = {"realise": 1 << 0, "r2c": 1 << 1, "c2c": 1 << 2, "lognormal": 1 << 3, "potential": 1 << 4, "lensing": 1 << 5,
            "mt19937": 1 << 6, "mt19937_shared": 1 << 7, "multi_rank": 1 << 8, "generic_shapes": 1 << 9, "exchange_chunks": 1 << 10,
            "diagnostics": 1 << 11, "direct_exchange": 1 << 12}

# name -> (restype, argtypes); every symbol of include/randomfield_hip.h (the consumer surface) ...
SIGNATURES = {
    "rf_version": (ctypes.c_int, []),
    "rf_abi_features": (ctypes.c_uint, []),
    "rf_shape_supported_dtype": (ctypes.c_int, [ctypes.c_int] * 4),
    "rf_last_error": (ctypes.c_char_p, []),
    "rf_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "rf_shape_supported": (ctypes.c_int, [ctypes.c_int] * 3),
    "rf_plan_create": (ctypes.c_int, [_c_void_pp] + [ctypes.c_int] * 7),
    "rf_plan_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_plan_nbytes": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]),
    "rf_plan_set_stream": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rf_plan_set_flag": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "rf_set_kgrid": (ctypes.c_int, [ctypes.c_void_p, _c_dp, _c_dp, _c_dp]),
    "rf_set_power": (ctypes.c_int, [ctypes.c_void_p, _c_dp, _c_dp, ctypes.c_int]),
    "rf_generate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, _c_dp]),
    "rf_mt_set_jump": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint16),
                                      ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "rf_noise_mt19937": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32),
                                        ctypes.POINTER(ctypes.c_ulonglong)]),
    "rf_noise_mt19937_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32),
                                           ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]),
    "rf_mt_share_segments": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int)] * 3),
    "rf_mt_share_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]),
    "rf_mt_share_gather": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    "rf_mt_share_pack": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    "rf_mt_share_exchange": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_mt_share_finish": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    "rf_can_regenerate_potential": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rf_realise_scaled_potential": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_double, _c_dp]),
    "rf_execute_c2r": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_execute_r2c": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_plan_create_c2c": (ctypes.c_int, [_c_void_pp] + [ctypes.c_int] * 5),
    "rf_upload_c": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rf_download_c": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rf_execute_c2c": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rf_realise": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, _c_dp]),
    "rf_realise_potential": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, _c_dp]),
    "rf_realise_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, _c_dp]),
    "rf_realise_batch_prepare": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rf_moments": (ctypes.c_int, [ctypes.c_void_p, _c_dp, _c_dp]),
    "rf_lognormal": (ctypes.c_int, [ctypes.c_void_p, _c_dp, _c_dp, ctypes.c_int, ctypes.c_double]),
    "rf_scale_z": (ctypes.c_int, [ctypes.c_void_p, _c_dp, ctypes.c_int]),
    "rf_affine_z": (ctypes.c_int, [ctypes.c_void_p, _c_dp, ctypes.c_int, ctypes.c_double]),
    "rf_save_potential": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_load_potential": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
    "rf_lensing_potential": (ctypes.c_int, [ctypes.c_void_p, _c_dp, ctypes.c_int, ctypes.c_double, ctypes.c_int]),
    "rf_download_aux": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "rf_upload_k": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rf_download_k": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rf_upload_real": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rf_download_real": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "rf_device_ptr": (ctypes.c_int, [ctypes.c_void_p, _c_void_pp, _c_void_pp]),
    "rf_set_host_sink": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rf_host_sink_delivered": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "rf_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_elapsed_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "rf_set_z_tables": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
    "rf_realise_lognormal": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double)]),
    "rf_realise_batch_reference": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "rf_can_batch_reference": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_comm_init": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "rf_comm_size": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "rf_comm_allreduce_f64": (ctypes.c_int, [ctypes.c_void_p, _c_dp, ctypes.c_int, ctypes.c_int]),
    "rf_comm_enable_direct": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "rf_comm_direct_enabled": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
}
# ... and of include/randomfield_hip_diag.h (per-kernel timing, launch structure, virtual ranks: tests, bench.py, tools)
DIAG_SIGNATURES = {
    "rf_mt_share_exchange_local": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    "rf_download_noise": (ctypes.c_int, [ctypes.c_void_p, _c_dp, ctypes.c_ulonglong, ctypes.c_ulonglong]),
    "rf_kernel_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "rf_set_merged_yz": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rf_merged_yz_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]),
    "rf_yz_slabs": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "rf_slab_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, _c_dp]),
    "rf_slab_forward_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, _c_dp, ctypes.c_int]),
    "rf_slab_exchange_local": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    "rf_slab_backward": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_slab_r2c_rows": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_slab_r2c_cols": (ctypes.c_int, [ctypes.c_void_p]),
    "rf_slab_exchange_local_reverse": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    "rf_slab_stats": (ctypes.c_int, [ctypes.c_void_p, _c_dp, _c_dp]),
    "rf_slab_set_exchange_standin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "rf_slab_set_exchange_standin_ex": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "rf_slab_link_direct": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int]),
    "rf_slab_set_direct_standin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "rf_slab_direct_export": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    "rf_slab_direct_import": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
}

_lib = None


def load():
    """Load the shared library (once) and declare every prototype.  Raises
    RuntimeError if it is missing or does not load -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "randomfield_amd: HIP extension %s has not been built "
            "(run `make -C randomfield_amd/csrc`); there is no CPU fallback for the hip backend." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover - depends on the machine
        raise RuntimeError("randomfield_amd: cannot load %s: %s" % (LIB_PATH, exc))
    try:
        lib.rf_version.restype = ctypes.c_int
        ver = int(lib.rf_version())
    except AttributeError:
        raise RuntimeError("randomfield_amd: %s is not this package's library (no rf_version)" % LIB_PATH)
    if (ver >> 16) != ABI_MAJOR or (ver & 0xffff) < ABI_MINOR:
        raise RuntimeError("randomfield_amd: %s has ABI version %d.%d, this binding needs %d.%d or a later minor one: rebuild it "
                           "(`make -C randomfield_amd/csrc`)" % (LIB_PATH, ver >> 16, ver & 0xffff, ABI_MAJOR, ABI_MINOR))
    for table in (SIGNATURES, DIAG_SIGNATURES):
        for name, (restype, argtypes) in table.items():
            fn = getattr(lib, name)          # AttributeError here means header and library disagree
            fn.restype = restype
            fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    return load().rf_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("randomfield_amd HIP error%s: %s" % (" in " + what if what else "", last_error()))


def device_count():
    n = ctypes.c_int(0)
    check(load().rf_device_count(ctypes.byref(n)), "rf_device_count")
    return n.value


def require_gpu():
    """Raise RuntimeError unless the library loads and sees at least one GPU."""
    try:
        n = device_count()
    except RuntimeError:
        raise
    if n < 1:
        raise RuntimeError("randomfield_amd: no HIP device visible; the hip backend has no CPU fallback.")
    return n


def shape_supported(nx, ny, nz, dtype=None):
    """Does the HIP path take this grid?  With a dtype (numpy complex64 / complex128, or RF_F32 / RF_F64) the answer is exactly what
    rf_plan_create accepts for it on one rank (generic axes: one LDS line of up to 8192 complex64 / 4096 complex128 points, or two
    factors that fit)."""
    if dtype is None:
        return bool(load().rf_shape_supported(int(nx), int(ny), int(nz)))
    code = dtype if dtype in (RF_F32, RF_F64) else (RF_F64 if np.dtype(dtype) in (np.dtype(np.complex128), np.dtype(np.float64)) else RF_F32)
    return bool(load().rf_shape_supported_dtype(int(nx), int(ny), int(nz), int(code)))


def abi_version():
    """(major, minor) of the loaded library (rf_version)."""
    v = int(load().rf_version())
    return v >> 16, v & 0xffff


def abi_features():
    """Names of the entry-point groups the loaded library exports (rf_abi_features)."""
    bits = int(load().rf_abi_features())
    return sorted(k for k, b in FEATURES.items() if bits & b)


def _dp(a):
    return a.ctypes.data_as(_c_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DevicePlan(object):
    """Thin object wrapper around ``rf_plan*``.  All methods raise RuntimeError on failure."""

    def __init__(self, nx, ny, nz, dtype=np.complex64, device=0, nranks=1, rank=0, unpacked=False):
        """``unpacked=True`` makes a complex-to-complex plan over the full (nx, ny, nz) complex array
        (``transform.Plan(packed=False)``): only ``upload_c`` / ``execute_c2c`` / ``download_c`` apply."""
        self._lib = load()
        require_gpu()
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.complex64), np.dtype(np.complex128)):
            raise ValueError("DevicePlan dtype must be complex64 or complex128: %r" % (dtype,))
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.nranks, self.rank = int(nranks), int(rank)
        self.nx_local = self.nx // self.nranks          # x planes of the real-space field held by this rank
        # power-of-two axes run on the tiled kernels (rf_shape_supported == 1); any other even shape on the generic ones
        self.tiled = int(load().rf_shape_supported(self.nx, self.ny, self.nz)) == 1
        self.complex_dtype = dtype
        self.real_dtype = np.dtype(np.float32 if dtype == np.complex64 else np.float64)
        self._h = ctypes.c_void_p()
        self.unpacked = bool(unpacked)
        # what a result that is formed again on demand (generate._RegeneratedPotential) depends on: bumped whenever the resident
        # deviates / the power and k tables of this device plan change, whoever changes them
        self.noise_epoch = 0
        self.power_epoch = 0
        if self.unpacked:
            if nranks != 1:
                raise ValueError("unpacked c2c plans are single-GPU")
            check(self._lib.rf_plan_create_c2c(ctypes.byref(self._h), self.nx, self.ny, self.nz,
                                               RF_F64 if dtype == np.complex128 else RF_F32, int(device)),
                  "rf_plan_create_c2c")
            return
        check(self._lib.rf_plan_create(ctypes.byref(self._h), self.nx, self.ny, self.nz,
                                       RF_F64 if dtype == np.complex128 else RF_F32, int(device),
                                       int(nranks), int(rank)), "rf_plan_create")

    # -- unpacked complex-to-complex plans -----------------------------------
    def upload_c(self, data):
        data = np.ascontiguousarray(data, self.complex_dtype)
        if data.shape != (self.nx, self.ny, self.nz):
            raise ValueError("expected a complex array of shape %r" % ((self.nx, self.ny, self.nz),))
        check(self._lib.rf_upload_c(self._h, data.ctypes.data_as(ctypes.c_void_p)), "rf_upload_c")

    def download_c(self, out=None):
        if out is None:
            out = np.empty((self.nx, self.ny, self.nz), self.complex_dtype)
        if out.shape != (self.nx, self.ny, self.nz) or out.dtype != self.complex_dtype or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous %s array of shape %r" % (self.complex_dtype, (self.nx, self.ny, self.nz)))
        check(self._lib.rf_download_c(self._h, out.ctypes.data_as(ctypes.c_void_p)), "rf_download_c")
        return out

    def execute_c2c(self, inverse):
        """In place: ``inverse=True`` is np.fft.ifftn (1/N), ``False`` np.fft.fftn."""
        check(self._lib.rf_execute_c2c(self._h, 1 if inverse else -1), "rf_execute_c2c")

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.rf_plan_destroy(self._h)
            self._h = ctypes.c_void_p()
        self._sink_keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nbytes(self):
        n = ctypes.c_size_t(0)
        check(self._lib.rf_plan_nbytes(self._h, ctypes.byref(n)), "rf_plan_nbytes")
        return n.value

    def set_exact_generation(self, on=True):
        """Native-noise float32 realisations with the reference's exact float64 chain (slower)."""
        check(self._lib.rf_plan_set_flag(self._h, 1, int(bool(on))), "rf_plan_set_flag")

    def set_replicated_generation(self, on=True):
        """Multi-rank plans: no all-to-all; every rank generates all of k space and keeps its x slab (native rng)."""
        check(self._lib.rf_plan_set_flag(self._h, 4, int(bool(on))), "rf_plan_set_flag")

    def set_force_slab_path(self, on=True):
        """Route this single-rank plan through the multi-GPU slab pipeline (test hook)."""
        check(self._lib.rf_plan_set_flag(self._h, 2, int(bool(on))), "rf_plan_set_flag")

    def set_transposed_intermediate(self, on=True):
        """Default OFF (one in-place buffer).  On: the x pass stores contiguous chunks into a blocked scratch array of the field's
        size, the y pass runs in place there and the z pass gathers from it into the field buffer (RF_FLAG_TRANSPOSED_INTERMEDIATE:
        5 % faster at 2048^3 float32, slower at 1024^3, twice the device memory)."""
        check(self._lib.rf_plan_set_flag(self._h, 8, int(bool(on))), "rf_plan_set_flag")

    def set_yz_slab_planes(self, planes=-1):
        """x planes per slab of the y / z passes (single-GPU plans): -1 automatic (about the Infinity Cache's size), 0 = whole grid."""
        check(self._lib.rf_plan_set_flag(self._h, 16, int(planes)), "rf_plan_set_flag")

    def set_merged_yz(self, mode=1):
        """The z pass of slab s and the y pass of slab s + 1 in one launch (rf_k_yz.hip): 0 never, 1 untimed calls (default),
        2 timed calls too -- :meth:`merged_yz_ms` then gives the merged launches' average duration."""
        check(self._lib.rf_set_merged_yz(self._h, int(mode)), "rf_set_merged_yz")

    def merged_yz_ms(self):
        """(summed duration in ms, number) of the merged launches of the last timed call under ``set_merged_yz(2)``."""
        ms, n = ctypes.c_float(), ctypes.c_int()
        check(self._lib.rf_merged_yz_ms(self._h, ctypes.byref(ms), ctypes.byref(n)), "rf_merged_yz_ms")
        return float(ms.value), int(n.value)

    def set_exchange_chunks(self, chunks=1):
        """Multi-rank plans: generate / transform / send the rank's kz slab as ``chunks`` sub-slabs (a power of two), the exchange
        of each overlapped with the forward passes of the next inside one realisation (RF_FLAG_EXCHANGE_CHUNKS)."""
        check(self._lib.rf_plan_set_flag(self._h, 32, int(chunks)), "rf_plan_set_flag")

    def set_exchange_standin(self, workgroups):
        """Diagnostics (virtual rank of a multi-rank plan, no communicator): realise / realise_batch run the multi-GPU schedule with
        the all-to-all replaced by a copy kernel of ``workgroups`` workgroups (rf_slab_set_exchange_standin); 0 = off."""
        check(self._lib.rf_slab_set_exchange_standin(self._h, int(workgroups)), "rf_slab_set_exchange_standin")

    def set_exchange_standin_ex(self, workgroups, read_percent=100, write_percent=100):
        """The stand-in with the two directions of its traffic taken apart (rf_slab_set_exchange_standin_ex)."""
        check(self._lib.rf_slab_set_exchange_standin_ex(self._h, int(workgroups), int(read_percent), int(write_percent)), "rf_slab_set_exchange_standin_ex")

    def set_stream(self, hip_stream):
        check(self._lib.rf_plan_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)), "rf_plan_set_stream")

    # -- inputs -----------------------------------------------------------
    def set_kgrid(self, kx2, ky2, kz2):
        kx2, ky2, kz2 = _f64(kx2), _f64(ky2), _f64(kz2)
        if (len(kx2), len(ky2), len(kz2)) != (self.nx, self.ny, self.nz // 2 + 1):
            raise ValueError("k-grid tables have the wrong lengths")
        check(self._lib.rf_set_kgrid(self._h, _dp(kx2), _dp(ky2), _dp(kz2)), "rf_set_kgrid")
        self._power_key = None
        self.power_epoch += 1

    def set_power(self, log10k, sigma, if_changed=False):
        """Upload the (log10 k, sigma) tables.  ``if_changed=True`` skips the upload (a stream sync and a table rebuild) when
        these are the tables this plan holds already -- whoever uploaded them: the key lives with the device plan, so a
        caller that changes the tables through ``plan.device`` is seen by every other user of the plan."""
        log10k, sigma = _f64(log10k), _f64(sigma)
        if log10k.shape != sigma.shape or log10k.ndim != 1:
            raise ValueError("log10k and sigma must be 1-D arrays of equal length")
        key = (log10k.tobytes(), sigma.tobytes())
        if if_changed and key == getattr(self, "_power_key", None):
            return False
        self._power_key = None
        self.power_epoch += 1
        check(self._lib.rf_set_power(self._h, _dp(log10k), _dp(sigma), len(log10k)), "rf_set_power")
        self._power_key = key
        return True

    # -- generation / transforms -----------------------------------------
    # -- the reference's noise stream generated on the GPU --------------------
    def reference_noise(self, seed, single=False):
        """Fill the device noise buffer with ``RandomState(seed).normal(size=2*M)`` (MT19937 + polar
        method replayed on the GPU) for any seed numpy's legacy seeding accepts: an integer < 2**32, an
        array of integers (``init_by_array``) or None.  Afterwards pass ``noise='resident'``.
        ``single=True`` (complex64 plans) keeps float32 copies of the deviates instead of the float64 values: the fused
        ``realise`` / ``realise_potential`` read half as many bytes; ``generate`` and ``download_noise`` need float64."""
        self._mt_prepare()
        from . import mt19937
        state = np.ascontiguousarray(mt19937.seed_state(seed), np.uint32)
        acc = ctypes.c_ulonglong(0)
        self.noise_epoch += 1
        check(self._lib.rf_noise_mt19937_ex(self._h, state.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                            ctypes.byref(acc), 1 if single else 0), "rf_noise_mt19937_ex")
        return acc.value

    # -- the same stream shared between the ranks of a kz-slab job (rf_mt_share_*) ---------------
    def share_segments(self):
        """(segments of the whole stream, this rank's first segment, this rank's number of segments)"""
        self._mt_prepare()
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(self._lib.rf_mt_share_segments(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)), "rf_mt_share_segments")
        return a.value, b.value, c.value

    def share_begin(self, seed, single=False):
        """Replay this rank's segments of ``RandomState(seed).normal``; returns their accepted-pair counts."""
        from . import mt19937
        _, _, count = self.share_segments()
        state = np.ascontiguousarray(mt19937.seed_state(seed), np.uint32)
        counts = np.zeros(count, np.uint64)
        self.noise_epoch += 1
        check(self._lib.rf_mt_share_begin(self._h, state.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 1 if single else 0,
                                          counts.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))), "rf_mt_share_begin")
        return counts

    def share_gather(self):
        nseg, _, _ = self.share_segments()
        counts = np.zeros(nseg, np.uint64)
        check(self._lib.rf_mt_share_gather(self._h, counts.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))), "rf_mt_share_gather")
        return counts

    def share_pack(self, counts_all):
        counts_all = np.ascontiguousarray(counts_all, np.uint64)
        nseg, _, _ = self.share_segments()
        if counts_all.shape != (nseg,):
            raise ValueError("counts_all must hold one count per segment of the stream (%d)" % nseg)
        check(self._lib.rf_mt_share_pack(self._h, counts_all.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))), "rf_mt_share_pack")

    def share_exchange(self):
        check(self._lib.rf_mt_share_exchange(self._h), "rf_mt_share_exchange")

    def share_finish(self):
        acc = ctypes.c_ulonglong(0)
        check(self._lib.rf_mt_share_finish(self._h, ctypes.byref(acc)), "rf_mt_share_finish")
        return acc.value

    def reference_noise_shared(self, seed, single=False):
        """Collective over the plan's communicator: ``RandomState(seed).normal(size=2*M)`` replayed ONCE by all ranks together
        (each rank 1/P of the stream, one all-to-all of deviates) instead of once per rank (:meth:`reference_noise`).
        Afterwards pass ``noise='resident'``."""
        self.share_begin(seed, single)
        self.share_pack(self.share_gather())
        self.share_exchange()
        return self.share_finish()

    @staticmethod
    def reference_noise_shared_local(plans, seed, single=False):
        """The same between virtual ranks on one device (tests): ``plans`` = ranks 0..n-1 of one n-rank job."""
        counts = np.concatenate([p.share_begin(seed, single) for p in plans])
        for p in plans:
            p.share_pack(counts)
        arr = (ctypes.c_void_p * len(plans))(*[p._h.value for p in plans])
        check(load().rf_mt_share_exchange_local(arr, len(plans)), "rf_mt_share_exchange_local")
        return [p.share_finish() for p in plans]

    def realise_batch_reference(self, seeds, want_rms=True):
        """``len(seeds)`` same-seed realisations back to back (random.py:24-28 per seed): the MT19937 replay of seed i + 1 runs
        on a second stream under the y / z passes of seed i (rf_realise_batch_reference).  complex64 plans, fast generation
        path.  The field of the last seed stays on the device; returns the rms of every field."""
        self._mt_prepare()
        from . import mt19937
        states = np.ascontiguousarray(np.stack([np.asarray(mt19937.seed_state(sd), np.uint32) for sd in seeds]), np.uint32)
        rms = np.empty(len(states), np.float64) if want_rms else None
        self.noise_epoch += 1
        check(self._lib.rf_realise_batch_reference(self._h, states.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), len(states),
                                                   _dp(rms) if want_rms else None), "rf_realise_batch_reference")
        return rms

    def can_batch_reference(self):
        """Does :meth:`realise_batch_reference` serve this plan (single-GPU complex64 plan on the fast generation path, tables
        set, segments long enough for float32 pairs)?  With one seed it is also the fastest route to ONE same-seed field."""
        if self.nranks != 1 or not self.tiled:
            return False
        self._mt_prepare()
        return bool(self._lib.rf_can_batch_reference(self._h))

    def set_mt_segment_blocks(self, blocks):
        """Override the replay's segment length (blocks of 624 words; default: :func:`mt19937.segment_blocks_for`).  Small
        grids have a single default segment; the shared replay needs at least one segment per rank (tests use short ones).
        Every rank of a job must choose the same value, before the first replay."""
        if getattr(self, "_mt_ready", False):
            raise ValueError("the jump table of this plan has been uploaded already")
        self._mt_bps = int(blocks)

    def _mt_prepare(self):
        """upload the jump table of the MT19937 replay for this grid (once per plan)"""
        from . import mt19937
        if not getattr(self, "_mt_ready", False):
            # segment length for this grid's stream (a whole multiple of the GPU's wave slots on large grids), the stages
            # of the radix-16 jump tree it needs, and their polynomials' set-bit positions
            bps = getattr(self, "_mt_bps", None) or mt19937.segment_blocks_for(self.nx * self.ny * (self.nz // 2 + 1))
            blocks = -(-4 * mt19937.attempts_needed(self.nx * self.ny * (self.nz // 2 + 1)) // 624)
            nseg, stages = -(-blocks // bps), 1
            while mt19937.TREE_RADIX ** stages < nseg:
                stages += 1
            polys = mt19937.tree_polynomials(stages, segment_blocks=bps)
            pos = [mt19937.set_bit_positions(p) for p in polys]
            stride = max(len(q) for q in pos)
            table = np.zeros((len(pos), stride), np.uint16)
            for i, q in enumerate(pos):
                table[i, :len(q)] = q
            npos = np.array([len(q) for q in pos], np.int32)
            check(self._lib.rf_mt_set_jump(self._h, len(pos), table.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                                           npos.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), stride,
                                           bps, mt19937.TREE_RADIX), "rf_mt_set_jump")
            self._mt_ready = True

    def lensing_potential(self, cot_z, spacing, i_min):
        """psi of the real field on the device into the auxiliary buffer (generate.py:352-416)."""
        cot_z = _f64(cot_z)
        check(self._lib.rf_lensing_potential(self._h, _dp(cot_z), len(cot_z), float(spacing), int(i_min)),
              "rf_lensing_potential")

    def download_aux(self, x0=0, x1=None, out=None):
        x1 = self.nx_local if x1 is None else x1
        if out is None:
            out = np.empty((x1 - x0, self.ny, self.nz), self.real_dtype)
        check(self._lib.rf_download_aux(self._h, out.ctypes.data_as(ctypes.c_void_p), int(x0), int(x1)),
              "rf_download_aux")
        return out

    def download_noise(self, first=0, count=None):
        """Resident deviates, 2 per cell of the (k_shape) side array, flattened."""
        total = 2 * int(np.prod(self.k_shape))
        count = total - first if count is None else count
        out = np.empty(count, np.float64)
        check(self._lib.rf_download_noise(self._h, _dp(out), int(first), int(count)), "rf_download_noise")
        return out

    def _noise_arg(self, noise):
        if isinstance(noise, str) and noise == "resident":
            return NOISE_RESIDENT, None, None
        if noise is None:
            return NOISE_NATIVE, None, None
        noise = _f64(noise).reshape(-1)
        if noise.size != 2 * self.nx * self.ny * (self.nz // 2 + 1):
            raise ValueError("noise must hold 2*nx*ny*(nz/2+1) float64 deviates")
        self.noise_epoch += 1                  # host deviates replace the resident ones
        return NOISE_EXTERNAL, _dp(noise), noise

    def generate(self, seed=0, noise=None):
        mode, ptr, keep = self._noise_arg(noise)
        check(self._lib.rf_generate(self._h, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), mode, ptr), "rf_generate")

    def realise(self, seed=0, noise=None):
        mode, ptr, keep = self._noise_arg(noise)
        check(self._lib.rf_realise(self._h, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), mode, ptr), "rf_realise")

    def realise_potential(self, seed=0, noise=None):
        """generate_delta_field(save_potential=True): the field, and delta(k)/k**2 in the potential buffer."""
        mode, ptr, keep = self._noise_arg(noise)
        check(self._lib.rf_realise_potential(self._h, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), mode, ptr),
              "rf_realise_potential")

    def can_regenerate_potential(self, noise=None):
        """May ``realise_scaled_potential`` stand in for a stored potential (fast generation pass; native generator or the
        replayed stream resident as float32 pairs)?"""
        if noise is not None and not (isinstance(noise, str) and noise == "resident"):
            return False
        return bool(self._lib.rf_can_regenerate_potential(self._h, NOISE_RESIDENT if noise is not None else NOISE_NATIVE))

    def realise_scaled_potential(self, seed=0, noise=None, scale=1.0, factor_z=None):
        """calculate_newtonian_potential without a stored potential: the inverse transform of ``scale * delta(k)/k**2`` with
        delta(k) regenerated as ``realise(seed, noise)`` generates it (rf_realise_scaled_potential); ``factor_z`` (nz values,
        optional): plane z times factor_z[z], applied in the z pass's store."""
        mode, ptr, keep = self._noise_arg(noise)
        fz = None
        if factor_z is not None:
            fz = _f64(factor_z)
            if fz.shape != (self.nz,):
                raise ValueError("factor_z must hold nz values")
        check(self._lib.rf_realise_scaled_potential(self._h, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), mode, float(scale),
                                                    _dp(fz) if fz is not None else None), "rf_realise_scaled_potential")

    def realise_batch(self, seeds, want_rms=True):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        rms = np.empty(len(seeds), np.float64) if want_rms else None
        check(self._lib.rf_realise_batch(self._h, seeds.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), len(seeds),
                                         _dp(rms) if want_rms else None), "rf_realise_batch")
        return rms

    def realise_batch_prepare(self, n):
        check(self._lib.rf_realise_batch_prepare(self._h, int(n)), "rf_realise_batch_prepare")

    def execute_c2r(self):
        check(self._lib.rf_execute_c2r(self._h), "rf_execute_c2r")

    def execute_r2c(self):
        check(self._lib.rf_execute_r2c(self._h), "rf_execute_r2c")

    def moments(self):
        m, s = ctypes.c_double(), ctypes.c_double()
        check(self._lib.rf_moments(self._h, ctypes.byref(m), ctypes.byref(s)), "rf_moments")
        return m.value, s.value

    def set_z_tables(self, growth_z, density_z=None):
        """growth(z) and, optionally, the mean-density factor (nz,) of fused lognormal realisations (rf_set_z_tables)."""
        g = _f64(np.broadcast_to(np.asarray(growth_z, np.float64), (self.nz,)))
        d = None if density_z is None else _f64(np.broadcast_to(np.asarray(density_z, np.float64), (self.nz,)))
        check(self._lib.rf_set_z_tables(self._h, _dp(g), _dp(d) if d is not None else None, self.nz), "rf_set_z_tables")

    def realise_lognormal(self, seed=0, noise=None, want_sigma=True):
        """Gaussian realisation + lognormal map (+ density factor) in one call of five sweeps; returns the Gaussian field's rms."""
        mode, ptr, keep = self._noise_arg(noise)
        sig = ctypes.c_double(0.0)
        check(self._lib.rf_realise_lognormal(self._h, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), mode, ptr,
                                             ctypes.byref(sig) if want_sigma else None), "rf_realise_lognormal")
        return sig.value if want_sigma else None

    def lognormal(self, a_z, b_z, sigma):
        a_z, b_z = _f64(a_z), _f64(b_z)
        check(self._lib.rf_lognormal(self._h, _dp(a_z), _dp(b_z), len(a_z), float(sigma)), "rf_lognormal")

    def scale_z(self, factor_z):
        f = _f64(factor_z)
        check(self._lib.rf_scale_z(self._h, _dp(f), len(f)), "rf_scale_z")

    def affine_z(self, mul_z, add):
        f = _f64(mul_z)
        check(self._lib.rf_affine_z(self._h, _dp(f), len(f), float(add)), "rf_affine_z")

    def save_potential(self):
        check(self._lib.rf_save_potential(self._h), "rf_save_potential")

    def load_potential(self, scale=1.0):
        check(self._lib.rf_load_potential(self._h, float(scale)), "rf_load_potential")

    # -- host <-> device --------------------------------------------------
    @property
    def k_shape(self):
        """Shape of this rank's k-space side arrays (k buffer, potential): (nx, ny, nz/2 + 1) on one GPU; a kz-slab rank
        holds its nz/(2 ranks) planes followed by the Nyquist plane (meaningful on rank 0, which packs it into kz = 0)."""
        return (self.nx, self.ny, self.nz // 2 // self.nranks + 1)

    def upload_k(self, data):
        if data.shape != self.k_shape or data.dtype != self.complex_dtype:
            raise ValueError("upload_k: wrong shape or dtype")
        data = np.ascontiguousarray(data)
        check(self._lib.rf_upload_k(self._h, data.ctypes.data_as(ctypes.c_void_p)), "rf_upload_k")

    def download_k(self, out=None):
        if out is None:
            out = np.empty(self.k_shape, self.complex_dtype)
        if not out.flags.c_contiguous or out.dtype != self.complex_dtype or out.shape != self.k_shape:
            raise ValueError("download_k: out must be a C-contiguous %r array of the plan's dtype" % (self.k_shape,))
        check(self._lib.rf_download_k(self._h, out.ctypes.data_as(ctypes.c_void_p)), "rf_download_k")
        return out

    def upload_real(self, data, padded=False):
        nzp = self.nz + 2 if padded else self.nz
        # (a multi-rank plan takes its own nx / ranks planes)
        if data.shape != (self.nx_local, self.ny, nzp) or data.dtype != self.real_dtype or not data.flags.c_contiguous:
            raise ValueError("upload_real: wrong shape, dtype or layout")
        check(self._lib.rf_upload_real(self._h, data.ctypes.data_as(ctypes.c_void_p),
                                       LAYOUT_PADDED if padded else LAYOUT_DENSE), "rf_upload_real")

    def download_real(self, out=None, padded=False, x0=0, x1=None):
        """Rows x0 <= ix < x1 (local plane indices on multi-GPU plans) of the real-space field."""
        x1 = self.nx_local if x1 is None else x1
        nzp = self.nz + 2 if padded else self.nz
        if out is None:
            out = np.empty((x1 - x0, self.ny, nzp), self.real_dtype)
        if out.shape != (x1 - x0, self.ny, nzp) or out.dtype != self.real_dtype or not out.flags.c_contiguous:
            raise ValueError("download_real: out has the wrong shape, dtype or layout")
        check(self._lib.rf_download_real(self._h, out.ctypes.data_as(ctypes.c_void_p),
                                         LAYOUT_PADDED if padded else LAYOUT_DENSE, int(x0), int(x1)),
              "rf_download_real")
        return out

    def arm_host_sink(self, out, padded=False):
        """The NEXT realisation of this (single-GPU, tiled) plan delivers its field into ``out`` -- (nx, ny, nz [+ 2 if padded]) of the
        plan's real dtype, C-contiguous -- slab by slab behind its z pass and returns when the field is there (rf_set_host_sink).
        Returns False when the plan cannot (multi-rank / generic plans): nothing is armed then.  ``host_sink_delivered()`` tells afterwards whether the call delivered."""
        nzp = self.nz + 2 if padded else self.nz
        if out.shape != (self.nx, self.ny, nzp) or out.dtype != self.real_dtype or not out.flags.c_contiguous:
            raise ValueError("arm_host_sink: out has the wrong shape, dtype or layout")
        if self.nranks != 1 or not self.tiled:
            return False
        rc = self._lib.rf_set_host_sink(self._h, out.ctypes.data_as(ctypes.c_void_p), LAYOUT_PADDED if padded else LAYOUT_DENSE)
        if rc == 0:
            self._sink_keepalive = out           # (the armed call writes into it: it must live until then)
        return rc == 0

    def host_sink_delivered(self):
        got = ctypes.c_int(0)
        check(self._lib.rf_host_sink_delivered(self._h, ctypes.byref(got)), "rf_host_sink_delivered")
        return bool(got.value)

    def device_ptrs(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        check(self._lib.rf_device_ptr(self._h, ctypes.byref(a), ctypes.byref(b)), "rf_device_ptr")
        return a.value, b.value

    # -- multi-GPU -------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL unique id (call on rank 0, hand to every rank's comm_init)."""
        buf = ctypes.create_string_buffer(128)
        check(load().rf_comm_unique_id(buf), "rf_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(self._lib.rf_comm_init(self._h, buf), "rf_comm_init")

    def comm_size(self):
        """Ranks of the plan's RCCL communicator as RCCL counts them (0: no communicator yet)."""
        n = ctypes.c_int(0)
        check(self._lib.rf_comm_size(self._h, ctypes.byref(n)), "rf_comm_size")
        return n.value

    def allreduce(self, values, op="sum"):
        """All-reduce 1 or 2 host doubles over the plan's RCCL communicator (also a barrier)."""
        a = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        check(self._lib.rf_comm_allreduce_f64(self._h, _dp(a), len(a), 0 if op == "sum" else 1), "rf_comm_allreduce_f64")
        return a

    def barrier(self):
        self.allreduce([0.0])

    def enable_direct_exchange(self, on=True):
        """COLLECTIVE over the plan's communicator (rf_comm_enable_direct): switch the exchange between the y and z passes to the
        direct form -- every rank's y pass stores its output tiles straight into the (IPC-mapped) receive buffers of the ranks
        that own their x planes, one tiny all-reduce per realisation as the barrier -- if EVERY rank can; returns whether the job
        now runs that way (False: all ranks keep the grouped ncclSend / ncclRecv exchange).  Same fields either way."""
        got = ctypes.c_int(0)
        check(self._lib.rf_comm_enable_direct(self._h, 1 if on else 0, ctypes.byref(got)), "rf_comm_enable_direct")
        return bool(got.value)

    def direct_exchange_enabled(self):
        got = ctypes.c_int(0)
        check(self._lib.rf_comm_direct_enabled(self._h, ctypes.byref(got)), "rf_comm_direct_enabled")
        return bool(got.value)

    @staticmethod
    def slab_link_direct(plans, on=True):
        """Diagnostics: the direct exchange between virtual ranks on one device (rf_slab_link_direct) -- ``slab_forward`` on every
        plan then stores into the others' receive buffers and ``slab_exchange_local`` is not called."""
        arr = (ctypes.c_void_p * len(plans))(*[p._h.value for p in plans])
        check(load().rf_slab_link_direct(arr, len(plans), 1 if on else 0), "rf_slab_link_direct")

    def slab_direct_export(self):
        """Diagnostics: this rank's record for the direct exchange between processes without a communicator (two IPC handles + a
        flag, rf_slab_direct_export) -- bytes to hand to every other rank by whatever transport the caller has."""
        buf = (ctypes.c_ubyte * DIRECT_RECORD_BYTES)()
        check(self._lib.rf_slab_direct_export(self._h, buf, DIRECT_RECORD_BYTES), "rf_slab_direct_export")
        return bytes(buf)

    def slab_direct_import(self, records):
        """Diagnostics: map the receive buffers named by the records of ALL ranks (rank order) and switch the storing y pass on
        (rf_slab_direct_import); False when some rank cannot take part.  The barrier between ``slab_forward`` and
        ``slab_backward`` is the caller's (``sync()`` + its own barrier)."""
        blob = b"".join(records)
        if len(blob) != DIRECT_RECORD_BYTES * len(records):
            raise ValueError("every record is {0} bytes".format(DIRECT_RECORD_BYTES))
        on = ctypes.c_int(0)
        check(self._lib.rf_slab_direct_import(self._h, blob, len(records), ctypes.byref(on)), "rf_slab_direct_import")
        return bool(on.value)

    def set_direct_standin(self, on=True, overlap=True):
        """Diagnostics: one virtual rank through the schedule of the direct exchange, its stores landing in its own receive
        buffers (rf_slab_set_direct_standin); not a field."""
        check(self._lib.rf_slab_set_direct_standin(self._h, 1 if on else 0, 1 if overlap else 0), "rf_slab_set_direct_standin")

    def slab_forward(self, seed=0, noise=None, source="generate"):
        """Forward half of the slab pipeline on this rank's kz planes.  ``source``: 'generate' (rows K..S fused into the
        x pass), 'potential' (the same and delta(k)/k**2 kept in the potential buffer) or 'kspace' (from the k buffer)."""
        mode, ptr, keep = self._noise_arg(noise)
        src = {"generate": 0, "potential": 1, "kspace": 2}[source]
        check(self._lib.rf_slab_forward_ex(self._h, ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), mode, ptr, src),
              "rf_slab_forward_ex")

    @staticmethod
    def slab_exchange_local(plans):
        arr = (ctypes.c_void_p * len(plans))(*[p._h.value for p in plans])
        check(load().rf_slab_exchange_local(arr, len(plans)), "rf_slab_exchange_local")

    def slab_backward(self):
        check(self._lib.rf_slab_backward(self._h), "rf_slab_backward")

    def slab_r2c_rows(self):
        """Multi-rank forward transform, first half: z pass on this rank's x slab, rows cut into the send blocks."""
        check(self._lib.rf_slab_r2c_rows(self._h), "rf_slab_r2c_rows")

    def slab_r2c_cols(self):
        """... second half: forward y and x passes on this rank's kz slab, result in the k-space side array."""
        check(self._lib.rf_slab_r2c_cols(self._h), "rf_slab_r2c_cols")

    @staticmethod
    def slab_exchange_local_reverse(plans):
        arr = (ctypes.c_void_p * len(plans))(*[p._h.value for p in plans])
        check(load().rf_slab_exchange_local_reverse(arr, len(plans)), "rf_slab_exchange_local_reverse")

    def slab_stats(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        check(self._lib.rf_slab_stats(self._h, ctypes.byref(a), ctypes.byref(b)), "rf_slab_stats")
        return a.value, b.value

    # -- sync / timing ----------------------------------------------------
    def sync(self):
        check(self._lib.rf_sync(self._h), "rf_sync")

    def elapsed_ms(self):
        ms = ctypes.c_float()
        check(self._lib.rf_elapsed_ms(self._h, ctypes.byref(ms)), "rf_elapsed_ms")
        return ms.value

    def yz_slabs(self):
        """(launches, x planes per launch) of the y and z passes (rf_yz_slabs)."""
        n, b = ctypes.c_int(), ctypes.c_int()
        check(self._lib.rf_yz_slabs(self._h, ctypes.byref(n), ctypes.byref(b)), "rf_yz_slabs")
        return n.value, b.value

    def kernel_ms(self):
        ms = (ctypes.c_float * 5)()      # x (main kernel), y, z, reduce, x kz=0 repair launch
        check(self._lib.rf_kernel_ms(self._h, ms), "rf_kernel_ms")
        return list(ms)
