"""
FFT plans on a single buffer -- host-side mirror of ``randomfield/transform.py``
with an MI355X (HIP) backend underneath.

Same names, argument meaning and error behaviour as the reference module
(file:line citations are to the reference checkout):

* :func:`allocate` (transform.py:10-43), :func:`expanded_shape` (:46-60),
  :func:`scalar_type` / :func:`complex_type` (:63-90), :func:`is_hermitian`
  (:93-111), :func:`symmetrize` (:114-158), :class:`Plan` (:161-315).

Backends.  The reference switches between pyFFTW and ``numpy.fft`` on the
``use_pyfftw`` flag (transform.py:247-270).  Here the choice is

* ``backend='hip'`` (default; or ``RANDOMFIELD_BACKEND=hip``): the transform runs
  on the GPU through ``librandomfield_hip.so``.  If the library is not built, no
  GPU is visible or the shape is not supported, the constructor raises
  ``RuntimeError`` -- it never silently computes on the CPU.
* ``backend='numpy'`` (explicit opt-in; or ``RANDOMFIELD_BACKEND=numpy``): the
  reference's own ``numpy.fft`` fallback, kept for hosts without a GPU and for shapes
  the HIP kernels do not cover (an odd axis; an axis beyond 8192 points -- 4096 for complex128 -- with no two factors that fit).  Power-of-two axes run on
  the tiled kernels; any other even shape -- the reference's own test shapes (4, 6, 8)
  and (40, 60, 80) among them -- on the generic mixed-radix kernels.

``use_pyfftw`` is accepted for signature compatibility and recorded, but pyFFTW
is never used.
"""
from __future__ import annotations

import os

import numpy as np

__all__ = ["allocate", "expanded_shape", "scalar_type", "complex_type", "is_hermitian", "symmetrize", "Plan",
           "resolve_backend"]


def resolve_backend(backend=None):
    """'hip' or 'numpy' from the keyword, then $RANDOMFIELD_BACKEND, then 'hip'."""
    if backend is None:
        backend = os.environ.get("RANDOMFIELD_BACKEND", "hip")
    backend = str(backend).lower()
    if backend not in ("hip", "numpy"):
        raise ValueError("Invalid backend: {0} (expected 'hip' or 'numpy').".format(backend))
    return backend


def allocate(shape, dtype, use_pyfftw=True):
    """
    Allocate a contiguous block of un-initialized typed memory (C order).

    Mirrors transform.py:10-43.  ``use_pyfftw`` is accepted and ignored; numpy's
    allocator already returns 16-byte (in practice 64-byte) aligned blocks for
    arrays of this size, which is what the SIMD-aligned pyFFTW path provided.
    """
    return np.empty(shape, dtype, order="C")


def expanded_shape(data, packed=False):
    """Logical (nx, ny, nz) of a 3D array; a packed array stores nz/2 + 1 planes (transform.py:46-60)."""
    nx, ny, nz_stored = data.shape
    if nx % 2 or ny % 2:
        raise ValueError("First two dimensions of array must be even.")
    stored_is_odd = bool(nz_stored % 2)
    if stored_is_odd != bool(packed):
        raise ValueError("Last dimension of packed array must be odd." if packed
                         else "Last dimension of unpacked array must be even.")
    return nx, ny, (2 * (nz_stored - 1) if packed else nz_stored)


def _sctype(rep):
    """numpy-2 replacement for ``np.obj2sctype``: scalar type object or None."""
    if rep is None:
        return None
    try:
        return np.dtype(rep).type
    except TypeError:
        return None


_SCALAR_OF = {np.csingle: np.single, np.cdouble: np.double, np.clongdouble: np.longdouble}
_COMPLEX_OF = {np.single: np.csingle, np.double: np.cdouble, np.longdouble: np.clongdouble}


def scalar_type(complex_type):
    """Type of the real and imaginary parts of a complex type (transform.py:63-75)."""
    t = _sctype(complex_type)
    if t in _SCALAR_OF:
        return _SCALAR_OF[t]
    raise ValueError("Invalid complex_type: {0}.".format(t))


def complex_type(scalar_type):
    """Complex type corresponding to a component scalar type (transform.py:78-90)."""
    t = _sctype(scalar_type)
    if t in _COMPLEX_OF:
        return _COMPLEX_OF[t]
    raise ValueError("Invalid scalar_type: {0}.".format(t))


def is_hermitian(data, packed=False, rtol=1e-08, atol=1e-08):
    """
    Test if a 3D field is Hermitian (transform.py:93-111).

    Same coverage as the reference: ix <= nx/2, iy <= ny/2 and, for a packed
    array, only the planes kz = 0 and kz = nz/2.  Vectorised over the plane
    instead of the reference's triple python loop.
    """
    nx, ny, nz = expanded_shape(data, packed=packed)
    ix = np.arange(nx // 2 + 1)
    iy = np.arange(ny // 2 + 1)
    jx, jy = (nx - ix) % nx, (ny - iy) % ny
    z_range = [0, nz // 2] if packed else range(nz // 2 + 1)
    for iz in z_range:
        jz = (nz - iz) % nz
        a = data[: nx // 2 + 1, : ny // 2 + 1, iz]
        b = np.conj(data[jx][:, jy, jz])
        if not np.allclose(a, b, rtol, atol):
            return False
    return True


def symmetrize(data, packed=False):
    """
    Symmetrize a complex 3D field so that its inverse FFT is real valued
    (transform.py:114-158).

    Rule: with j = ((-ix) % nx, (-iy) % ny, (-iz) % nz) the
    *destination* cells take conj(value at j); the 8 self-conjugate vertices
    keep only their real part; finally the DC mode is zeroed.  In the z = 0 and
    z = nz/2 planes destinations are iy > ny/2, or iy in {0, ny/2} with
    ix > nx/2.  For an unpacked array the cells with 0 < iz < nz/2 are sources
    except the octant (ix > nx/2, iy > ny/2), and the cells with iz > nz/2 are
    destinations except the octant (0 < ix < nx/2, 0 < iy < ny/2)
    (transform.py:125-138).  Sources are never overwritten before they are read.
    """
    nx, ny, nz = expanded_shape(data, packed=packed)
    jx = (-np.arange(nx)) % nx
    jy = (-np.arange(ny)) % ny
    if not packed:
        x_lo = ((np.arange(nx) > 0) & (np.arange(nx) < nx // 2))[:, None, None]
        y_lo = ((np.arange(ny) > 0) & (np.arange(ny) < ny // 2))[None, :, None]
        x_hi = (np.arange(nx) > nx // 2)[:, None, None]
        y_hi = (np.arange(ny) > ny // 2)[None, :, None]
        z = np.arange(nz)[None, None, :]
        dest3 = ((z > nz // 2) & ~(x_lo & y_lo)) | ((z > 0) & (z < nz // 2) & x_hi & y_hi)
        mirrored3 = np.conj(data[jx][:, jy][:, :, (-np.arange(nz)) % nz])
        np.copyto(data, mirrored3, where=dest3)
    ixg = np.arange(nx)[:, None]
    iyg = np.arange(ny)[None, :]
    x_edge = (ixg == 0) | (ixg == nx // 2)
    y_edge = (iyg == 0) | (iyg == ny // 2)
    vertex = x_edge & y_edge
    dest = ((iyg > ny // 2) | (y_edge & (ixg > nx // 2))) & ~vertex
    for iz in (0, nz // 2):
        plane = data[:, :, iz]
        mirrored = np.conj(plane[jx][:, jy])
        plane[dest] = mirrored[dest]
        plane.imag[vertex] = 0
    data.real[0, 0, 0] = 0


class _Layout(object):
    """Shapes and scalar types of a plan's two sides."""
    __slots__ = ("shape", "shape_in", "shape_out", "dtype_in", "dtype_out")

    def __init__(self, shape, shape_in, shape_out, dtype_in, dtype_out):
        self.shape, self.shape_in, self.shape_out = shape, shape_in, shape_out
        self.dtype_in, self.dtype_out = dtype_in, dtype_out


# (packed, inverse) -> (required family of dtype_in, message when it is not met, dtype_out from dtype_in)
_KINDS = {
    (True, True): (np.complexfloating, "Invalid dtype_in for inverse packed transform (should be complex): {0}.",
                   lambda t: scalar_type(t)),
    (True, False): (np.floating, "Invalid dtype_in for forward packed transform (should be floating): {0}.",
                    lambda t: complex_type(t)),
    (False, True): (np.complexfloating, "Expected complex dtype_in for transform: {0}.", lambda t: t),
    (False, False): (np.complexfloating, "Expected complex dtype_in for transform: {0}.", lambda t: t),
}


def _resolve_layout(shape, dtype_in, data_in, overwrite, inverse, packed):
    """Validate the constructor arguments of :class:`Plan` and work out both sides' shapes and types
    (the argument rules, exception types and messages of transform.py:170-225)."""
    try:
        nx, ny, nz = shape
    except (TypeError, ValueError):
        raise ValueError("Expected 3D shape.")
    if nx % 2 or ny % 2 or nz % 2:
        raise ValueError("All shape dimensions must be even.")
    shape = (nx, ny, nz)
    if data_in is not None:
        if not isinstance(data_in, np.ndarray):
            raise ValueError("Invalid type for data_in: {0}.".format(type(data_in)))
        dtype_in = data_in.dtype
    dtype_in = _sctype(dtype_in)
    if dtype_in is None:
        raise ValueError("Invalid dtype_in: {0}.".format(dtype_in))
    family, complaint, out_of = _KINDS[(bool(packed), bool(inverse))]
    if not issubclass(dtype_in, family):
        raise ValueError(complaint.format(dtype_in))
    half, padded = (nx, ny, nz // 2 + 1), (nx, ny, nz + 2)       # complex side / in-place real side of a packed plan
    real_side = padded if overwrite else shape
    if not packed:
        shape_in = shape_out = shape
    elif inverse:
        shape_in, shape_out = half, real_side
    else:
        shape_in, shape_out = real_side, half
    if data_in is not None and data_in.shape != shape_in:
        raise ValueError("data_in has wrong shape {0}, expected {1}.".format(data_in.shape, shape_in))
    return _Layout(shape, shape_in, shape_out, dtype_in, out_of(dtype_in))


class Plan(object):
    """
    A plan for performing fast Fourier transforms on a single buffer
    (transform.py:161-315).  Transforms follow the numpy.fft normalisation:
    forward unnormalised, inverse divided by nx*ny*nz.

    Host arrays keep the reference's ownership and aliasing rules: the plan owns
    one buffer; ``data_in``, ``data_out`` and ``data_out_padded`` /
    ``data_in_padded`` alias it when ``overwrite`` is set; a caller-supplied
    ``data_in`` is adopted (``nbytes_allocated == 0``).

    With the ``hip`` backend ``execute()`` uploads ``data_in``, runs the
    transform on the GPU and downloads the result into ``data_out`` (the host
    copies are the price of the numpy-array API; device-resident use goes through
    :class:`randomfield_amd.generate.Generator` or :attr:`device`).  Packed inverse
    (c2r) and forward (r2c) plans and unpacked complex-to-complex plans (both directions)
    run on the GPU.  A c2r transform treats the kz = 0 and kz = nz/2 planes exactly as
    ``np.fft.irfftn`` does: only their 2-D Hermitian parts contribute (the device packs the
    two planes into one complex plane and projects them on upload; Hermitian input, e.g.
    after :func:`symmetrize`, passes through unchanged).
    """

    def __init__(self, shape, dtype_in=None, data_in=None, overwrite=True, inverse=True, packed=True,
                 use_pyfftw=True, backend=None, _device=None):
        lay = _resolve_layout(shape, dtype_in, data_in, overwrite, inverse, packed)
        nz = lay.shape[2]
        self.nbytes_allocated = 0
        if data_in is None:
            data_in = allocate(lay.shape_in, lay.dtype_in, use_pyfftw=use_pyfftw)
            self.nbytes_allocated += data_in.nbytes
        # One buffer, several views (transform.py:227-242): an overwriting packed plan reinterprets its input
        # buffer as the output type; whichever side is real carries FFTW's two padding columns, exposed as
        # `*_padded` and hidden from `data_in` / `data_out` by a slice (no copy).
        self.data_in = data_in
        if not overwrite:
            self.data_out = allocate(lay.shape_out, lay.dtype_out, use_pyfftw=use_pyfftw)
            self.nbytes_allocated += self.data_out.nbytes
        elif not packed:
            self.data_out = data_in
        else:
            alias = data_in.view(lay.dtype_out).reshape(lay.shape_out)
            if inverse:
                self.data_out_padded, self.data_out = alias, alias[:, :, :nz]
            else:
                self.data_in_padded, self.data_in, self.data_out = data_in, data_in[:, :, :nz], alias

        self.use_pyfftw = False          # pyFFTW is never used (kept as an attribute for compatibility)
        self.shape = lay.shape
        self.inverse = inverse
        self.packed = packed
        self.overwrite = overwrite
        self.backend = resolve_backend(backend)
        self.device = None

        if self.backend == "hip":
            from . import _hip
            _hip.require_gpu()           # raises: library missing / no GPU
            nx, ny = lay.shape[0], lay.shape[1]
            cdtype = lay.dtype_in if (inverse or not packed) else lay.dtype_out
            if np.dtype(cdtype) not in (np.dtype(np.complex64), np.dtype(np.complex128)):
                raise RuntimeError("hip backend supports complex64 / complex128 only: {0}.".format(cdtype))
            # power-of-two axes (8..2048; nz from 16) run on the tiled kernels, any other even shape on the generic mixed-radix
            # kernels (csrc/rf_generic.h): an axis of up to 8192 (complex64) / 4096 (complex128) points as one line in LDS, a longer
            # one as two passes over factors that fit; the library decides and refuses the rest
            if packed and not _hip.shape_supported(nx, ny, nz, cdtype):
                raise RuntimeError(
                    "hip backend: shape {0} is not supported (even axes that fit one line of the LDS -- 8192 points for complex64, "
                    "4096 for complex128 -- or split into two factors that do); "
                    "use backend='numpy' explicitly for this shape.".format(tuple(shape)))
            # a reverse plan that shares our memory also shares our device plan (one device buffer, as the
            # reference's pair of plans shares one host buffer)
            self.device = _device if _device is not None else _hip.DevicePlan(nx, ny, nz, cdtype, unpacked=not packed)
        else:
            if inverse:
                self.transformer = np.fft.irfftn if packed else np.fft.ifftn
            else:
                self.transformer = np.fft.rfftn if packed else np.fft.fftn

    def create_reverse_plan(self, reuse_output=True, overwrite=True):
        """
        Create a plan that reverses this plan (transform.py:278-301).

        When reuse_output is set, the new plan's data_in uses the same memory
        as our data_out.  Otherwise a new un-initialized data_in buffer is
        allocated for the new plan.
        """
        spec = dict(shape=self.shape, overwrite=overwrite, inverse=not self.inverse, packed=self.packed,
                    use_pyfftw=self.use_pyfftw, backend=self.backend,
                    _device=self.device if self.backend == "hip" else None)
        if not reuse_output:
            return Plan(dtype_in=self.data_out.dtype, **spec)
        # an overwriting reverse of a packed c2r plan needs the padded real buffer, which exists only if this plan
        # overwrites too
        wants_padded = self.packed and self.inverse and overwrite
        if wants_padded and not self.overwrite:
            raise RuntimeError("Cannot re-use output for reverse plan.")
        return Plan(data_in=self.data_out_padded if wants_padded else self.data_out, **spec)

    def execute(self):
        """Run the transform; returns ``data_out`` (transform.py:303-315)."""
        if self.backend == "hip":
            dev = self.device
            if not self.packed:                     # complex-to-complex over the full array
                dev.upload_c(self.data_in)
                dev.execute_c2c(self.inverse)
                if self.data_out.flags.c_contiguous:
                    dev.download_c(self.data_out)
                else:
                    self.data_out[:] = dev.download_c()
                return self.data_out
            if self.inverse:
                dev.upload_k(np.ascontiguousarray(self.data_in))
                dev.execute_c2r()
                if self.overwrite:
                    dev.download_real(self.data_out_padded, padded=True)
                else:
                    dev.download_real(self.data_out, padded=False)
            else:
                if self.overwrite:
                    dev.upload_real(self.data_in_padded, padded=True)
                else:
                    dev.upload_real(np.ascontiguousarray(self.data_in), padded=False)
                dev.execute_r2c()
                dev.download_k(self.data_out)
            return self.data_out
        if self.packed and self.inverse:
            nx, ny, nz = self.shape
            self.data_out[:] = self.transformer(self.data_in, s=(nx, ny, nz), axes=(0, 1, 2))
        else:
            self.data_out[:] = self.transformer(self.data_in, axes=(0, 1, 2))
        return self.data_out
