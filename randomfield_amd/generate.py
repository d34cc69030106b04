"""
High-level functions to generate random fields -- mirror of
``randomfield/generate.py`` :class:`Generator` for the Fourier-space sampling
path, running on MI355X through ``librandomfield_hip.so``.

Same constructor / method signatures as the reference (generate.py:74-75,
144-146, 232-233, 282) plus keyword-only extensions.  The call order of the
reference's ``generate_delta_field`` (generate.py:191-199,218-219)

    fill_with_log10k -> filter_power -> tabulate_sigmas -> randomize
    -> symmetrize -> [save potential] -> Plan.execute -> np.std

is preserved in meaning, but on the ``hip`` backend the first five steps are
evaluated per cell inside the first FFT pass of one fused GPU pipeline
(``rf_realise``); the k-space array is only materialised when
``save_potential=True`` needs it.

Out of scope here (host-only side-cars of the reference):
``calculate_lensing_potential`` and ``plot_slice``.
"""
from __future__ import annotations

import os

import numpy as np

from . import cosmotools, powertools, transform
from . import random as rf_random

__all__ = ["Generator"]


class _DevicePotential(object):
    """Marker for a saved delta(k)/k**2 field that lives in device memory."""

    def __init__(self, generator):
        self._generator = generator

    def download(self):
        """Host copy (nx, ny, nz/2+1) of the saved potential."""
        dev = self._generator.plan_c2r.device
        dev.load_potential(1.0)
        return dev.download_k()


class _RegeneratedPotential(object):
    """The saved potential of a ``generate_delta_field(save_potential=True)`` call that was NOT written to memory: the native
    generator is keyed by (seed, cell) and the replayed reference stream stays resident on the device, so delta(k)/k**2 is
    formed again inside the generation pass when :meth:`Generator.calculate_newtonian_potential` asks for it
    (rf_realise_scaled_potential) -- the default call writes 20 instead of 28 B/cell and the later transform reads nothing."""

    def __init__(self, generator, seed, noise):
        self._generator, self.seed, self.noise = generator, seed, noise
        dev = generator.plan_c2r.device
        # the device state this potential is a function of: the power / k tables, and (replayed stream) the resident deviates
        self._epochs = (dev.power_epoch, dev.noise_epoch if noise is not None else None)

    def check_current(self):
        """Raise unless the device plan still holds what the delta field was made from.  The reference's stored
        ``self.potential`` cannot change behind the caller's back; a regenerated one could, if somebody replaced the resident
        deviates (``reference_noise``, a same-seed batch, host deviates) or the power tables (``set_power``) through
        ``plan_c2r.device`` in between -- the result would silently belong to another field."""
        dev = self._generator.plan_c2r.device
        now = (dev.power_epoch, dev.noise_epoch if self.noise is not None else None)
        if now != self._epochs:
            what = "power / k tables" if now[0] != self._epochs[0] else "resident deviates"
            raise RuntimeError("The saved potential of the last generate_delta_field() call can no longer be formed: the device "
                               "plan's {0} have changed since.  Generate the field again, or use "
                               "Generator(store_potential=True) to keep delta(k)/k**2 in memory.".format(what))

    def download(self):
        """Host copy (nx, ny, nz/2+1) of the potential.  Runs the storing form of the call (rf_realise_potential) with the same
        seed.  Side effect: the plan's FIELD buffer is overwritten with the delta field of that seed again -- whatever
        ``convert_delta_to_density`` / ``calculate_newtonian_potential`` had left there is gone (download the field first)."""
        self.check_current()
        dev = self._generator.plan_c2r.device
        dev.realise_potential(self.seed, self.noise)
        dev.load_potential(1.0)
        self._generator._field_on_host = False
        return dev.download_k()


class Generator(object):
    """
    Manage random field generation for a specified geometry.

    Parameters (as the reference, generate.py:43-73)
    ----------
    nx, ny, nz : int
        Grid size; z is the line-of-sight direction.
    grid_spacing_Mpc_h : float
        Uniform grid spacing in Mpc/h.
    num_plot_sections : int
        Kept for compatibility: nz must be divisible by it (generate.py:86-90).
    cosmology : object, optional
        Kept for the caller's reference; the package never evaluates it.  The O(nz)
        tables the reference derives from it (generate.py:104-129) are separate
        arguments below, and passing ``cosmology`` without ``growth_function`` and
        ``mean_matter_density`` raises ``ValueError``.  Only its ``Om0`` attribute is
        read, by :meth:`calculate_newtonian_potential` when ``scale`` is not given.
    power : numpy.ndarray, optional
        Structured array with fields 'k', 'Pk' (powertools.validate_power).
        Defaults to the shipped Planck13 table.
    verbose : bool, optional

    Keyword-only extensions
    -----------------------
    backend : 'hip' (default) or 'numpy'
        See :mod:`randomfield_amd.transform`.  'hip' raises RuntimeError when the
        GPU path is unavailable; it never falls back to the CPU.
    dtype : numpy complex dtype
        complex64 (reference behaviour, generate.py:77) or complex128.
    rng : 'reference' (default) or 'native'
        'reference' draws the reference's own stream (``RandomState(seed).normal``,
        random.py:24) -- replayed on the GPU (MT19937 with jump-ahead + polar method)
        for integer seeds -- so the same seed gives the reference's field (to float32
        FFT rounding).
        'native' uses the GPU's counter-based Philox4x32-7 + Box-Muller generator:
        no host work, different (statistically equivalent) realisations.
    growth_function, mean_matter_density, redshifts : (nz,) arrays, optional
        The cosmology tables along z (the reference builds them with astropy, generate.py:104-129).
    transverse_distance : (nz,) array, optional
        Comoving transverse distance DA along z in Mpc/h (``self.DA`` of the reference,
        generate.py:117-118), for :meth:`calculate_lensing_potential`.
    curvature_K : float, optional
        Curvature constant K in (Mpc/h)**-2 (generate.py:380-381: -Ok0 (H0/c)**2); 0 = flat.
    distributed : bool, optional
        One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK from the environment, as ``torch.distributed.run`` sets
        them): k space is split by kz planes, the field by x planes, and every method works on this rank's
        ``nx / WORLD_SIZE`` planes of the field (``data_out`` has that many; see :mod:`randomfield_amd.slab`).  All ranks
        must make the same calls with the same arguments; ``seed=None`` is agreed between the ranks.
    exchange : {None, 'auto', 'direct', 'rccl'}, optional
        ``distributed=True``: how the blocks move between the y and z passes -- 'direct': every rank's y pass stores its output
        straight into the IPC-mapped receive buffers of its peers (no send / receive kernels, no extra sweep of local memory);
        'rccl': one grouped ncclSend / ncclRecv all-to-all; 'auto' (default, also ``RANDOMFIELD_EXCHANGE``): direct if every
        rank can, else rccl.  The fields are identical.  ``exchange_chunks`` (or ``RANDOMFIELD_EXCHANGE_CHUNKS``: 'auto' = 1, or
        a power of two) cuts the rank's kz slab into sub-slabs that travel while the next one is being generated.
    store_potential : bool, optional
        hip backend: ``generate_delta_field(save_potential=True)`` normally does not write delta(k)/k**2 to memory when it can be
        formed again on demand from the seed (``rng='native'``) or from the replayed deviates still on the device (complex64,
        ``rng='reference'``): :meth:`calculate_newtonian_potential` then regenerates it inside its generation pass.  ``True``
        stores it at generation time as the reference does (generate.py:200-217).  The results agree to float32 rounding.
    """

    def __init__(self, nx, ny, nz, grid_spacing_Mpc_h, num_plot_sections=4, cosmology=None, power=None,
                 verbose=False, *, backend=None, dtype=np.complex64, rng="reference", growth_function=None,
                 mean_matter_density=None, redshifts=None, transverse_distance=None, curvature_K=0.0, distributed=False,
                 store_potential=False, exchange_chunks=None, exchange=None):
        self.backend = transform.resolve_backend(backend)
        self.distributed = bool(distributed)
        self.store_potential = bool(store_potential)
        if self.distributed and self.backend != "hip":
            raise ValueError("distributed=True needs backend='hip'.")
        if rng not in ("reference", "native"):
            raise ValueError("Invalid rng: {0} (expected 'reference' or 'native').".format(rng))
        if rng == "native" and self.backend != "hip":
            raise ValueError("rng='native' is the GPU generator; it needs backend='hip'.")
        self.rng = rng
        if self.distributed:
            from . import slab
            if nx % 2 or ny % 2 or nz % 2:
                raise ValueError("All shape dimensions must be even.")
            self.plan_c2r = slab.SlabHostPlan(slab.DistributedPlan(nx, ny, nz, dtype, exchange=exchange), dtype)
            # every call here is ONE realisation.  Its all-to-all can be cut into sub-slabs that travel while the next sub-slab is
            # still being generated and transformed (RF_FLAG_EXCHANGE_CHUNKS) -- OPT-IN (`exchange_chunks=` or the environment
            # variable RANDOMFIELD_EXCHANGE_CHUNKS): the two-stream schedule is verified with virtual ranks and the forced slab
            # path on one GPU only; the default stays the plain forward -> exchange -> backward sequence on one stream until a
            # multi-GPU run has shown the same field and a gain
            chunks = slab.exchange_chunks_setting(exchange_chunks)      # (one parser for bench.py and this class)
            if chunks == "auto":
                chunks = 1
            if self.plan_c2r.device.nranks > 1 and chunks > 1:
                self.plan_c2r.device.set_exchange_chunks(chunks)
            # generate.py:79-80: the forward plan over the same memory (here: this rank's window of the field in, its kz planes out)
            self.plan_r2c = self.plan_c2r.create_reverse_plan(reuse_output=True, overwrite=True)
        else:
            self.plan_c2r = transform.Plan(shape=(nx, ny, nz), dtype_in=dtype, packed=True, overwrite=True,
                                           inverse=True, use_pyfftw=True, backend=self.backend)
            self.plan_r2c = self.plan_c2r.create_reverse_plan(reuse_output=True, overwrite=True)
        self.grid_spacing_Mpc_h = grid_spacing_Mpc_h
        self.k_min, self.k_max = powertools.get_k_bounds(self.plan_c2r.data_in, grid_spacing_Mpc_h, packed=True)
        self.potential = None

        if nz % num_plot_sections != 0:
            raise ValueError("Z-axis does not evenly divided into {0} plot sections.".format(num_plot_sections))
        self.num_plot_sections = num_plot_sections

        # The reference derives its O(nz) background tables (redshifts, growth function, mean matter density, DA;
        # generate.py:104-129) from an astropy cosmology.  That derivation is outside this package:
        # the tables are constructor arguments.  A `cosmology` object is kept for the caller's reference only, and
        # asking for one WITHOUT the tables it would have produced is refused here rather than failing later.
        self.cosmology = cosmology
        if cosmology is not None and (growth_function is None or mean_matter_density is None):
            raise ValueError("cosmology= is not evaluated by this package: pass the tables it implies as arrays "
                             "(growth_function=, mean_matter_density=, and redshifts= / transverse_distance= if "
                             "needed).")
        if power is None:
            if cosmology is not None:
                raise ValueError("A tabulated power= is required with a custom cosmology "
                                 "(the CLASS-based calculate_power is not part of this package).")
            power = powertools.load_default_power()
        self.power = powertools.validate_power(power)

        self.redshifts = None if redshifts is None else np.asarray(redshifts, float).reshape(nz)
        self.growth_function = None if growth_function is None else np.asarray(growth_function, float).reshape(nz)
        self.mean_matter_density = (None if mean_matter_density is None
                                    else np.asarray(mean_matter_density, float).reshape(nz))
        self.DC = np.arange(nz) * self.grid_spacing_Mpc_h
        self.DA = None if transverse_distance is None else np.asarray(transverse_distance, float).reshape(nz)
        self.curvature_K = float(curvature_K)

        self.delta_field_rms = None
        self.smoothed_power = None
        self._field_on_host = False
        if self.backend == "hip":
            dev = self.plan_c2r.device
            dev.set_kgrid(*powertools.ksq_axes(nx, ny, nz, grid_spacing_Mpc_h))

        self.verbose = verbose
        if self.verbose:
            Mb = (self.plan_c2r.nbytes_allocated + (self.plan_r2c.nbytes_allocated if self.plan_r2c else 0)) / 2.0 ** 20
            print("Allocated {0:.1f} Mb for {1} x {2} x {3} grid.".format(Mb, nx, ny, nz))
            print("{0} Mpc/h spacing covered by k = {1:.5f} - {2:.5f} h/Mpc."
                  .format(self.grid_spacing_Mpc_h, self.k_min, self.k_max))
            if self.backend == "hip":
                print("Device buffers: {0:.1f} Mb.".format(self.plan_c2r.device.nbytes / 2.0 ** 20))

    # ------------------------------------------------------------------
    @property
    def shape(self):
        return self.plan_c2r.shape

    def _native_seed(self, seed):
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
            if self.distributed:
                seed = self.plan_c2r.agree_on(seed >> 12)          # 52 bits survive the float64 all-reduce
        return int(seed) & (2 ** 64 - 1)

    def generate_delta_field(self, smoothing_length_Mpc_h=0., seed=None, save_potential=True, show_plot=False,
                             save_plot_name=None, *, download=True):
        """
        Generate a delta-field realization (generate.py:144-230).

        The delta field is calculated at redshift zero and sampled from a
        distribution with mean zero and k-space variance proportional to the
        smoothed power spectrum.  ``seed=None`` draws a fresh seed.

        Returns a 3D array of delta values that is a *view* of the plan's host
        buffer and will be overwritten by subsequent operations (as in the
        reference).  With ``download=False`` (hip backend) the field stays on the
        GPU, ``None`` is returned, and :meth:`download_field` fetches it later.
        """
        if show_plot or save_plot_name is not None:
            raise NotImplementedError("plot_slice is outside the accelerated path (see DESIGN.md).")
        nx, ny, nz = self.plan_c2r.shape
        self.smoothed_power = powertools.filter_power(self.power, smoothing_length_Mpc_h)

        if self.backend == "numpy":
            data = self.plan_c2r.data_in
            powertools.fill_with_log10k(data, spacing=self.grid_spacing_Mpc_h, packed=True)
            powertools.tabulate_sigmas(data, power=self.smoothed_power, spacing=self.grid_spacing_Mpc_h, packed=True)
            rf_random.randomize(data, seed=seed)
            transform.symmetrize(data, packed=True)
            if save_potential:
                if self.potential is None or isinstance(self.potential, _DevicePotential):
                    self.potential = np.empty_like(data)
                self.potential.imag = 0.
                kx2, ky2, kz2 = powertools.create_ksq_grids(self.potential, spacing=self.grid_spacing_Mpc_h,
                                                            packed=True)
                np.add(kx2, ky2, out=self.potential.real, casting="same_kind")
                np.add(self.potential.real, kz2, out=self.potential.real, casting="same_kind")
                with np.errstate(divide="ignore"):
                    np.reciprocal(self.potential.real, out=self.potential.real)
                self.potential[0, 0, 0] = 0.
                self.potential *= data
            else:
                self.potential = None
            delta = self.plan_c2r.execute()
            self.delta_field_rms = np.std(delta.reshape(-1))
            self._field_on_host = True
        else:
            dev = self.plan_c2r.device
            log10_k, sigma = powertools.sigma_table(self.smoothed_power, (nx, ny, nz), self.grid_spacing_Mpc_h)
            # same power and smoothing as the tables the device plan holds: nothing to upload (a stream sync and a table rebuild
            # per call otherwise); the plan itself remembers what it was last given, whoever gave it
            dev.set_power(log10_k, sigma, if_changed=True)
            realised = False
            # generate.py:184-189,230 returns a HOST array, and the device -> host copy is 15 x the realisation at 1024^3: armed with
            # the plan's host buffer, the realisation below delivers slab by slab behind its z pass (rf_set_host_sink)
            sink = bool(download) and not self.distributed and dev.arm_host_sink(self.plan_c2r.data_out_padded, padded=True)
            self._sink_armed = sink
            if self.rng == "reference":
                # RandomState(seed).normal (random.py:24): MT19937 + polar method replayed on the GPU from the seed's
                # 624-word start state -- integer seeds, array seeds (init_by_array) and None alike; no host deviates
                if seed is None and self.distributed:
                    seed = self.plan_c2r.agree_on(int.from_bytes(os.urandom(4), "little"))
                # (a complex64 plan keeps float32 copies of the deviates: its cells sigma * g are float32 anyway)
                single = self.plan_c2r.data_out.dtype == np.float32
                if self.distributed and dev.nranks > 1 and dev.tiled and dev.share_segments()[0] >= dev.nranks:
                    # one stream, P ranks: each replays 1/P of it, one all-to-all of deviates (rf_mt_share_*); the first
                    # time round under the watchdog that names a rank which never arrived (slab.Deadline)
                    if getattr(self, "_shared_replay_ran", False):
                        dev.reference_noise_shared(seed, single=single)
                    else:
                        with self.plan_c2r.dist.deadline("first shared replay of the reference stream"):
                            dev.reference_noise_shared(seed, single=single)
                        self._shared_replay_ran = True
                elif (single and not self.distributed and not (save_potential and self.store_potential)
                      and dev.can_batch_reference()):
                    # one device call: the replay and the passes queued back to back (no host synchronisation between them,
                    # untimed passes: the z pass of a slab shares its launch with the y pass of the next) -- a same-seed
                    # batch of one (rf_realise_batch_reference); the deviates stay resident as float32 pairs
                    dev.realise_batch_reference([seed], want_rms=False)
                    realised = True
                else:
                    dev.reference_noise(seed, single=single)
                noise = "resident"
                dseed = 0
            else:
                noise = None
                dseed = self._native_seed(seed)
            if save_potential:
                # generate.py:200-217.  Native noise: delta(k)/k**2 is a second store stream of the generation pass;
                # otherwise the library runs rows K,T,R,S -> k-space, the division, and the c2r transform.
                if not self.store_potential and dev.can_regenerate_potential(noise):
                    # nothing to store: the potential can be formed again from the seed / the resident deviates on demand
                    if not realised:
                        self._realise(dev, dseed, noise)
                    self.potential = _RegeneratedPotential(self, dseed, noise)
                else:
                    dev.realise_potential(dseed, noise)
                    self.potential = _DevicePotential(self)
            else:
                if not realised:
                    self._realise(dev, dseed, noise)    # fused: k-space never materialised
                self.potential = None
            mean, std = dev.moments()
            self.delta_field_rms = self.plan_c2r.data_out.dtype.type(std)
            self._field_on_host = bool(sink and dev.host_sink_delivered())
            self._sink_armed = False
            delta = self.download_field() if download else None

        if self.verbose:
            print("Delta field has standard deviation {0:.3f}.".format(self.delta_field_rms))
        return delta

    def _realise(self, dev, dseed, noise):
        """One fused realisation on the device.  Native generator on a single-GPU tiled plan: replayed from a captured
        one-realisation hipGraph (rf_realise_batch with one seed, read from device memory) -- the call is ~35 kernel launches, and
        their launch gaps are 5 % of a 1024^3 realisation when issued one by one; everything else issues them eagerly."""
        if noise is None and not self.distributed and dev.tiled and dev.nranks == 1 and not getattr(self, "_sink_armed", False):
            dev.realise_batch(np.array([dseed], np.uint64), want_rms=False)       # (a captured graph keeps its field on the device: not with a host sink)
        else:
            dev.realise(dseed, noise)

    def generate_density_field(self, smoothing_length_Mpc_h=0., seed=None, *, download=True):
        """
        ``generate_delta_field(save_potential=False)`` followed by ``convert_delta_to_density(apply_lognormal_transform=True)``
        (generate.py:191-199,218-219 and 266-273) as ONE device call (an extension; the reference has no such method).

        On a single-GPU hip plan with power-of-two axes the rms of the Gaussian field is taken from the transform's second
        pass (Parseval) and the lognormal map and the mean-density factor run in the epilogue of the last pass
        (``rf_realise_lognormal``): five sweeps of the array instead of seven, no host round trip for sigma.  The result is
        the reference's density field up to a few ulp in the argument of ``exp``.  Everywhere else (numpy backend, multi-GPU
        plans, generic shapes) the two reference calls run one after the other.  ``delta_field_rms`` is set as usual.
        """
        growth = self._need_table("growth_function")
        density = self._need_table("mean_matter_density")
        dev = self.plan_c2r.device if self.backend == "hip" else None
        fused = dev is not None and not self.distributed and dev.nranks == 1 and dev.tiled
        if not fused:
            self.generate_delta_field(smoothing_length_Mpc_h, seed, save_potential=False, download=False) if self.backend == "hip" \
                else self.generate_delta_field(smoothing_length_Mpc_h, seed, save_potential=False)
            return self.convert_delta_to_density(apply_lognormal_transform=True, download=download)
        nx, ny, nz = self.plan_c2r.shape
        self.smoothed_power = powertools.filter_power(self.power, smoothing_length_Mpc_h)
        log10_k, sigma = powertools.sigma_table(self.smoothed_power, (nx, ny, nz), self.grid_spacing_Mpc_h)
        dev.set_power(log10_k, sigma, if_changed=True)
        tables = (np.asarray(growth, np.float64).tobytes(), np.asarray(density, np.float64).tobytes())
        if getattr(dev, "_z_tables_key", None) != tables:
            dev.set_z_tables(growth, density)
            dev._z_tables_key = tables
        if self.rng == "reference":
            dev.reference_noise(seed, single=self.plan_c2r.data_out.dtype == np.float32)
            rms = dev.realise_lognormal(0, "resident")
        else:
            rms = dev.realise_lognormal(self._native_seed(seed), None)
        self.potential = None
        self.delta_field_rms = self.plan_c2r.data_out.dtype.type(rms)
        self._field_on_host = False
        if self.verbose:
            print("Delta field has standard deviation {0:.3f}.".format(self.delta_field_rms))
        return self.download_field() if download else None

    def download_field(self):
        """Copy the device-resident field into the plan's host buffer and return the view."""
        if self.backend == "hip" and not self._field_on_host:
            self.plan_c2r.device.download_real(self.plan_c2r.data_out_padded, padded=True)
            self._field_on_host = True
        return self.plan_c2r.data_out

    def _need_table(self, name):
        value = getattr(self, name)
        if value is None:
            arg = {"DA": "transverse_distance"}.get(name, name)
            raise RuntimeError("Generator.{0} is not available: pass {1}= (an (nz,) array) to the constructor "
                               "(astropy-based tables are outside the accelerated path).".format(name, arg))
        return value

    def convert_delta_to_density(self, apply_lognormal_transform=True, show_plot=False, save_plot_name=None, *,
                                 download=True):
        """
        Convert a delta field into a density field with light-cone evolution
        (generate.py:232-280): lognormal map with sigma = delta_field_rms and the
        growth function along z (or ``delta*growth + 1``), then multiplication by
        the mean matter density along z.
        """
        if show_plot or save_plot_name is not None:
            raise NotImplementedError("plot_slice is outside the accelerated path (see DESIGN.md).")
        if self.delta_field_rms is None:
            raise RuntimeError("No delta field has been generated.")
        growth = self._need_table("growth_function")
        density = self._need_table("mean_matter_density")
        nz = self.plan_c2r.shape[2]
        if self.backend == "numpy":
            delta = self.plan_c2r.data_out
            if apply_lognormal_transform:
                delta = cosmotools.apply_lognormal_transform(delta, growth, sigma=self.delta_field_rms)
            else:
                delta *= growth
                delta += 1
            delta *= density
            return delta
        dev = self.plan_c2r.device
        if apply_lognormal_transform:
            a_z, b_z = cosmotools.lognormal_tables(growth, self.delta_field_rms, nz)
            dev.lognormal(a_z, b_z, float(self.delta_field_rms))
        else:
            dev.affine_z(growth, 1.0)
        dev.scale_z(density)
        self._field_on_host = False
        return self.download_field() if download else None

    def calculate_newtonian_potential(self, light_cone=True, show_plot=False, save_plot_name=None, *,
                                      scale=None, download=True):
        """
        Calculate the Newtonian potential Phi(r) (generate.py:282-350): inverse
        transform of ``scale * delta(k)/k**2`` with scale = -3/2 H0**2 Omega_m
        (in s**-2, H0 = 100 km/s/Mpc), optionally times G(z)/(1+z) along z.

        ``scale`` may be given explicitly; otherwise it is -3/2 H0**2 ``cosmology.Om0``.
        """
        if show_plot or save_plot_name is not None:
            raise NotImplementedError("plot_slice is outside the accelerated path (see DESIGN.md).")
        if self.potential is None:
            raise RuntimeError("No saved potential field.")
        if scale is None:
            om0 = getattr(self.cosmology, "Om0", None)
            if om0 is None:
                raise RuntimeError("calculate_newtonian_potential needs scale= (or a cosmology= object with Om0).")
            H0 = 100.0 / 3.0856775814913673e19          # 100 km/s/Mpc in 1/s
            scale = -1.5 * H0 ** 2 * float(om0)
        factor = None
        if light_cone:
            factor = self._need_table("growth_function") / (1 + self._need_table("redshifts"))
        if self.backend == "numpy":
            self.plan_c2r.data_in[:] = self.potential
            self.plan_c2r.data_in *= scale
            field = self.plan_c2r.execute()
            if light_cone:
                field *= self._need_table("growth_function")
                field /= 1 + self._need_table("redshifts")
            return field
        dev = self.plan_c2r.device
        if isinstance(self.potential, _RegeneratedPotential):
            self.potential.check_current()
            dev.realise_scaled_potential(self.potential.seed, self.potential.noise, scale, factor_z=factor)
            factor = None           # (applied by the z pass itself)
        else:
            dev.load_potential(scale)
            dev.execute_c2r()
        if factor is not None:
            dev.scale_z(factor)
        self._field_on_host = False
        return self.download_field() if download else None

    def calculate_lensing_potential(self, i_min=None, show_plot=False, save_plot_name=None):
        """
        Calculate the lensing potential psi(r) (generate.py:352-416):

            psi(r, z) = integral from D_min to D_src of -2 [cotK(D) - cotK(D_src)] dPhi(r, z) dD

        with the reference's Simpson rule along z, D_min = ``DC[i_min]`` (default nz // 32).
        Call :meth:`calculate_newtonian_potential` (``light_cone=True``) just before.  Needs the
        ``transverse_distance=`` table (and ``curvature_K=`` for a curved cosmology).

        Returns a new array; the plan's buffer (the Newtonian potential) is not modified.  On the
        hip backend the O(nz**2)-per-column loop of the reference runs as one prefix scan per row.
        """
        if show_plot or save_plot_name is not None:
            raise NotImplementedError("plot_slice is outside the accelerated path (see DESIGN.md).")
        nDC = self.DC.size
        if i_min is None:
            i_min = nDC // 32
        if i_min < 0 or i_min >= nDC:
            raise ValueError("Invalid i_min {}. Expected 0 - {}.".format(i_min, nDC - 1))
        DA = self._need_table("DA")
        K = self.curvature_K
        if K < 0:
            cosK = np.cosh(np.sqrt(-K) * self.DC)
        elif K > 0:
            cosK = np.cos(np.sqrt(K) * self.DC)
        else:
            cosK = np.ones_like(DA, dtype=float)
        cotK = np.ones_like(cosK)
        cotK[1:] = cosK[1:] / DA[1:]
        if self.backend == "numpy":
            dPhi = self.plan_c2r.data_out
            psi = np.empty_like(dPhi)
            h = float(self.grid_spacing_Mpc_h)
            for i in range(nDC, i_min, -1):
                psi[:, :, i_min:i] = dPhi[:, :, i_min:i]
                psi[:, :, i_min:i] *= -2 * (cotK[i_min:i] - cotK[i - 1])
                psi[:, :, i - 1] = cosmotools.simps_avg(psi[:, :, i_min:i], h)
            if i_min > 0:
                psi[:, :, :i_min] = 0.
            return psi
        dev = self.plan_c2r.device
        dev.lensing_potential(cotK, self.grid_spacing_Mpc_h, i_min)
        return dev.download_aux()

    def plot_slice(self, *args, **kwargs):
        raise NotImplementedError("plot_slice is outside the accelerated path (see DESIGN.md).")
