"""Device-resident driver of the hot path: torch owns the HBM buffers and the stream, the C ABI
(`*_dev` entry points) does the work.  Used by bench.py, the GPU tests and smoke(); a Julia
caller with device-resident data (AMDGPU.jl ROCArrays) would call the same `_dev` symbols.

All arrays are flat torch tensors whose memory is Julia's column-major (nx,ny,nz) layout.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi
from .capi import HDIRS, MATS, PHI_ORDER


def _flat(a, dtype=np.float64):
    return np.asfortranarray(a, dtype=dtype).ravel(order="F")


class DeviceAssembler:
    """Holds one grid (gridmetrics + indices + parameters) in HBM and assembles transport matrices
    for successive (umo, vmo) fields -- the TMIP workflow of building T for many time slices."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceAssembler needs a GPU (no CPU fallback)")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ctx = capi.Context(device)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.lib = capi.lib()
        self.out = None
        # facefluxes also accumulates the tile counts of the transportmatrix that follows (otmb_facefluxes_counts_dev); False: the plain
        # kernel + the whole push mask + a counting pass, as before round 5 (A/B, and tests that look at the mask itself)
        self.count_in_ff = os.environ.get("OTMB_COUNT_IN_FF", "1") != "0"
        arena_gb = float(os.environ.get("OTMB_ARENA_GB", "0"))
        if arena_gb > 0 and os.environ.get("OTMB_ARENA_WHEN", "start") == "start":  # (see _empty)
            _arena = torch.empty(int(arena_gb * 2 ** 30), dtype=torch.uint8, device=self.device)
            del _arena

    def _t(self, a, dtype=np.float64):
        h = torch.from_numpy(_flat(a, dtype))
        d = self._empty(h.numel(), h.dtype)
        d.copy_(h)
        return d

    def _empty(self, n, dtype):
        """Device array of n elements.  Placement experiments (profiles/r04/README.md section 8; tools/placement_search.sh):
        OTMB_ARENA_GB=<GB> reserves one block when the assembler is created and hands it back to torch's caching allocator, so
        that every later array is carved out of that ONE device allocation, back to back; OTMB_ARENA_ALIGN=<bytes> rounds every
        array's size up to a multiple (array starts on that grid), OTMB_ARENA_PAD=<bytes> leaves a gap behind every array.
        OTMB_STAGGER=<bytes> (round 3, tools/placement_study.py) starts the k-th array k*stagger bytes into its allocation."""
        stagger = int(os.environ.get("OTMB_STAGGER", "0"))
        align = int(os.environ.get("OTMB_ARENA_ALIGN", "0"))
        gap = int(os.environ.get("OTMB_ARENA_PAD", "0"))
        item = torch.empty(0, dtype=dtype).element_size()
        if align > 0 or gap > 0:
            nbytes = max(n * item, 1)
            if align > 0:
                nbytes = (nbytes + align - 1) // align * align
            raw = torch.empty(nbytes + gap, dtype=torch.uint8, device=self.device)
            return raw[: n * item].view(dtype)
        if stagger <= 0:
            return torch.empty(n, dtype=dtype, device=self.device)
        self._nalloc = getattr(self, "_nalloc", 0) + 1
        pad = (self._nalloc * stagger) % (1 << 21)
        pad -= pad % 256
        raw = torch.empty(n * item + pad, dtype=torch.uint8, device=self.device)
        return raw[pad:pad + n * item].view(dtype)

    # ---- grid ---------------------------------------------------------------------------------
    def set_grid(self, gridmetrics, mlotst, rho, kappaH=500.0, kappaVML=0.1, kappaVdeep=1.0e-5, upwind=True):
        gm = gridmetrics
        self.shape = tuple(int(x) for x in gm["v3D"].shape)
        self.nx, self.ny, self.nz = self.shape
        self.G = self.nx * self.ny * self.nz
        t = gm["gridtopology"]
        self.topology = int(t["kind"]) if isinstance(t, dict) else int(t)
        self.v3d = self._t(gm["v3D"])
        self.thk = self._t(gm["thkcello"])
        self.edge = [self._t(gm["edge_length_2D"][d]) for d in HDIRS]
        self.dist = [self._t(gm["distance_to_neighbour_2D"][d]) for d in HDIRS]
        self.area = self._t(gm["area2D"])
        self.zt = self._t(gm["zt"])
        self.mlotst = self._t(mlotst)
        self.rho = None if np.ndim(rho) == 0 else self._t(rho)
        self.rho_scalar = float(rho) if np.ndim(rho) == 0 else 0.0
        self.kappa = (float(kappaH), float(kappaVML), float(kappaVdeep))
        self.upwind = bool(upwind)
        self.makeindices()

    def set_grid_tensors(self, *, shape, topology, v3d, thkcello, edge_length, dist_nbr, area2d, zt, mlotst, rho,
                         kappaH=500.0, kappaVML=0.1, kappaVdeep=1.0e-5, upwind=True):
        """Grid whose arrays already live on this GPU (flat float64 tensors in Julia's column-major order; edge_length /
        dist_nbr: four (nx*ny) tensors in OTMB_DIR_* order west, east, south, north; rho: tensor or number).  Nothing
        is copied: the caller keeps the tensors alive through this object."""
        self.shape = tuple(int(x) for x in shape)
        self.nx, self.ny, self.nz = self.shape
        self.G = self.nx * self.ny * self.nz
        self.topology = int(topology)
        self.v3d, self.thk = v3d, thkcello
        self.edge, self.dist = list(edge_length), list(dist_nbr)
        self.area, self.zt, self.mlotst = area2d, zt, mlotst
        self.rho = rho if torch.is_tensor(rho) else None
        self.rho_scalar = 0.0 if torch.is_tensor(rho) else float(rho)
        for t in (self.v3d, self.thk, *self.edge, *self.dist, self.area, self.zt, self.mlotst) + ((self.rho,) if self.rho is not None else ()):
            if t.device != self.device or t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError("set_grid_tensors: contiguous float64 tensors on this assembler's device are required")
        if self.v3d.numel() != self.G or self.thk.numel() != self.G or self.zt.numel() != self.nz:
            raise ValueError("set_grid_tensors: array sizes do not match shape")
        self.kappa = (float(kappaH), float(kappaVML), float(kappaVdeep))
        self.upwind = bool(upwind)
        self.makeindices()

    def set_grid_from_raw(self, *, areacello, volcello, lon, lat, lev, lon_vertices, lat_vertices, mlotst, rho,
                          kappaH=500.0, kappaVML=0.1, kappaVdeep=1.0e-5, upwind=True):
        """makegridmetrics on the device (otmb_makegridmetrics_dev): the raw CMIP arrays go up once, every derived
        array stays in HBM.  Host work is limited to the vertex permutation and the topology test."""
        from . import gridtopology as gt
        from ._nt import data_and_props
        from .gridmetrics import vertexpermutation

        area, ap = data_and_props(areacello)
        vol, vp = data_and_props(volcello)
        area = np.asfortranarray(area, dtype=np.float64)
        vol = np.asfortranarray(vol, dtype=np.float64)
        lonv = np.asfortranarray(data_and_props(lon_vertices)[0], dtype=np.float64)
        latv = np.asfortranarray(data_and_props(lat_vertices)[0], dtype=np.float64)
        perm = vertexpermutation(lonv, latv)
        topo = gt.getgridtopology(lonv[perm], latv[perm])
        self.shape = tuple(int(x) for x in vol.shape)
        self.nx, self.ny, self.nz = self.shape
        self.G, P = self.nx * self.ny * self.nz, self.nx * self.ny
        self.topology = topo
        f64 = lambda n: torch.empty(n, dtype=torch.float64, device=self.device)
        self.area, self.v3d, self.thk, self.z3d = f64(P), f64(self.G), f64(self.G), f64(self.G)
        self.edge, self.dist_edge, self.dist = [f64(P) for _ in range(4)], [f64(P) for _ in range(4)], [f64(P) for _ in range(4)]
        d_vol, d_area = self._t(vol), self._t(area)
        d_lon, d_lat = self._t(np.asarray(data_and_props(lon)[0])), self._t(np.asarray(data_and_props(lat)[0]))
        d_lonv, d_latv = self._t(lonv), self._t(latv)
        pa = (C.c_int32 * 4)(*perm)
        self.ctx.check(self.lib.otmb_makegridmetrics_dev(
            self.ctx.handle, d_vol.data_ptr(), d_area.data_ptr(), float(ap.get("_FillValue", np.nan)),
            float(vp.get("_FillValue", np.nan)), d_lon.data_ptr(), d_lat.data_ptr(), d_lonv.data_ptr(), d_latv.data_ptr(),
            C.byref(pa), self.nx, self.ny, self.nz, topo, self.area.data_ptr(), self.v3d.data_ptr(), self.thk.data_ptr(),
            self.z3d.data_ptr(), C.byref(capi.ptr_array(4, [t.data_ptr() for t in self.edge])),
            C.byref(capi.ptr_array(4, [t.data_ptr() for t in self.dist_edge])),
            C.byref(capi.ptr_array(4, [t.data_ptr() for t in self.dist]))))
        self.ctx.synchronize()
        self.zt = self._t(np.asarray(data_and_props(lev)[0]))
        self.mlotst = self._t(mlotst)
        self.rho = None if np.ndim(rho) == 0 else self._t(rho)
        self.rho_scalar = float(rho) if np.ndim(rho) == 0 else 0.0
        self.kappa = (float(kappaH), float(kappaVML), float(kappaVdeep))
        self.upwind = bool(upwind)
        self.makeindices()

    # ---- operators the caller passes (transportmatrix's Tadv = / TκH = / TκVML = / TκVdeep = keywords, src/matrixbuilding.jl:133-143) ----
    def set_given(self, **ops):
        """ops: name -> (colptr, rowval, nzval) device tensors (int64, int64, float64; rowval / nzval exactly nnz long) or None.  A passed
        operator is not built (otmb_tm_args.given): a TκH / TκVdeep with the rows the library derives for this grid is neither written nor
        counted -- its values are re-derived in registers or read where they lie (those of another κ included: ctx.given_state(m) == 3);
        any other matrix makes T a device sparse add (two-phase protocol only: transportmatrix(); the asynchronous calls raise
        GIVEN_FOREIGN)."""
        given = dict(getattr(self, "given", {}) or {})
        for name, triple in ops.items():
            if name not in MATS[1:]:
                raise ValueError(f"{name}: not an operator of transportmatrix")
            if triple is None:
                given.pop(name, None)
                continue
            cp, rv, nz = (t.contiguous() for t in triple)
            if cp.dtype != torch.int64 or rv.dtype != torch.int64 or nz.dtype != torch.float64 or cp.numel() != self.N + 1 or rv.numel() != nz.numel():
                raise ValueError(f"{name}: (colptr int64 [N + 1], rowval int64 [nnz], nzval float64 [nnz]) device tensors are required")
            given[name] = (cp, rv, nz)
        self.given = given
        self._given_key = None

    def _given_versions(self):
        ts = [self.lwet3d, self.lwet, self.v3d, self.thk, self.area, self.zt, *self.edge, *self.dist]
        for name in MATS[1:]:
            ts += list(self.given.get(name, ()))
        return tuple((t.data_ptr(), t._version) for t in ts) + tuple(self.kappa)

    def makeindices(self):
        """otmb_makeindices_dev on the resident v3D (src/matrixbuilding.jl:10-24)."""
        self._mask_key = None
        self.given, self._given_key = {}, None  # (operators of another grid)  # (fluxes of an earlier facefluxes call no longer come with counts / a mask for THESE indices)
        self.lwet3d = self._empty(self.G, torch.int64)
        self.lwet = self._empty(self.G, torch.int64)
        self.wet3d = torch.empty(self.G, dtype=torch.uint8, device=self.device)
        n = C.c_int64(0)
        self.ctx.check(self.lib.otmb_makeindices_dev(self.ctx.handle, self.v3d.data_ptr(), self.nx, self.ny, self.nz,
                                                     self.lwet3d.data_ptr(), self.lwet.data_ptr(),
                                                     self.wet3d.data_ptr(), C.byref(n)))
        self.N = int(n.value)
        # the five wet bytes nofluxboundaries! reads per cell and level, folded into one (otmb_wetflags_dev): once per grid
        self.wetflags = torch.empty(self.G, dtype=torch.uint8, device=self.device)
        self.ctx.check(self.lib.otmb_wetflags_dev(self.ctx.handle, self.wet3d.data_ptr(), self.nx, self.ny, self.nz, self.topology,
                                                  self.wetflags.data_ptr()))
        self._wetflags_version = self.wet3d._version
        self._count_tables()
        return self.N

    def _count_tables(self):
        """Counts in facefluxes (otmb_facefluxes_counts_dev), once per grid: the wet rank at which every 64-cell wave segment starts and the
        per-tile row counts that depend on the wet mask alone."""
        nb = int(self.lib.otmb_count_tables_bytes(self.ctx.handle, self.nx, self.ny, self.nz, self.N))
        self.count_tables = torch.empty(max((nb + 7) // 8, 1), dtype=torch.int64, device=self.device)
        self.ctx.check(self.lib.otmb_count_tables_dev(self.ctx.handle, self.lwet3d.data_ptr(), self.lwet.data_ptr(), self.wetflags.data_ptr(),
                                                      self.N, self.nx, self.ny, self.nz, self.topology, self.count_tables.data_ptr()))

    # ---- per time slice -----------------------------------------------------------------------
    def facefluxes(self, umo, vmo, fill):
        """umo/vmo: flat device tensors (float64 or float32).  Returns the six ϕ tensors (reused).  Raises the
        reference's "all values missing" assertion (velocities.jl:199-200) like otmb_facefluxes_dev."""
        phi = self.facefluxes_async(umo, vmo, fill)
        self._check_missing()
        return phi

    def facefluxes_async(self, umo, vmo, fill):
        """Same kernel, no host round trip: the "all values missing" assertion is evaluated by finish().  The
        kernel also writes the push mask of these fluxes (include/otmb.h), which lets the counting pass of the
        following transportmatrix skip the six ϕ arrays."""
        if getattr(self, "phi", None) is None:
            self.phi = [self._empty(self.G, torch.float64) for _ in range(6)]
            self.push_mask = self._empty(self.G, torch.int16)
        ptrs = capi.ptr_array(6, [p.data_ptr() for p in self.phi])
        self._mask_key = None
        self._note("_ff_seq")
        if self.wet3d._version != self._wetflags_version:  # the mask was edited in place (tests do): fold it again
            self.ctx.check(self.lib.otmb_wetflags_dev(self.ctx.handle, self.wet3d.data_ptr(), self.nx, self.ny, self.nz, self.topology,
                                                      self.wetflags.data_ptr()))
            self._wetflags_version = self.wet3d._version
        # The kernel also accumulates the tile counts of the transportmatrix these fluxes will be handed to (same mlotst, weighting, indices):
        # that call then has no counting pass.  (The library falls back to the plain kernel where it cannot count: nx < 3, OTMB_COUNT_IN_FF=0.)
        if not self.count_in_ff:
            self.ctx.check(self.lib.otmb_facefluxes_flags_dev(self.ctx.handle, umo.data_ptr(), vmo.data_ptr(),
                                                              int(umo.dtype == torch.float32), self.wetflags.data_ptr(), float(fill),
                                                              self.nx, self.ny, self.nz, self.topology, C.byref(ptrs), None,
                                                              self.push_mask.data_ptr()))
            self._mask_key = self._phi_key(self.phi)
            return self.phi
        cnt = capi.FfCounts()
        cnt.tables, cnt.lwet3d = self.count_tables.data_ptr(), self.lwet3d.data_ptr()
        cnt.mlotst, cnt.zt, cnt.n_wet = self.mlotst.data_ptr(), self.zt.data_ptr(), self.N
        cnt.upwind, cnt.only_t = int(self.upwind), 1 if getattr(self, "only_T", False) else 0
        self.ctx.check(self.lib.otmb_facefluxes_counts_dev(self.ctx.handle, umo.data_ptr(), vmo.data_ptr(),
                                                           int(umo.dtype == torch.float32), self.wetflags.data_ptr(), float(fill),
                                                           self.nx, self.ny, self.nz, self.topology, C.byref(ptrs),
                                                           self.push_mask.data_ptr(), C.byref(cnt)))
        self._mask_key = self._phi_key(self.phi)
        return self.phi

    def _note(self, which):
        """Asynchronous calls of both kinds are numbered in one sequence, so that finish() can tell which of a facefluxes
        failure (counted among facefluxes calls) and a transportmatrix failure (counted among transportmatrix calls) came
        FIRST even when the two kinds of call were not issued in pairs."""
        self._seq = getattr(self, "_seq", 0) + 1
        if not hasattr(self, which):
            setattr(self, which, [])
        getattr(self, which).append(self._seq)

    def _phi_key(self, phi):
        # the mask (and the tile counts that came with it) describes exactly the values facefluxes wrote, and the mixed-layer inputs and
        # weighting it was told: any later in-place torch op bumps _version
        return tuple((p.data_ptr(), p._version) for p in (*phi, self.mlotst, self.zt)) + (bool(self.upwind), bool(getattr(self, "only_T", False)))

    PIPELINE_DEPTH = 60  # the library remembers the verdicts of its 64 most recent asynchronous calls: drain before that

    def _first_missing(self):
        """Index (among the facefluxes calls since the previous check) of the first call whose umo or vmo held no valid
        value at all (the reference asserts per call, velocities.jl:199-200), or None; n = number of calls examined."""
        cap = 64
        u, v, n = (C.c_int32 * cap)(), (C.c_int32 * cap)(), C.c_int32(0)
        self.ctx.check(self.lib.otmb_facefluxes_pending_flags(self.ctx.handle, cap, u, v, C.byref(n)))
        self._ff_pending = 0
        self._ff_seq_drained, self._ff_seq = getattr(self, "_ff_seq", [])[-n.value:] if n.value else [], []
        for q in range(n.value):
            if not (u[q] and v[q]):
                return q, n.value
        return None, n.value

    def _check_missing(self):
        bad, n = self._first_missing()
        if bad is not None:
            where = f" (asynchronous step {bad + 1} of {n})" if n > 1 else ""
            raise capi.OtmbError(8, self.lib.otmb_status_string(8).decode() + where, step=bad)

    def finish_facefluxes(self):
        """Drain a pipeline of facefluxes_async calls (no transportmatrix): raises the assertion of the first field
        without any valid value."""
        self._check_missing()
        return self.phi

    def step_async(self, umo, vmo, fill):
        """Enqueue one pass of the hot path (facefluxes -> count -> scan -> fill) without any host synchronisation;
        successive calls pipeline on the stream.  finish() synchronises and raises what the FIRST failing pass found
        (every pass keeps its own error flags on the device; OtmbError.step is its index)."""
        if getattr(self, "_ff_pending", 0) >= self.PIPELINE_DEPTH:
            self.finish()
        out = self.transportmatrix_onepass(self.facefluxes_async(umo, vmo, fill), sync=False)
        self._ff_pending = getattr(self, "_ff_pending", 0) + 1
        return out

    def step_fused_async(self, umo, vmo, fill, out=None):
        """Extension (otmb_step_dev): one pass from (umo, vmo) to the five matrices WITHOUT the six ϕ arrays in memory -- only ϕtop is stored,
        the fill pass re-derives the other five fluxes from umo / vmo where it uses them.  The same matrices bit for bit as step_async, 64
        bytes per cell less HBM traffic.  Not the drop-in path: facefluxesfrommasstransport returns the six arrays."""
        if getattr(self, "_ff_pending", 0) >= self.PIPELINE_DEPTH:
            self.finish()
        if getattr(self, "phi_top", None) is None:
            self.phi_top = self._empty(self.G, torch.float64)
        if out is None and (self.out is None or getattr(self, "_out_cap", None) is None):
            self.out = self.new_output_set()
            self._out_cap = [self.N * k + 1 for k in self.PER_COLUMN_MAX]
        if out is None:
            out = self.out
        if self.wet3d._version != self._wetflags_version:
            self.ctx.check(self.lib.otmb_wetflags_dev(self.ctx.handle, self.wet3d.data_ptr(), self.nx, self.ny, self.nz, self.topology,
                                                      self.wetflags.data_ptr()))
            self._wetflags_version = self.wet3d._version
            self._count_tables()
        self._mask_key = None
        a = self._args([self.phi_top] * 6)
        cp, rv, nz = self._out_ptrs(out)
        caps = (C.c_int64 * 5)(*[self.N * k + 1 for k in self.PER_COLUMN_MAX])
        self._note("_ff_seq")
        self.ctx.check(self.lib.otmb_step_dev(self.ctx.handle, umo.data_ptr(), vmo.data_ptr(), int(umo.dtype == torch.float32), float(fill),
                                              self.wetflags.data_ptr(), self.count_tables.data_ptr(), self.phi_top.data_ptr(), C.byref(a),
                                              C.byref(cp), C.byref(rv), C.byref(nz), C.byref(caps)))
        self._note("_tm_seq")
        self._ff_pending = getattr(self, "_ff_pending", 0) + 1
        return out

    def finish(self):
        """Drain the pipeline: the earliest failing step wins; within a step facefluxes' assertion comes first, as in
        the reference (facefluxes runs before transportmatrix)."""
        bad, n = self._first_missing()
        ff_seq = self._ff_seq_drained
        tm_seq, self._tm_seq = getattr(self, "_tm_seq", []), []
        err = None
        try:
            out = self.result()
        except capi.OtmbError as e:
            err = e
        ff_first = True
        if bad is not None and err is not None and err.step is not None and bad < len(ff_seq) and err.step < len(tm_seq):
            ff_first = ff_seq[bad] < tm_seq[err.step]  # which of the two calls was issued first
        if bad is not None and (err is None or err.step is None or ff_first):
            where = f" (asynchronous step {bad + 1} of {n})" if n > 1 else ""
            raise capi.OtmbError(8, self.lib.otmb_status_string(8).decode() + where, step=bad)
        if err is not None:
            raise err
        return out

    def _args(self, phi):
        a = capi.TmArgs()
        a.nx, a.ny, a.nz = self.nx, self.ny, self.nz
        a.topology, a.upwind, a.n_wet = self.topology, int(self.upwind), self.N
        for k in range(6):
            a.phi[k] = phi[k].data_ptr()
        a.v3d, a.thkcello = self.v3d.data_ptr(), self.thk.data_ptr()
        a.rho = self.rho.data_ptr() if self.rho is not None else None
        a.rho_scalar = self.rho_scalar
        a.lwet3d = self.lwet3d.data_ptr()
        a.lwet = self.lwet.data_ptr()
        for k in range(4):
            a.edge_length[k] = self.edge[k].data_ptr()
            a.dist_nbr[k] = self.dist[k].data_ptr()
        a.area2d, a.zt, a.mlotst = self.area.data_ptr(), self.zt.data_ptr(), self.mlotst.data_ptr()
        a.kappa_h, a.kappa_vml, a.kappa_vdeep = self.kappa
        # ϕ straight from this object's facefluxes and untouched since: hand over its push mask
        fresh = getattr(self, "_mask_key", None) is not None and self._mask_key == self._phi_key(phi)
        a.push_mask = self.push_mask.data_ptr() if fresh else None
        a.only_t = 1 if getattr(self, "only_T", False) else 0  # extension: materialise T alone (outputs of the operators unused)
        if getattr(self, "given", None):
            for m, name in enumerate(MATS):
                if name in self.given:
                    cp, rv, nz = self.given[name]
                    a.given[m].colptr, a.given[m].rowval, a.given[m].nzval, a.given[m].nnz = cp.data_ptr(), rv.data_ptr(), nz.data_ptr(), rv.numel()
            key = self._given_versions()  # the library keys its verdicts to addresses: any in-place edit (torch bumps _version) makes it look again
            if key != self._given_key:
                self.ctx.forget_given()
                self._given_key = key
        return a

    def _out_ptrs(self, out):
        """The three pointer arrays of an output set; NULL for an operator the caller passes (nothing of it is written)."""
        skip = set(getattr(self, "given", None) or ())
        return tuple(capi.ptr_array(5, [None if m in skip else out[m][q].data_ptr() for m in MATS]) for q in range(3))

    def plan(self, phi):
        a = self._args(phi)
        nnz = (C.c_int64 * 5)()
        self.ctx.check(self.lib.otmb_transportmatrix_plan_dev(self.ctx.handle, C.byref(a), C.byref(nnz)))
        self.nnz = [int(x) for x in nnz]
        return self.nnz

    def fill(self):
        """Write the five CSC matrices into device tensors (allocated once per capacity)."""
        if self.out is None or any(self.out[m][1].numel() < self.nnz[k] for k, m in enumerate(MATS)):
            self._out_cap = None
            self.out = {m: (torch.empty(self.N + 1, dtype=torch.int64, device=self.device),
                            torch.empty(max(self.nnz[k], 1) + self.nnz[k] // 64, dtype=torch.int64, device=self.device),
                            torch.empty(max(self.nnz[k], 1) + self.nnz[k] // 64, dtype=torch.float64, device=self.device))
                        for k, m in enumerate(MATS)}
        cp, rv, nz = self._out_ptrs(self.out)
        self.ctx.check(self.lib.otmb_transportmatrix_fill_dev(self.ctx.handle, C.byref(cp), C.byref(rv), C.byref(nz)))
        final = (C.c_int64 * 5)()
        self.ctx.check(self.lib.otmb_transportmatrix_nnz(self.ctx.handle, C.byref(final)))
        self.nnz = [int(x) for x in final]  # T's planned count is an upper bound
        return self.out

    def transportmatrix(self, phi):
        """Two-phase protocol (what a caller that must size its outputs first does): plan, then fill."""
        self.plan(phi)
        return self.fill()

    PER_COLUMN_MAX = (7, 7, 5, 3, 3)  # rows a column of T, Tadv, TκH, TκVML, TκVdeep can hold
    FILL_KERNELS = ("tm_kernel<fill>",)  # the pass that writes the matrices (otmb_kernel_name)

    def new_output_set(self):
        """A set of five CSC output buffers at their upper bound (for transportmatrix_onepass(out=...): a pipeline whose
        steps each keep their own matrices)."""
        cap = [self.N * k + 1 for k in self.PER_COLUMN_MAX]
        if os.environ.get("OTMB_ARENA_WHEN") == "outputs" and float(os.environ.get("OTMB_ARENA_GB", "0")) > 0 and not getattr(self, "_arena_done", False):
            self._arena_done = True  # experiment: only the OUTPUT arrays are carved out of one allocation
            _arena = torch.empty(int(float(os.environ["OTMB_ARENA_GB"]) * 2 ** 30), dtype=torch.uint8, device=self.device)
            del _arena
        return {m: (self._empty(self.N + 1, torch.int64), self._empty(cap[k], torch.int64),
                    self._empty(cap[k], torch.float64)) for k, m in enumerate(MATS)}

    def choose_placement(self, umo, vmo, fill, candidates=4, reps=3, budget_fraction=0.6, min_output_bytes=4 << 30):
        """Set-up time only, never results: pick WHERE this assembler's flux arrays and output matrices live.
        On MI355X the time of the two write-heavy passes depends reproducibly on which device allocations their arrays were given -- inside one
        process the fill pass ran 5.86 ... 6.57 ms (another box: 5.97 ... 7.35 ms) on eight output sets whose virtual layout is identical, every set
        repeating to 0.1 % (profiles/r04/README.md section 12): it is the physical backing of the ~30 concurrently streamed arrays that differs, and
        no per-buffer probe sees it (single streams differ by <= 4 %).  So: allocate `candidates` flux sets and time facefluxes on each, keep the fastest;
        then `candidates` output sets (new_output_set()) and time the fill pass on each with the kept fluxes, keep the fastest; free the rest.
        Nothing is chosen when the candidates would not fit in `budget_fraction` of the free device memory (the 0.1 degree grid), nor on grids whose
        output set is smaller than `min_output_bytes` (1 degree: the candidates differ by 1-2 % there, nothing to choose).  Returns a record of every
        candidate's time (bench.py prints it)."""
        rec = {"candidates": int(candidates), "facefluxes_ms": [], "fill_ms": [], "chosen": None}
        if candidates < 2:
            return rec
        free_b, _ = torch.cuda.mem_get_info(self.device)
        phi_bytes = self.G * (6 * 8 + 2)
        out_bytes = sum((self.N * k + 1) * 16 + (self.N + 1) * 8 for k in self.PER_COLUMN_MAX)
        if out_bytes < min_output_bytes:
            rec["skipped"] = "small grid: placements differ by 1-2 %"
            return rec
        if candidates * (phi_bytes + out_bytes) > budget_fraction * free_b:
            rec["skipped"] = "candidates do not fit"
            return rec

        def timed(kernels, launch):
            """mean duration of the pass `kernels` names"""
            for _ in range(2):
                launch()
            self.ctx.synchronize()
            self.ctx.timing_enable(True)
            for _ in range(reps):
                launch()
            self.ctx.synchronize()
            t = self.ctx.timing_collect()
            self.ctx.timing_enable(False)
            ran = [t[k] for k in kernels if k in t]
            if not ran:
                raise RuntimeError(f"choose_placement: none of {kernels} was launched ({sorted(t)})")
            return max(ms / cnt for ms, cnt in ran)

        phis = [([self._empty(self.G, torch.float64) for _ in range(6)], self._empty(self.G, torch.int16)) for _ in range(candidates)]
        for p, m in phis:
            self.phi, self.push_mask = p, m
            rec["facefluxes_ms"].append(timed(("facefluxes_kernel",), lambda: self.facefluxes_async(umo, vmo, fill)))
        self.finish_facefluxes()
        kp = int(np.argmin(rec["facefluxes_ms"]))
        self.phi, self.push_mask = phis[kp]
        del phis, p, m
        phi = self.facefluxes(umo, vmo, fill)
        outs = [self.new_output_set() for _ in range(candidates)]
        for o in outs:
            rec["fill_ms"].append(timed(self.FILL_KERNELS, lambda: self.transportmatrix_onepass(phi, sync=False, out=o)))
            self.result()
        ko = int(np.argmin(rec["fill_ms"]))
        self.out, self._out_cap = outs[ko], [self.N * k + 1 for k in self.PER_COLUMN_MAX]
        del outs, o
        torch.cuda.empty_cache()  # the candidates that were not kept go back to the driver
        rec["chosen"] = [kp, ko]
        return rec

    def transportmatrix_onepass(self, phi, sync=True, out=None):
        """Asynchronous protocol (otmb_transportmatrix_dev): outputs preallocated at their upper bound, count ->
        scan -> fill enqueued without a host round trip.  With sync=False the nnz/errors are collected later by
        result().  out: an output set of new_output_set() (default: this object's one set, overwritten by every call)."""
        if out is None and (self.out is None or getattr(self, "_out_cap", None) is None):
            self.out = self.new_output_set()
            self._out_cap = [self.N * k + 1 for k in self.PER_COLUMN_MAX]
        if out is None:
            out = self.out
        a = self._args(phi)
        cp, rv, nz = self._out_ptrs(out)
        caps = (C.c_int64 * 5)(*[self.N * k + 1 for k in self.PER_COLUMN_MAX])
        self.ctx.check(self.lib.otmb_transportmatrix_dev(self.ctx.handle, C.byref(a), C.byref(cp), C.byref(rv),
                                                         C.byref(nz), C.byref(caps)))
        self._note("_tm_seq")
        return self.result() if sync else out

    def result(self):
        nnz = (C.c_int64 * 5)()
        rc = self.lib.otmb_transportmatrix_result(self.ctx.handle, C.byref(nnz))
        self._tm_seq = []
        if rc != capi.OK:
            step = C.c_int64(-1)
            self.lib.otmb_transportmatrix_failed_step(self.ctx.handle, C.byref(step))
            raise capi.OtmbError(rc, self.lib.otmb_last_error(self.ctx.handle).decode("utf-8"),
                                 step=int(step.value) if step.value >= 0 else None)
        self.nnz = [int(x) for x in nnz]
        return self.out

    def result_step(self, k):
        """(status, nnz) of the k-th asynchronous call covered by the last result(): every call keeps its own verdict and
        its own nnz (and its own T, compacted if entries cancelled, when it was given its own output arrays)."""
        nnz = (C.c_int64 * 5)()
        rc = self.lib.otmb_transportmatrix_result_step(self.ctx.handle, int(k), C.byref(nnz))
        return rc, [int(x) for x in nnz]

    def step(self, umo, vmo, fill, onepass=True):
        """One pass of the hot path, all device resident: facefluxes -> transportmatrix."""
        phi = self.facefluxes(umo, vmo, fill)
        return self.transportmatrix_onepass(phi) if onepass else self.transportmatrix(phi)

    def lump_and_spray(self, mask=None, di=2, dj=2, dk=1, matrix="T"):
        """lump_and_spray (src/extratools.jl:38-119) on the resident grid and the resident result `matrix` (only its
        pattern is read; the volumes are v3D on the wet cells).  mask: flat uint8/bool device tensor or None.
        Returns device tensors: LUMP (colptr, rowval, nzval), SPRAY (colptr, rowval, nzval), vol_c."""
        k = MATS.index(matrix)
        cp, rv, _ = self.out[matrix]
        m = None
        if mask is not None:
            m = mask.to(self.device).reshape(-1).to(torch.uint8).contiguous()
            if m.numel() != self.G:
                raise ValueError("mask must have one entry per grid cell")
        nc = C.c_int64(0)
        self.ctx.check(self.lib.otmb_lump_and_spray_plan_dev(
            self.ctx.handle, self.wet3d.data_ptr(), None if m is None else m.data_ptr(), self.lwet3d.data_ptr(),
            self.lwet.data_ptr(), self.N, self.nx, self.ny, self.nz, cp.data_ptr(), rv.data_ptr(), int(di), int(dj), int(dk),
            C.byref(nc)))
        Nc, N = int(nc.value), self.N
        vol = self.v3d[self.lwet[:N] - 1].contiguous()
        i64 = lambda n: torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        f64 = lambda n: torch.empty(max(n, 1), dtype=torch.float64, device=self.device)
        L = (i64(N + 1), i64(N), f64(N))
        S = (i64(Nc + 1), i64(N), f64(N))
        vc = f64(Nc)
        self.ctx.check(self.lib.otmb_lump_and_spray_fill_dev(self.ctx.handle, vol.data_ptr(), L[0].data_ptr(), L[1].data_ptr(),
                                                             L[2].data_ptr(), S[0].data_ptr(), S[1].data_ptr(), S[2].data_ptr(),
                                                             vc.data_ptr()))
        return (L[0], L[1][:N], L[2][:N]), (S[0], S[1][:N], S[2][:N]), vc[:Nc]

    def result_to_host(self):
        self.ctx.synchronize()
        if any(self.out[m][1].numel() < self.nnz[k] for k, m in enumerate(MATS)):
            raise RuntimeError("output buffers smaller than nnz")
        res = {}
        for k, m in enumerate(MATS):
            cp, rv, nz = self.out[m]
            res[m] = (cp.cpu().numpy(), rv[: self.nnz[k]].cpu().numpy(), nz[: self.nnz[k]].cpu().numpy())
        return res

    # ---- the reference's two-step formulation (general path) ----------------------------------------------------
    def sparse_entries(self, which, phi=None):
        """COO triplets (I, J, V device tensors) of one operator in the reference's push order
        (advection/horizontal/vertical *_operator_sparse_entries, src/matrixbuilding.jl:221-479).
        which: "Tadv" | "TκH" | "TκVML" | "TκVdeep"."""
        code = {"Tadv": 0, "TκH": 1, "TκVML": 2, "TκVdeep": 3}[which]
        a = self._args(phi if phi is not None else self.phi)
        n = C.c_int64(0)
        self.ctx.check(self.lib.otmb_sparse_entries_plan_dev(self.ctx.handle, code, C.byref(a), C.byref(n)))
        ln = int(n.value)
        I = torch.empty(max(ln, 1), dtype=torch.int64, device=self.device)
        J = torch.empty(max(ln, 1), dtype=torch.int64, device=self.device)
        V = torch.empty(max(ln, 1), dtype=torch.float64, device=self.device)
        self.ctx.check(self.lib.otmb_sparse_entries_fill_dev(self.ctx.handle, I.data_ptr(), J.data_ptr(), V.data_ptr()))
        return I[:ln], J[:ln], V[:ln]

    def sparse(self, I, J, V, m, n):
        """SparseArrays.sparse(I, J, V, m, n) on the device -> (colptr, rowval, nzval) tensors."""
        I, J, V = I.contiguous(), J.contiguous(), V.contiguous()
        nnz = C.c_int64(0)
        self.ctx.check(self.lib.otmb_sparse_plan_dev(self.ctx.handle, I.data_ptr(), J.data_ptr(), V.data_ptr(), I.numel(), m, n,
                                                     C.byref(nnz)))
        k = int(nnz.value)
        cp = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        rv = torch.empty(max(k, 1), dtype=torch.int64, device=self.device)
        nz = torch.empty(max(k, 1), dtype=torch.float64, device=self.device)
        self.ctx.check(self.lib.otmb_sparse_fill_dev(self.ctx.handle, cp.data_ptr(), rv.data_ptr(), nz.data_ptr()))
        self.ctx.synchronize()
        return cp, rv[:k], nz[:k]

    def spadd(self, A, B, n):
        """A + B on the device (SparseArrays' map(+): union pattern, exact-zero sums dropped; src/matrixbuilding.jl:147).
        A, B: (colptr, rowval, nzval) device tensors of n columns; returns the same triple."""
        Ap, Ai, Ax = (t.contiguous() for t in A)
        Bp, Bi, Bx = (t.contiguous() for t in B)
        nnz = C.c_int64(0)
        self.ctx.check(self.lib.otmb_spadd_plan_dev(self.ctx.handle, n, Ap.data_ptr(), Ai.data_ptr(), Ax.data_ptr(), Bp.data_ptr(),
                                                    Bi.data_ptr(), Bx.data_ptr(), C.byref(nnz)))
        k = int(nnz.value)
        Cp = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        Ci = torch.empty(max(k, 1), dtype=torch.int64, device=self.device)
        Cx = torch.empty(max(k, 1), dtype=torch.float64, device=self.device)
        self.ctx.check(self.lib.otmb_spadd_fill_dev(self.ctx.handle, n, Ap.data_ptr(), Ai.data_ptr(), Ax.data_ptr(), Bp.data_ptr(),
                                                    Bi.data_ptr(), Bx.data_ptr(), Cp.data_ptr(), Ci.data_ptr(), Cx.data_ptr()))
        self.ctx.synchronize()
        return Cp, Ci[:k], Cx[:k]

    # ---- accounting ---------------------------------------------------------------------------
    def algorithmic_bytes(self):
        """bytes the assembly must move with all five matrices returned
        (inputs read once, outputs written once; intermediate traffic is overhead and not counted)."""
        r, w = self.algorithmic_bytes_split()
        return r + w

    def algorithmic_bytes_split(self):
        """(bytes read, bytes written) of algorithmic_bytes().  An operator the caller passes and the fill pass re-derives (set_given) is
        neither read nor written: its 16 nnz + 8 (N + 1) bytes are not part of the pass."""
        n3d = 9 + (1 if self.rho is not None else 0)
        skip = set(getattr(self, "given", None) or ())
        return (8 * self.G * n3d + 80 * self.nx * self.ny + 8 * self.nz,
                sum(16 * z + 8 * (self.N + 1) for m, z in zip(MATS, self.nnz) if m not in skip))

    def fill_pass_stream_mix(self):
        """What an ideal streaming kernel reaches over the fill pass's OWN arrays (otmb_ctx_stream_mix): its ten 3-D inputs (+ the 2-D
        metrics) read once, its fifteen output arrays written once at their actual lengths, in as many slices as the pass has tiles.
        DESTROYS the matrices of self.out: call it after the results have been used.  {columns per slice: GB/s}."""
        b8 = lambda t, n=None: (t.data_ptr(), 8 * (t.numel() if n is None else n))
        ins = [b8(p) for p in self.phi] + [b8(self.v3d), b8(self.thk), b8(self.lwet3d)] + ([b8(self.rho)] if self.rho is not None else [])
        ins += [b8(t) for t in (*self.edge, *self.dist, self.area, self.mlotst)]
        outs = []
        for k, m in enumerate(MATS):
            cp, rv, nz = self.out[m]
            outs += [b8(cp, self.N + 1), b8(rv, self.nnz[k]), b8(nz, self.nnz[k])]
        # the fill pass's own granularity (256 columns per slice) and longer slices: a plain stream likes them longer where the grid is large
        # enough to still fill the chip (profiles/r05: 4.3 / 4.3 / 3.9 / 3.7 TB/s at 1 degree, 4.9 / 5.4 / 5.6 / 5.5 TB/s at 0.25 degree)
        return {cols: self.ctx.stream_mix(ins, outs, max(8, self.N // cols)) for cols in (256, 512, 1024, 2048)}

    def facefluxes_bytes(self, itemsize=8):
        """umo, vmo, wet3D read once; six ϕ arrays written once."""
        return self.G * (2 * itemsize + 1 + 6 * 8)
