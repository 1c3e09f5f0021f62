"""Host-side mirror of the reference's API for the hot path -- same function names, keyword
arguments, return fields and error messages as the Julia originals, over the C ABI
(HOST-pointer entry points, i.e. exactly what julia/OceanTransportMatrixBuilderAMD.jl does
with ccall).  Arrays are numpy, Fortran-ordered, Julia shapes; indices stay 1-based.

    makeindices(v3D)                                        src/matrixbuilding.jl:10-24
    facefluxesfrommasstransport(; umo, vmo, gridmetrics, indices)   src/velocities.jl:118-130
    facefluxes(umo, vmo, gridmetrics, indices; FillValue)           src/velocities.jl:190-255
    transportmatrix(; ϕ, mlotst, gridmetrics, indices, ρ, κH, κVML, κVdeep, upwind, ...)
                                                            src/matrixbuilding.jl:128-150
"""
import ctypes as C
import os
import threading

import numpy as np

from . import capi
from ._nt import NT, data_and_props
from .capi import HDIRS, MATS, PHI_ORDER

_ctx = {}
_mgpu = {}
last_call_seconds = {}  # time spent inside the C ABI by the last transportmatrix call: {"plan": s, "fetch": s} (bench.py)


def context(device=0):
    if device not in _ctx:
        _ctx[device] = capi.Context(device)
    return _ctx[device]


def mgpu(devices):
    """The otmb_mgpu of a device list (`devices = 0:7` in the Julia shim): created once, kept."""
    key = tuple(int(d) for d in devices)
    if key not in _mgpu:
        _mgpu[key] = capi.Mgpu(key)
    return _mgpu[key]


def _f64(a):
    return np.asfortranarray(a, dtype=np.float64)


def _topology_kind(gridmetrics):
    t = gridmetrics["gridtopology"]
    return int(t["kind"]) if isinstance(t, dict) else int(t)


class SparseMatrixCSC:
    """Julia's SparseMatrixCSC{Float64,Int64} layout: m, n, colptr (n+1), rowval, nzval; 1-based."""

    def __init__(self, m, n, colptr, rowval, nzval):
        self.m, self.n, self.colptr, self.rowval, self.nzval = m, n, colptr, rowval, nzval

    @property
    def shape(self):
        return (self.m, self.n)

    @property
    def nnz(self):
        return len(self.rowval)

    def to_scipy(self):
        import scipy.sparse as sp

        return sp.csc_matrix((self.nzval, self.rowval - 1, self.colptr - 1), shape=(self.m, self.n))

    def __iter__(self):  # (colptr, rowval, nzval)
        return iter((self.colptr, self.rowval, self.nzval))


def spadd(A, B, *, device=0):
    """A + B with SparseArrays' semantics (union pattern, exact-zero results dropped), on the device."""
    ctx = context(device)
    n = A.n
    assert A.shape == B.shape
    arrs = [np.ascontiguousarray(x, dtype=t) for x, t in ((A.colptr, np.int64), (A.rowval, np.int64), (A.nzval, np.float64),
                                                          (B.colptr, np.int64), (B.rowval, np.int64), (B.nzval, np.float64))]
    cap = max(1, A.nnz + B.nnz)
    Cp = np.empty(n + 1, dtype=np.int64)
    Ci = np.empty(cap, dtype=np.int64)
    Cx = np.empty(cap, dtype=np.float64)
    nnz = C.c_int64(0)
    ctx.check(capi.lib().otmb_spadd(ctx.handle, n, *[x.ctypes.data for x in arrs], Cp.ctypes.data, Ci.ctypes.data,
                                    Cx.ctypes.data, C.byref(nnz)))
    return SparseMatrixCSC(A.m, n, Cp, Ci[: nnz.value].copy(), Cx[: nnz.value].copy())


def makeindices(v3D, *, device=0):
    """matrixbuilding.jl:10-24 -> NT(wet3D, L, Lwet, N, Lwet3D, C).  Lwet3D uses 0 for `missing`;
    L and C (lazy Linear/CartesianIndices in Julia) are the grid shape here."""
    ctx = context(device)
    v = _f64(v3D)
    nx, ny, nz = v.shape
    G = v.size
    lwet3d = np.empty(v.shape, dtype=np.int64, order="F")
    lwet = np.empty(G, dtype=np.int64)
    wet3d = np.empty(v.shape, dtype=np.uint8, order="F")
    n = C.c_int64(0)
    ctx.check(capi.lib().otmb_makeindices(ctx.handle, v.ctypes.data, nx, ny, nz, lwet3d.ctypes.data, lwet.ctypes.data,
                                          wet3d.ctypes.data, C.byref(n)))
    N = int(n.value)
    return NT(wet3D=wet3d.view(np.bool_), L=v.shape, Lwet=lwet[:N].copy(), N=N, Lwet3D=lwet3d, C=v.shape)


PINNED_OUTPUTS = True  # results live in pinned host memory owned by the context (otmb_host_alloc): no staging copy, no page faults


def _out_array(ctx, shape, dtype):
    """Output array of a host-pointer call: pinned memory of the context when it can be had (the library DMAs straight into
    it; the block returns to the context's pool when the array is garbage collected), ordinary memory otherwise."""
    if PINNED_OUTPUTS:
        try:
            return ctx.pinned_empty(shape, dtype)
        except capi.OtmbError:
            pass
    return np.empty(shape, dtype=dtype, order="F")


def facefluxes(umo, vmo, gridmetrics, indices, *, FillValue, device=0, devices=None):
    """velocities.jl:190-255.  umo/vmo are not modified (the reference mutates its converted copies).
    devices=[...] (extension): depth slabs over several GPUs (otmb_mgpu_facefluxes), same six arrays bit for bit."""
    ctx = context(device if devices is None else list(devices)[0])
    u = np.asarray(umo)
    v = np.asarray(vmo)
    is32 = u.dtype == np.float32 and v.dtype == np.float32
    dt = np.float32 if is32 else np.float64
    u = np.asfortranarray(u, dtype=dt)
    v = np.asfortranarray(v, dtype=dt)
    nx, ny, nz = u.shape
    wet = np.asfortranarray(indices["wet3D"]).view(np.uint8)
    out = {k: _out_array(ctx, u.shape, np.float64) for k in PHI_ORDER}  # pinned: the DMA writes the results in place
    ptrs = capi.ptr_array(6, [out[k].ctypes.data for k in PHI_ORDER])
    if devices is not None:
        return _facefluxes_mgpu(devices, u, v, is32, wet, FillValue, gridmetrics, out, ptrs)
    ctx.check(capi.lib().otmb_facefluxes(ctx.handle, u.ctypes.data, v.ctypes.data, int(is32), wet.ctypes.data,
                                         float(FillValue), nx, ny, nz, _topology_kind(gridmetrics), C.byref(ptrs)))
    return NT(**out)


def _facefluxes_mgpu(devices, u, v, is32, wet, FillValue, gridmetrics, out, ptrs):
    mg = mgpu(devices)
    nx, ny, nz = u.shape
    mg.check(capi.lib().otmb_mgpu_facefluxes(mg.handle, u.ctypes.data, v.ctypes.data, int(is32), wet.ctypes.data,
                                             float(FillValue), nx, ny, nz, _topology_kind(gridmetrics), C.byref(ptrs)))
    return NT(**out)


def facefluxesfrommasstransport(*, umo, vmo, gridmetrics, indices, device=0, devices=None):
    """velocities.jl:118-130."""
    u, up = data_and_props(umo)
    v, vp = data_and_props(vmo)
    fill = up["_FillValue"]
    fv = vp["_FillValue"]
    assert (fill == fv) or (np.isnan(fill) and np.isnan(fv))  # @assert isequal(...), :121
    return facefluxes(u, v, gridmetrics, indices, FillValue=fill, device=device, devices=devices)


def _velocity_flux(which, a_i, a_j, gridmetrics, rho, device):
    ctx = context(device)
    x = np.asarray(data_and_props(a_i)[0])
    y = np.asarray(data_and_props(a_j)[0])
    is32 = x.dtype == np.float32 and y.dtype == np.float32
    dt = np.float32 if is32 else np.float64
    x = np.asfortranarray(x, dtype=dt)
    y = np.asfortranarray(y, dtype=dt)
    nx, ny, nz = x.shape
    thk = _f64(gridmetrics["thkcello"])
    ee = _f64(gridmetrics["edge_length_2D"]["east"])
    en = _f64(gridmetrics["edge_length_2D"]["north"])
    if np.ndim(rho) == 0:
        rp, rs, keep = None, float(rho), None
    else:
        keep = _f64(rho)
        rp, rs = keep.ctypes.data, 0.0
    oi = np.empty(x.shape, dtype=np.float64, order="F")
    oj = np.empty(x.shape, dtype=np.float64, order="F")
    fn = getattr(capi.lib(), which)
    ctx.check(fn(ctx.handle, x.ctypes.data, y.ctypes.data, int(is32), rp, rs, thk.ctypes.data, ee.ctypes.data, en.ctypes.data,
                 nx, ny, nz, _topology_kind(gridmetrics), oi.ctypes.data, oj.ctypes.data))
    return oi, oj


def interpolateontodefaultCgrid(u, u_lon, u_lat, v, v_lon, v_lat, gridmetrics, *, device=0):
    """src/gridcellgeometry.jl:103-140: C-grid fields pass through; B-grid fields whose velocity points sit on the NE
    corner are averaged onto the east/north faces (on the device); A-grid raises as in the reference."""
    from .gridmetrics import getarakawagrid, midpointonsphere

    kind, u_pos, v_pos = getarakawagrid(u_lon, u_lat, v_lon, v_lat, gridmetrics)
    if kind == "C":
        return u, u_lon, u_lat, v, v_lon, v_lat
    if kind == "A":
        raise RuntimeError("Interpolation not implemented for A-grid type")
    if not (u_pos == v_pos == "NE"):
        raise RuntimeError(f"Interpolation not implemented for this B-grid({u_pos},{v_pos}) type")  # :109
    ud, up = data_and_props(u)
    vd, _ = data_and_props(v)
    fill = up["_FillValue"]  # :111
    x = np.asarray(ud)
    y = np.asarray(vd)
    is32 = x.dtype == np.float32 and y.dtype == np.float32
    dt = np.float32 if is32 else np.float64
    x = np.asfortranarray(x, dtype=dt)
    y = np.asfortranarray(y, dtype=dt)
    nx, ny, nz = x.shape
    u2 = np.empty(x.shape, dtype=np.float64, order="F")
    v2 = np.empty(x.shape, dtype=np.float64, order="F")
    ctx = context(device)
    ctx.check(capi.lib().otmb_bgrid_to_cgrid(ctx.handle, x.ctypes.data, y.ctypes.data, int(is32), float(fill), nx, ny, nz,
                                             u2.ctypes.data, v2.ctypes.data))
    lv, tv = gridmetrics["lon_vertices"], gridmetrics["lat_vertices"]
    u2_lon, u2_lat = midpointonsphere(lv[2], tv[2], lv[1], tv[1])  # :132  (NE, SE)
    v2_lon, v2_lat = midpointonsphere(lv[3], tv[3], lv[2], tv[2])  # :133  (NW, NE)
    return u2, u2_lon, u2_lat, v2, v2_lon, v2_lat


def velocity2fluxes(u, u_lon, u_lat, v, v_lon, v_lat, gridmetrics, ρ, *, device=0):
    """src/velocities.jl:10-39 -> (ϕᵢ, ϕⱼ)."""
    u, _, _, v, _, _ = interpolateontodefaultCgrid(u, u_lon, u_lat, v, v_lon, v_lat, gridmetrics, device=device)
    return _velocity_flux("otmb_velocity2fluxes", u, v, gridmetrics, ρ, device)


def fluxes2velocity(ϕᵢ, ϕⱼ, gridmetrics, ρ, *, device=0):
    """src/velocities.jl:50-74 -> (u, v) on the C-grid."""
    return _velocity_flux("otmb_fluxes2velocity", ϕᵢ, ϕⱼ, gridmetrics, ρ, device)


def facefluxesfromvelocities(*, uo, uo_lon, uo_lat, vo, vo_lon, vo_lat, gridmetrics, indices, ρ=None, rho=None, device=0):
    """src/velocities.jl:140-151."""
    rho = ρ if ρ is not None else rho
    _, up = data_and_props(uo)
    _, vp = data_and_props(vo)
    fill = up["_FillValue"]
    fv = vp["_FillValue"]
    assert (fill == fv) or (np.isnan(fill) and np.isnan(fv))  # :143
    umo, vmo = velocity2fluxes(uo, uo_lon, uo_lat, vo, vo_lon, vo_lat, gridmetrics, rho, device=device)
    return facefluxes(umo, vmo, gridmetrics, indices, FillValue=fill, device=device)


def makegridmetrics_gpu(*, areacello, volcello, lon, lat, lev, lon_vertices, lat_vertices, device=0):
    """makegridmetrics (src/gridcellgeometry.jl:265-311) with the array work done by the library on host arrays (otmb_makegridmetrics):
    what `makegridmetrics(...; gpu = true)` of the Julia shim runs.  Vertex permutation and topology test are host decisions, as there.
    Same 13-field result as gridmetrics.makegridmetrics; the haversine distances agree to 1e-12, everything else bit for bit."""
    from . import gridtopology as gt
    from ._nt import NT
    from .gridmetrics import vertexpermutation

    area, ap = data_and_props(areacello)
    vol, vp = data_and_props(volcello)
    area = _f64(area)
    vol = _f64(vol)
    nx, ny, nz = vol.shape
    lonv = _f64(data_and_props(lon_vertices)[0])
    latv = _f64(data_and_props(lat_vertices)[0])
    lon2 = _f64(data_and_props(lon)[0])
    lat2 = _f64(data_and_props(lat)[0])
    zt = np.array(data_and_props(lev)[0], dtype=np.float64)
    perm = vertexpermutation(lonv, latv)
    lonv_p = np.asfortranarray(lonv[perm, :, :])
    latv_p = np.asfortranarray(latv[perm, :, :])
    topo = gt.getgridtopology(lonv_p, latv_p)
    f2 = lambda: np.empty((nx, ny), dtype=np.float64, order="F")
    f3 = lambda: np.empty((nx, ny, nz), dtype=np.float64, order="F")
    area2D, v3D, thk, Z3D = f2(), f3(), f3(), f3()
    el, de, dn = [f2() for _ in range(4)], [f2() for _ in range(4)], [f2() for _ in range(4)]
    ctx = context(device)
    pa = (C.c_int32 * 4)(*[int(q) for q in perm])
    ctx.check(capi.lib().otmb_makegridmetrics(
        ctx.handle, vol.ctypes.data, area.ctypes.data, float(ap.get("_FillValue", np.nan)), float(vp.get("_FillValue", np.nan)),
        lon2.ctypes.data, lat2.ctypes.data, lonv.ctypes.data, latv.ctypes.data, C.byref(pa), nx, ny, nz, int(topo),
        area2D.ctypes.data, v3D.ctypes.data, thk.ctypes.data, Z3D.ctypes.data,
        C.byref(capi.ptr_array(4, [a.ctypes.data for a in el])), C.byref(capi.ptr_array(4, [a.ctypes.data for a in de])),
        C.byref(capi.ptr_array(4, [a.ctypes.data for a in dn]))))
    bydir = lambda arrs: {d: arrs[k] for k, d in enumerate(capi.HDIRS)}
    return NT(area2D=area2D, v3D=v3D, thkcello=thk, lon_vertices=lonv_p, lat_vertices=latv_p, lon=lon2, lat=lat2, Z3D=Z3D, zt=zt,
              edge_length_2D=bydir(el), distance_to_edge_2D=bydir(de), distance_to_neighbour_2D=bydir(dn), gridtopology=NT(kind=int(topo), name=gt.NAMES[int(topo)], nx=nx, ny=ny, nz=len(zt)))


def bolus_GM_velocity(ρ, gridmetrics, indices, *, κGM=600, maxslope=0.01, device=0):
    """src/RediGM.jl:46-79 -> (u, v).  Experimental in the reference; parity unpinned (oracle only)."""
    ctx = context(device)
    rho = _f64(ρ)
    nx, ny, nz = rho.shape
    z3d = _f64(gridmetrics["Z3D"])
    wet = np.asfortranarray(indices["wet3D"]).view(np.uint8)
    de = _f64(gridmetrics["distance_to_neighbour_2D"]["east"])
    dn = _f64(gridmetrics["distance_to_neighbour_2D"]["north"])
    u = np.empty(rho.shape, dtype=np.float64, order="F")
    v = np.empty(rho.shape, dtype=np.float64, order="F")
    ctx.check(capi.lib().otmb_bolus_gm_velocity(ctx.handle, rho.ctypes.data, z3d.ctypes.data, wet.ctypes.data, de.ctypes.data,
                                                dn.ctypes.data, nx, ny, nz, _topology_kind(gridmetrics), float(κGM),
                                                float(maxslope), u.ctypes.data, v.ctypes.data))
    return u, v


def _tm_args(phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, keep, grid_passthrough=None, given=None):
    """grid_passthrough (list): receives False for every grid-constant array that had to be converted (copied) on the way
    to the C ABI -- a temporary's address says nothing about its content, so reuse_grid must not rely on it.
    given: {name: SparseMatrixCSC or None} -- operators the caller passes (otmb_tm_args.given)."""
    if grid_passthrough is None:
        grid_passthrough = []

    def grid(x, conv):
        y = conv(x)
        grid_passthrough.append(y is x)
        return y

    v3d = grid(gridmetrics["v3D"], _f64)
    nx, ny, nz = v3d.shape
    a = capi.TmArgs()
    a.nx, a.ny, a.nz = nx, ny, nz
    a.topology = _topology_kind(gridmetrics)
    a.upwind = int(bool(upwind))
    a.n_wet = int(indices["N"])

    def hold(x):
        keep.append(x)
        return x.ctypes.data

    for k, name in enumerate(PHI_ORDER):
        a.phi[k] = hold(_f64(phi[name]))
    a.v3d = hold(v3d)
    a.thkcello = hold(grid(gridmetrics["thkcello"], _f64))
    if np.ndim(rho) == 0:
        a.rho = None
        a.rho_scalar = float(rho)
    else:
        r = _f64(rho)
        assert r.shape == v3d.shape
        a.rho = hold(r)
    a.lwet3d = hold(grid(indices["Lwet3D"], lambda x: np.asfortranarray(x, dtype=np.int64)))
    a.lwet = hold(grid(indices["Lwet"], lambda x: np.ascontiguousarray(x, dtype=np.int64)))
    for k, d in enumerate(HDIRS):
        a.edge_length[k] = hold(grid(gridmetrics["edge_length_2D"][d], _f64))
        a.dist_nbr[k] = hold(grid(gridmetrics["distance_to_neighbour_2D"][d], _f64))
    a.area2d = hold(grid(gridmetrics["area2D"], _f64))
    a.zt = hold(grid(gridmetrics["zt"], lambda x: np.ascontiguousarray(x, dtype=np.float64)))
    ml, _ = data_and_props(mlotst)
    a.mlotst = hold(_f64(ml))
    a.kappa_h, a.kappa_vml, a.kappa_vdeep = float(kH), float(kVML), float(kVdeep)
    N = int(indices["N"])
    for m, name in enumerate(MATS):
        A = None if given is None else given.get(name)
        if A is None:
            continue
        if A.shape != (N, N):
            raise ValueError(f"{name} is {A.shape[0]}x{A.shape[1]}, expected {N}x{N}")
        # (TκH / TκVdeep are grid constants of a time loop: their arrays fall under the reuse_grid promise like gridmetrics / indices)
        a.given[m].colptr = hold(grid(A.colptr, lambda x: np.ascontiguousarray(x, dtype=np.int64)))
        a.given[m].rowval = hold(grid(A.rowval, lambda x: np.ascontiguousarray(x, dtype=np.int64)))
        a.given[m].nzval = hold(grid(A.nzval, lambda x: np.ascontiguousarray(x, dtype=np.float64)))
        a.given[m].nnz = int(len(A.rowval))
    return a


def transportmatrix(*, ϕ=None, phi=None, mlotst, gridmetrics, indices, ρ=None, rho=None, κH=500.0, κVML=0.1,
                    κVdeep=1.0e-5, kappaH=None, kappaVML=None, kappaVdeep=None, Tadv=None, TκH=None, TκVML=None,
                    TκVdeep=None, upwind=True, operators=True, reuse_grid=False, reuse_fluxes=False, device=0, devices=None, slabs=None):
    """matrixbuilding.jl:128-150 -> NT(T, Tadv, TκH, TκVML, TκVdeep), each a SparseMatrixCSC.
    ASCII aliases (phi, rho, kappaH, ...) are accepted beside the reference's Unicode keywords.
    operators=False (extension; the reference always returns all five): only T is materialised, the other four come
    back as None -- the same T, half the bytes written and a third of the bytes copied back to the host.
    reuse_grid=True (extension): the caller promises that the gridmetrics / indices arrays are the very arrays of the
    previous call, unmodified (a loop over time slices): they are not copied to the device again (otmb_ctx_set_reuse_grid).
    reuse_fluxes=True (extension): ϕ is what facefluxes* returned last on this device, unmodified: its device copy is used
    (otmb_ctx_set_reuse_fluxes).
    devices=[0, 1, ...] (extension): the grid is cut into depth slabs, one per listed GPU of this process, each moved over its own
    PCIe link (otmb_mgpu_transportmatrix_plan / _fetch); the same five matrices bit for bit.
    slabs=S (extension, speed only): the pipelined one-phase build (otmb_mgpu_transportmatrix_onepass) on S depth slabs -- of `device`, or one per
    entry of `devices` when that is given (S is then ignored): a slab uploads while the one above it copies its columns home, so the
    link carries both directions at once.  The matrices' arrays are views of upper-bound allocations (7N, 7N, 5N, 3N, 3N entries).
    slabs=None (default): default_slabs() -- 4 on large grids, the two-phase call (slabs=0) otherwise."""
    phi = ϕ if ϕ is not None else phi
    rho = ρ if ρ is not None else rho
    kH = κH if kappaH is None else kappaH
    kVML = κVML if kappaVML is None else kappaVML
    kVdeep = κVdeep if kappaVdeep is None else kappaVdeep
    given = dict(Tadv=Tadv, TκH=TκH, TκVML=TκVML, TκVdeep=TκVdeep)
    if all(x is None for x in given.values()):
        given = None
    else:
        # matrixbuilding.jl:140-143: an operator that is passed in is NOT built -- nothing it alone would read is read (ϕ / ρ for Tadv,
        # mlotst for TκVML: harmless stand-ins take their place), it is returned as the very object passed (:149), and
        # T = ((Tadv + TκH) + TκVML) + TκVdeep (:147) is formed with it (otmb_tm_args.given)
        shape = np.asarray(gridmetrics["v3D"]).shape
        if given["Tadv"] is not None:
            z = np.zeros(shape, dtype=np.float64, order="F")
            phi, rho = {k: z for k in PHI_ORDER}, 1035.0
        if mlotst is None:
            mlotst = np.full(shape[:2], np.nan)
        operators = True  # (the built operators are operands of T and are returned)
    common = (phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, operators)
    fkey = _foreign_key(given)
    if fkey in _foreign_seen and devices is None:  # known to need the sparse-add path: straight to the call that has it
        return _transportmatrix_fused(*common, reuse_grid, reuse_fluxes, device, 0, None, given)
    try:
        trial = None
        if slabs is None:
            slabs = default_slabs(int(indices["N"]), int(np.asarray(gridmetrics["v3D"]).shape[2]), reuse_fluxes, devices)
            if slabs:
                trial = Trial.of(int(device), int(indices["N"]), operators, np.ndim(rho) != 0, reuse_grid, given)
                slabs = slabs if trial.pipelined() else 0
        if trial is not None:
            import time as _time

            t0 = _time.perf_counter()
            tm = (_transportmatrix_onepass(*common, [int(device)] * slabs, 0, reuse_grid, reuse_fluxes, given) if slabs else
                  _transportmatrix_fused(*common, reuse_grid, reuse_fluxes, device, 0, None, given))
            trial.record(_time.perf_counter() - t0)
            return tm
        if slabs:
            nz_levels = int(np.asarray(gridmetrics["v3D"]).shape[2])
            devs = list(devices) if devices is not None else [int(device)] * max(1, min(int(slabs), nz_levels))
            return _transportmatrix_onepass(*common, devs, 0, reuse_grid, reuse_fluxes, given)
        return _transportmatrix_fused(*common, reuse_grid, reuse_fluxes, device, 0, devices, given)
    except capi.OtmbError as e:
        if e.status != capi.GIVEN_FOREIGN or given is None:
            raise
        # a given operator does not have the rows the library derives for this grid (another pattern, Tadv, TκVML): T is then the
        # device sparse add of four materialised operands, which the single-context two-phase call does
        _foreign_seen.add(fkey)
        return _transportmatrix_fused(*common, False, False, device if devices is None else list(devices)[0], 0, None, given)


_foreign_seen = set()  # given-operator sets that the pipelined / multi-slab builds refused (OTMB_ERR_GIVEN_FOREIGN): next time straight to the two-phase call


def _foreign_key(given):
    if given is None:
        return None
    return tuple((name, A.nzval.ctypes.data, len(A.rowval)) for name, A in given.items() if A is not None and hasattr(A.nzval, "ctypes"))


def _result(N, colptr, rowval, nzval, final, operators, given, skip_ops=0):
    """NT(T, Tadv, TκH, TκVML, TκVdeep): the built matrices wrapped at their final counts, a given operator as the very object passed
    (matrixbuilding.jl:149), None for what was not asked for (operators = False)."""
    out = {}
    for m, name in enumerate(MATS):
        if given is not None and given.get(name) is not None:
            out[name] = given[name]
        elif (operators or m == 0) and not (skip_ops >> m) & 1:
            out[name] = SparseMatrixCSC(N, N, colptr[m], rowval[m][: final[m]], nzval[m][: final[m]])
        else:
            out[name] = None
    return NT(**out)


def _wanted(m, operators, given, skip_ops=0):
    return (operators or m == 0) and not (given is not None and given.get(MATS[m]) is not None) and not (skip_ops >> m) & 1


def _transportmatrix_fused(phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, operators, reuse_grid, reuse_fluxes, device,
                           ignore_ops, devices=None, given=None, skip_ops=0):
    """otmb_ctx_set_reuse_grid -> otmb_ctx_set_reuse_fluxes -> otmb_transportmatrix_plan -> otmb_transportmatrix_fetch."""
    if devices is not None:
        return _transportmatrix_mgpu(phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, operators, devices, ignore_ops,
                                     reuse_grid, reuse_fluxes, given)
    ctx = context(device)
    keep, passthrough = [], []
    a = _tm_args(phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, keep, passthrough, given)
    reuse_grid = _reuse_grid_for(device, "ctx", reuse_grid)
    ctx.set_reuse_grid(bool(reuse_grid) and all(passthrough))  # converted temporaries have no identity to rely on
    ctx.set_reuse_fluxes(bool(reuse_fluxes))
    a.only_t = 0 if operators else 1
    a.ignore_ops = int(ignore_ops)
    a.skip_ops = int(skip_ops)
    nnz = (C.c_int64 * 5)()
    import time as _time

    t0 = _time.perf_counter()
    ctx.check(capi.lib().otmb_transportmatrix_plan(ctx.handle, C.byref(a), C.byref(nnz)))
    last_call_seconds["plan"] = _time.perf_counter() - t0
    N = int(indices["N"])
    t0 = _time.perf_counter()
    want = [_wanted(m, operators, given, skip_ops) for m in range(5)]
    colptr = [_out_array(ctx, N + 1, np.int64) if want[m] else None for m in range(5)]
    rowval = [_out_array(ctx, int(nnz[m]), np.int64) if want[m] else None for m in range(5)]
    nzval = [_out_array(ctx, int(nnz[m]), np.float64) if want[m] else None for m in range(5)]
    last_call_seconds["alloc"] = _time.perf_counter() - t0
    ptrs = lambda arrs: capi.ptr_array(5, [None if x is None else x.ctypes.data for x in arrs])
    cp, rv, nz = ptrs(colptr), ptrs(rowval), ptrs(nzval)
    final = (C.c_int64 * 5)()
    t0 = _time.perf_counter()
    ctx.check(capi.lib().otmb_transportmatrix_fetch(ctx.handle, C.byref(cp), C.byref(rv), C.byref(nz), C.byref(final)))
    last_call_seconds["fetch"] = _time.perf_counter() - t0
    ctx.set_reuse_fluxes(False)
    # plan's count for T is the union-pattern bound; entries that summed to exactly zero are dropped (:147)
    return _result(N, colptr, rowval, nzval, final, operators, given, skip_ops)


def default_slabs(N, nz, reuse_fluxes, devices):
    """slabs=None: the pipelined one-phase build on 4 slabs of the device for grids where the transfers dominate (2^18 ... 2^25 wet cells, 8
    levels and more) -- unless the fluxes are promised to be resident on the single-GPU context (reuse_fluxes: nothing to upload beside
    the download then, the two-phase call is as fast) or a device list was given (`devices` keeps the two-phase protocol unless slabs is
    set).  ENV OTMB_HOST_SLABS overrides the 4 (0: always two-phase)."""
    if devices is not None or reuse_fluxes:
        return 0
    s = int(os.environ.get("OTMB_HOST_SLABS", "4"))
    # (round 6: no upper limit any more -- the result arrays are sized from the wet mask and the previous slice's counts, ~6 % over what is used,
    # where rounds 4-5 pinned 7N / 7N / 5N / 3N / 3N entries, +30 %, and left grids above 2^25 wet cells to the two-phase call)
    return s if (s > 0 and N >= (1 << 18) and nz >= 2 * s) else 0


class Trial:
    """The default call's choice between the pipelined and the two-phase protocol is MEASURED, because it depends on the host: the pipelined
    build needs the link to carry both directions at once and a few free host threads; where it does not get them it has been seen slower
    than the two-phase call (27.9 against 23.6 ms; usually 20 against 25: profiles/r05/README.md 9d).  One trial per KIND of call -- device,
    grid size, operators or T alone, scalar or 3-D ρ, the reuse_grid promise, which operators are passed in -- since each of those changes the
    bytes either protocol moves.  Schedule: calls 1-2 pipelined (they allocate), 3-4 pipelined and timed, 5 two-phase (allocates), 6-7
    two-phase and timed; from call 8 on whichever was faster (the minimum of its two samples).  The verdict is not for life: every
    REMEASURE_EVERY-th call runs the protocol that lost and refreshes its time, and a chosen protocol that takes more than SLOW_FACTOR x its
    recorded time SLOW_STREAK calls in a row (a host that got busy) starts the trial over.  Thread-safe.  An explicit slabs= bypasses it."""
    _all = {}
    _lock = threading.Lock()
    REMEASURE_EVERY, SLOW_FACTOR, SLOW_STREAK = 64, 1.3, 3

    def __init__(self):
        self.n, self.t, self.now, self.slow = 0, {True: None, False: None}, True, 0

    @classmethod
    def key(cls, device, N, operators=True, rho3d=False, reuse_grid=False, given=None):
        g = tuple(name for name in MATS if given is not None and given.get(name) is not None)
        return (int(device), int(N), bool(operators), bool(rho3d), bool(reuse_grid), g)

    @classmethod
    def of(cls, device, N, operators=True, rho3d=False, reuse_grid=False, given=None):
        with cls._lock:
            return cls._all.setdefault(cls.key(device, N, operators, rho3d, reuse_grid, given), cls())

    @classmethod
    def peek(cls, device, N, operators=True, rho3d=False, reuse_grid=False, given=None):
        """The trial of this kind of call if one exists (bench.py reports its verdict), else None."""
        return cls._all.get(cls.key(device, N, operators, rho3d, reuse_grid, given))

    def decided(self):
        return self.t[True] is not None and self.t[False] is not None

    def pipelined(self):
        with self._lock:
            self.n += 1
            if self.n <= 4:
                self.now = True
            elif self.n <= 7:
                self.now = False
            elif not self.decided():  # (a call that raised was not timed: stay with the pipelined build)
                self.now = True
            else:
                best = self.t[True] <= self.t[False]
                self.now = (not best) if (self.n % self.REMEASURE_EVERY == 0) else best
            return self.now

    def record(self, seconds):
        with self._lock:
            if self.n in (3, 4, 6, 7):
                self.t[self.now] = seconds if self.t[self.now] is None else min(self.t[self.now], seconds)
            elif self.n > 7 and self.decided():
                if self.n % self.REMEASURE_EVERY == 0:
                    self.t[self.now] = seconds
                    return
                if seconds > self.SLOW_FACTOR * self.t[self.now]:
                    self.slow += 1
                    if self.slow >= self.SLOW_STREAK:  # this host is not what it was: measure both again (no warm-ups needed: the blocks are pooled)
                        self.n, self.t, self.slow = 2, {True: None, False: None}, 0
                else:
                    self.slow = 0
                    self.t[self.now] = min(self.t[self.now], seconds) if seconds < self.t[self.now] else 0.9 * self.t[self.now] + 0.1 * seconds


# Which engine served the previous host-pointer transportmatrix of a device: the single-device context ("ctx") or an otmb_mgpu (its device
# tuple).  reuse_grid is the caller's promise about "the previous call" -- but each engine checks it against ITS OWN previous call.  When the
# engine changes (the Trial does that by itself), the new engine's residency keys may describe arrays the caller has edited since, legitimately
# passing reuse_grid = False in between: the promise is therefore not forwarded to an engine that did not serve the previous call.
_last_engine = {}
_engine_lock = threading.Lock()


def _reuse_grid_for(device, engine, reuse_grid):
    with _engine_lock:
        same = _last_engine.get(int(device)) == engine
        _last_engine[int(device)] = engine
    return bool(reuse_grid) and same


PER_COLUMN_MAX = (7, 7, 5, 3, 3)  # rows a column of T, Tadv, TκH, TκVML, TκVdeep can hold (matrixbuilding.jl:244-296, :348-415, :450-477)


def _transportmatrix_mgpu(phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, operators, devices, ignore_ops,
                          reuse_grid=False, reuse_fluxes=False, given=None):
    """otmb_mgpu_set_reuse -> otmb_mgpu_transportmatrix_plan -> otmb_mgpu_transportmatrix_fetch: the same build cut into depth slabs,
    one per listed GPU."""
    mg = mgpu(devices)
    ctx = context(list(devices)[0])  # (pinned result arrays only: the pool is the process's, every device's DMA reaches it)
    keep, passthrough = [], []
    a = _tm_args(phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, keep, passthrough, given)
    reuse_grid = _reuse_grid_for(list(devices)[0], tuple(int(d) for d in devices), reuse_grid)
    mg.check(capi.lib().otmb_mgpu_set_reuse(mg.handle, int(bool(reuse_grid) and all(passthrough)), int(bool(reuse_fluxes))))
    a.only_t = 0 if operators else 1
    a.ignore_ops = int(ignore_ops)
    nnz = (C.c_int64 * 5)()
    import time as _time

    t0 = _time.perf_counter()
    mg.check(capi.lib().otmb_mgpu_transportmatrix_plan(mg.handle, C.byref(a), C.byref(nnz)))
    last_call_seconds["plan"] = _time.perf_counter() - t0
    N = int(indices["N"])
    want = [_wanted(m, operators, given) for m in range(5)]
    colptr = [_out_array(ctx, N + 1, np.int64) if want[m] else None for m in range(5)]
    rowval = [_out_array(ctx, int(nnz[m]), np.int64) if want[m] else None for m in range(5)]
    nzval = [_out_array(ctx, int(nnz[m]), np.float64) if want[m] else None for m in range(5)]
    ptrs = lambda arrs: capi.ptr_array(5, [None if x is None else x.ctypes.data for x in arrs])
    cp, rv, nz = ptrs(colptr), ptrs(rowval), ptrs(nzval)
    final = (C.c_int64 * 5)()
    t0 = _time.perf_counter()
    mg.check(capi.lib().otmb_mgpu_transportmatrix_fetch(mg.handle, C.byref(cp), C.byref(rv), C.byref(nz), C.byref(final)))
    last_call_seconds["fetch"] = _time.perf_counter() - t0
    return _result(N, colptr, rowval, nzval, final, operators, given)


def _transportmatrix_onepass(phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, operators, devices, ignore_ops,
                             reuse_grid=False, reuse_fluxes=False, given=None):
    """otmb_mgpu_set_reuse -> result arrays at their upper bounds -> otmb_mgpu_transportmatrix_onepass: no nnz round trip, every slab's upload
    beside the download of the slab above it.  The matrices' rowval / nzval are the first nnz entries of those arrays."""
    import time as _time

    mg = mgpu(devices)
    ctx = context(list(devices)[0])
    keep, passthrough = [], []
    a = _tm_args(phi, mlotst, gridmetrics, indices, rho, kH, kVML, kVdeep, upwind, keep, passthrough, given)
    reuse_grid = _reuse_grid_for(list(devices)[0], tuple(int(d) for d in devices), reuse_grid)
    mg.check(capi.lib().otmb_mgpu_set_reuse(mg.handle, int(bool(reuse_grid) and all(passthrough)), int(bool(reuse_fluxes))))
    a.only_t = 0 if operators else 1
    a.ignore_ops = int(ignore_ops)
    N = int(indices["N"])
    want = [_wanted(m, operators, given) for m in range(5)]
    ckey, bound = _capacity_bounds(indices, gridmetrics)
    ptrs = lambda arrs: capi.ptr_array(5, [None if x is None else x.ctypes.data for x in arrs])
    for attempt in (0, 1):
        # result arrays at capacities that suffice: the wet mask's bounds, and for what varies between time slices (Tadv, TκVML) the previous
        # slice's counts with a margin -- a slice that outgrows them (OTMB_ERR_CAPACITY) is built again at the mask's bounds
        cap = [(c + 1 if want[m] else 0) for m, c in enumerate(_capacities(ckey, bound) if attempt == 0 else bound)]
        colptr = [_out_array(ctx, N + 1, np.int64) if want[m] else None for m in range(5)]
        rowval = [_out_array(ctx, cap[m], np.int64) if want[m] else None for m in range(5)]
        nzval = [_out_array(ctx, cap[m], np.float64) if want[m] else None for m in range(5)]
        cp, rv, nz = ptrs(colptr), ptrs(rowval), ptrs(nzval)
        caps, final = (C.c_int64 * 5)(*cap), (C.c_int64 * 5)()
        t0 = _time.perf_counter()
        rc = capi.lib().otmb_mgpu_transportmatrix_onepass(mg.handle, C.byref(a), C.byref(cp), C.byref(rv), C.byref(nz), C.byref(caps), C.byref(final))
        if rc == capi.CAPACITY and attempt == 0:
            _prev_nnz.pop(ckey, None)
            del colptr, rowval, nzval
            continue
        mg.check(rc)
        break
    last_call_seconds["plan"], last_call_seconds["fetch"] = _time.perf_counter() - t0, 0.0
    last_call_seconds["result_bytes_pinned"] = sum(16 * c for c in cap) + 8 * (N + 1) * sum(want)
    last_call_seconds["result_bytes_used"] = sum(16 * int(final[m]) for m in range(5) if want[m]) + 8 * (N + 1) * sum(want)
    if all(want):
        _prev_nnz[ckey] = [int(x) for x in final]
    return _result(N, colptr, rowval, nzval, final, operators, given)


_static_cap = {}  # (address of indices.wet3D, shape, topology) -> otmb_static_capacity's five bounds
_prev_nnz = {}    # the same key -> the five counts of the previous full build on that grid
GROWTH = (1.0, 1.25, 1.0, 1.5, 1.0)  # margins over the previous slice's counts (T, TκH, TκVdeep: the mask's bounds are exact enough already)


def _capacity_bounds(indices, gridmetrics):
    """otmb_static_capacity of this grid (host arithmetic, once per indices object): entries that always suffice, from the wet mask alone."""
    wet = np.asfortranarray(indices["wet3D"]).view(np.uint8)
    topo = _topology_kind(gridmetrics)
    key = (wet.ctypes.data, wet.shape, topo)
    if key not in _static_cap:
        if len(_static_cap) > 64:
            _static_cap.clear()
            _prev_nnz.clear()
        out = (C.c_int64 * 5)()
        rc = capi.lib().otmb_static_capacity(wet.ctypes.data, *wet.shape, topo, C.byref(out))
        if rc != capi.OK:
            raise capi.OtmbError(rc, "otmb_static_capacity")
        _static_cap[key] = ([int(x) for x in out], wet)  # (the array is kept: its address is the key)
    return key, _static_cap[key][0]


def _capacities(key, bound):
    prev = _prev_nnz.get(key)
    if prev is None:
        return list(bound)
    return [min(b, int(p * g) + 4096) for b, p, g in zip(bound, prev, GROWTH)]


def _build_operator(which, *, gridmetrics, indices, phi=None, rho=1035.0, mlotst=None, kappa=(500.0, 0.1, 1.0e-5), upwind=True,
                    device=0):
    """buildTadv / buildTκH / buildTκVML / buildTκVdeep (src/matrixbuilding.jl:31-120): ONE operator, by the fused build with every other
    matrix switched off (otmb_tm_args.skip_ops: neither counted, written nor copied home) and nothing the others alone would raise raised
    (ignore_ops).  What the other operators would read and this one does not gets harmless stand-ins, as the reference never looks at it."""
    m = MATS.index(which)
    shape = np.asarray(gridmetrics["v3D"]).shape
    if phi is None:
        z = np.zeros(shape, dtype=np.float64, order="F")
        phi = {k: z for k in PHI_ORDER}
    if mlotst is None:
        mlotst = np.full(shape[:2], np.nan)
    others = 0x1e & ~(1 << m)
    r = _transportmatrix_fused(phi, mlotst, gridmetrics, indices, rho, *kappa, upwind, True, False, False, device, others, None, None,
                               0x1f & ~(1 << m))
    return r[which]


def buildTadv(*, ϕ=None, phi=None, gridmetrics, indices, ρ=None, rho=None, upwind=True, device=0):
    """matrixbuilding.jl:31-44."""
    return _build_operator("Tadv", gridmetrics=gridmetrics, indices=indices, phi=ϕ if ϕ is not None else phi,
                           rho=ρ if ρ is not None else rho, upwind=upwind, device=device)


def buildTκH(*, gridmetrics, indices, ρ=None, rho=None, κH=None, kappaH=None, device=0):
    """matrixbuilding.jl:51-66 (ρ is accepted and unused, as in the reference)."""
    k = κH if κH is not None else kappaH
    return _build_operator("TκH", gridmetrics=gridmetrics, indices=indices, kappa=(float(k), 0.1, 1.0e-5), device=device)


def buildTκVML(*, mlotst, gridmetrics, indices, κVML=None, kappaVML=None, device=0):
    """matrixbuilding.jl:74-95."""
    k = κVML if κVML is not None else kappaVML
    return _build_operator("TκVML", gridmetrics=gridmetrics, indices=indices, mlotst=mlotst, kappa=(500.0, float(k), 1.0e-5), device=device)


def buildTκVdeep(*, mlotst=None, gridmetrics, indices, κVdeep=None, kappaVdeep=None, device=0):
    """matrixbuilding.jl:103-120 (mlotst is accepted and unused, as in the reference)."""
    k = κVdeep if κVdeep is not None else kappaVdeep
    return _build_operator("TκVdeep", gridmetrics=gridmetrics, indices=indices, kappa=(500.0, 0.1, float(k)), device=device)


buildTkH, buildTkVML, buildTkVdeep = buildTκH, buildTκVML, buildTκVdeep  # ASCII aliases


def lump_and_spray(wet3D, vol, T, mask=None, *, di=2, dj=2, dk=1, device=0):
    """LUMP, SPRAY, vol_c = lump_and_spray(wet3D, vol, T, mask; di, dj, dk) (src/extratools.jl:38-119): coarsening in
    di x dj x dk blocks inside `mask`, never across cells that T's pattern does not connect.  `LUMP * x` is the coarse
    vector, `LUMP * T * SPRAY` the coarse operator.  T: SparseMatrixCSC (only its pattern is read)."""
    ctx = context(device)
    wet = np.asfortranarray(np.asarray(wet3D) != 0).astype(np.uint8)
    nx, ny, nz = wet.shape
    m = None if mask is None else np.asfortranarray(np.asarray(mask) != 0).astype(np.uint8)
    if m is not None and m.shape != wet.shape:
        raise ValueError("mask must have the shape of wet3D")
    v = np.ascontiguousarray(vol, dtype=np.float64)
    N = v.size
    if T.n != N or T.m != N:
        raise ValueError("T must be N x N with N = length(vol)")
    Tp = np.ascontiguousarray(T.colptr, dtype=np.int64)
    Ti = np.ascontiguousarray(T.rowval, dtype=np.int64)
    lrow, lval = np.empty(max(N, 1), np.int64), np.empty(max(N, 1), np.float64)
    scp, srow, vc = np.empty(N + 2, np.int64), np.empty(max(N, 1), np.int64), np.empty(max(N, 1), np.float64)
    nc = C.c_int64(0)
    ctx.check(capi.lib().otmb_lump_and_spray(ctx.handle, wet.ctypes.data, None if m is None else m.ctypes.data, nx, ny, nz,
                                            v.ctypes.data, N, Tp.ctypes.data, Ti.ctypes.data, int(di), int(dj), int(dk),
                                            lrow.ctypes.data, lval.ctypes.data, scp.ctypes.data, srow.ctypes.data,
                                            vc.ctypes.data, C.byref(nc)))
    Nc = int(nc.value)
    LUMP = SparseMatrixCSC(Nc, N, np.arange(1, N + 2, dtype=np.int64), lrow[:N].copy(), lval[:N].copy())
    SPRAY = SparseMatrixCSC(N, Nc, scp[: Nc + 1].copy(), srow[:N].copy(), np.ones(N))
    return LUMP, SPRAY, vc[:Nc].copy()


def as2D(x, wet3D):
    """src/extratools.jl:111-115: scatter a surface vector back onto the (nx,ny) grid, NaN on land."""
    wet = np.asfortranarray(wet3D).astype(bool)
    out = np.full(wet.shape[:2], np.nan, order="F")
    out.T[wet[:, :, 0].T] = np.asarray(x, dtype=np.float64)  # column-major order of the wet cells
    return out


def as3D(x, wet3D):
    """src/extratools.jl:122-126: scatter a wet-cell vector back onto the (nx,ny,nz) grid, NaN on land."""
    wet = np.asfortranarray(wet3D).astype(bool)
    out = np.full(wet.shape, np.nan, order="F")
    out.ravel(order="K")[np.flatnonzero(wet.ravel(order="K"))] = np.asarray(x, dtype=np.float64)
    return out
