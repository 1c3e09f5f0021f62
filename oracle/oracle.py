"""ctypes front end of the C oracle (oracle/otmb_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED -- see the header of otmb_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libotmb_oracle.so")

ERRORS = {
    -1: "ρ contains NaNs", -2: "Tadv contains NaNs.", -3: "TκH contains NaNs.", -4: "TκVML contains NaNs.",
    -5: "TκVdeep contains NaNs.", -6: "flux into land or outside the grid", -7: "Unknown grid type",
    -8: "AssertionError: all fluxes missing", -9: "allocation failed",
    -20: "ArgumentError: Adjacency / distance matrices must be symmetric",
    -21: "ArgumentError: Collection must contain exactly 1 element (vertexpermutation)", -22: "Unknown Arakawa grid type",
}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(ERRORS.get(code, f"oracle error {code}"))
        self.code = code


def build(force=False):
    src = os.path.join(_HERE, "otmb_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB


class _Grid(C.Structure):
    _fields_ = [("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64), ("topo", C.c_int32)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_bp = C.POINTER(C.c_uint8)


class _TmArgs(C.Structure):
    _fields_ = [
        ("phi", _dp * 6), ("v3D", _dp), ("thk", _dp), ("rho", _dp), ("rho_scalar", C.c_double),
        ("Lwet", _ip), ("Lwet3D", _ip), ("N", C.c_int64), ("g", _Grid),
        ("edge", _dp * 4), ("dist", _dp * 4), ("area2D", _dp), ("zt", _dp), ("mlotst", _dp),
        ("kappaH", C.c_double), ("kappaVML", C.c_double), ("kappaVdeep", C.c_double), ("upwind", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.orc_makeindices.restype = C.c_int64
        _lib.orc_advection_entries.restype = C.c_int64
        _lib.orc_hdiff_entries.restype = C.c_int64
        _lib.orc_vdiff_entries.restype = C.c_int64
        _lib.orc_sparse.restype = C.c_int64
        _lib.orc_spadd.restype = C.c_int64
        _lib.orc_haversine.restype = C.c_double
        _lib.orc_haversine.argtypes = [C.c_double] * 4
    return _lib


def _f(a):
    return np.asfortranarray(a, dtype=np.float64)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def _b(a):
    return a.ctypes.data_as(_bp) if a is not None else None


def _grid(shape, topo):
    return _Grid(int(shape[0]), int(shape[1]), int(shape[2]), int(topo))


PHI_ORDER = ("east", "west", "north", "south", "top", "bottom")
HDIRS = ("west", "east", "south", "north")
MATS = ("T", "Tadv", "TκH", "TκVML", "TκVdeep")


def makeindices(v3D):
    v3D = _f(v3D)
    G = v3D.size
    Lwet = np.empty(G, dtype=np.int64)
    Lwet3D = np.empty(v3D.shape, dtype=np.int64, order="F")
    wet3D = np.empty(v3D.shape, dtype=np.uint8, order="F")
    N = lib().orc_makeindices(_d(v3D), C.c_int64(G), _i(Lwet), _i(Lwet3D), _b(wet3D))
    return dict(N=int(N), Lwet=Lwet[:N].copy(), Lwet3D=Lwet3D, wet3D=wet3D)


def facefluxes(umo, vmo, wet3D, fill, topo, top_below=None, return_flags=False):
    """umo/vmo: Float64 copies are made here (velocities.jl:125-126).  top_below / return_flags:
    depth-slab test support (see orc_facefluxes_slab)."""
    u = np.array(umo, dtype=np.float64, order="F")
    v = np.array(vmo, dtype=np.float64, order="F")
    wet3D = np.asfortranarray(wet3D, dtype=np.uint8)
    g = _grid(u.shape, topo)
    out = {k: np.empty(u.shape, dtype=np.float64, order="F") for k in PHI_ORDER}
    tb = None if top_below is None else np.ascontiguousarray(np.asarray(top_below, dtype=np.float64).ravel(order="F"))
    flags = (C.c_int32 * 2)() if return_flags else None
    rc = lib().orc_facefluxes_slab(_d(u), _d(v), _b(wet3D), C.c_double(float(fill)), C.byref(g),
                                   _d(out["east"]), _d(out["west"]), _d(out["north"]), _d(out["south"]),
                                   _d(out["top"]), _d(out["bottom"]), _d(tb), flags)
    if rc:
        raise OracleError(rc)
    if return_flags:
        return out, (bool(flags[0]), bool(flags[1]))
    return out


def _rho_args(rho, shape):
    if np.ndim(rho) == 0:
        return None, float(rho)
    r = _f(rho)
    assert r.shape == tuple(shape)
    return r, 0.0


def advection_entries(phi, v3D, rho, Lwet, Lwet3D, topo, upwind=True):
    v3D = _f(v3D)
    ph = [_f(phi[k]) for k in PHI_ORDER]
    N = len(Lwet)
    rho_a, rho_s = _rho_args(rho, v3D.shape)
    cap = max(1, 12 * N)
    I = np.empty(cap, np.int64); J = np.empty(cap, np.int64); V = np.empty(cap, np.float64)
    g = _grid(v3D.shape, topo)
    Lwet = np.ascontiguousarray(Lwet, dtype=np.int64)
    Lwet3D = np.asfortranarray(Lwet3D, dtype=np.int64)
    n = lib().orc_advection_entries((_dp * 6)(*[_d(p) for p in ph]), _d(v3D), _d(rho_a), C.c_double(rho_s),
                                    _i(Lwet), _i(Lwet3D), C.c_int64(N), C.byref(g), C.c_int32(int(upwind)),
                                    _i(I), _i(J), _d(V))
    if n < 0:
        raise OracleError(n)
    return I[:n].copy(), J[:n].copy(), V[:n].copy()


def hdiff_entries(v3D, thk, edge, dist, Lwet, Lwet3D, topo, kappaH, OmegaH=None):
    v3D = _f(v3D); thk = _f(thk)
    e = [_f(edge[d]) for d in HDIRS]; dd = [_f(dist[d]) for d in HDIRS]
    N = len(Lwet); cap = max(1, 8 * N)
    I = np.empty(cap, np.int64); J = np.empty(cap, np.int64); V = np.empty(cap, np.float64)
    g = _grid(v3D.shape, topo)
    Lwet = np.ascontiguousarray(Lwet, dtype=np.int64); Lwet3D = np.asfortranarray(Lwet3D, dtype=np.int64)
    Om = None if OmegaH is None else np.ascontiguousarray(OmegaH, dtype=np.uint8)
    n = lib().orc_hdiff_entries(_d(v3D), _d(thk), (_dp * 4)(*[_d(x) for x in e]), (_dp * 4)(*[_d(x) for x in dd]),
                                _i(Lwet), _i(Lwet3D), C.c_int64(N), C.byref(g), C.c_double(kappaH), _b(Om),
                                _i(I), _i(J), _d(V))
    if n < 0:
        raise OracleError(n)
    return I[:n].copy(), J[:n].copy(), V[:n].copy()


def ml_mask(zt, mlotst, Lwet, shape, topo=0):
    N = len(Lwet)
    Om = np.empty(max(N, 1), np.uint8)
    zt = np.ascontiguousarray(zt, dtype=np.float64); ml = _f(mlotst)
    Lwet = np.ascontiguousarray(Lwet, dtype=np.int64)
    g = _grid(shape, topo)
    lib().orc_ml_mask(_d(zt), _d(ml), _i(Lwet), C.c_int64(N), C.byref(g), _b(Om))
    return Om[:N]


def vdiff_entries(v3D, area2D, zt, Lwet, Lwet3D, topo, kappaV, Omega=None):
    v3D = _f(v3D); area2D = _f(area2D); zt = np.ascontiguousarray(zt, dtype=np.float64)
    N = len(Lwet); cap = max(1, 4 * N)
    I = np.empty(cap, np.int64); J = np.empty(cap, np.int64); V = np.empty(cap, np.float64)
    g = _grid(v3D.shape, topo)
    Lwet = np.ascontiguousarray(Lwet, dtype=np.int64); Lwet3D = np.asfortranarray(Lwet3D, dtype=np.int64)
    Om = None if Omega is None else np.ascontiguousarray(Omega, dtype=np.uint8)
    n = lib().orc_vdiff_entries(_d(v3D), _d(area2D), _d(zt), _i(Lwet), _i(Lwet3D), C.c_int64(N), C.byref(g),
                                C.c_double(kappaV), _b(Om), _i(I), _i(J), _d(V))
    if n < 0:
        raise OracleError(n)
    return I[:n].copy(), J[:n].copy(), V[:n].copy()


def sparse(I, J, V, m, n):
    I = np.ascontiguousarray(I, np.int64); J = np.ascontiguousarray(J, np.int64); V = np.ascontiguousarray(V, np.float64)
    ln = len(I)
    colptr = np.empty(n + 1, np.int64); rowval = np.empty(max(ln, 1), np.int64); nzval = np.empty(max(ln, 1), np.float64)
    nnz = lib().orc_sparse(_i(I), _i(J), _d(V), C.c_int64(ln), C.c_int64(m), C.c_int64(n), _i(colptr), _i(rowval), _d(nzval))
    if nnz < 0:
        raise OracleError(nnz)
    return colptr, rowval[:nnz].copy(), nzval[:nnz].copy()


def spadd(A, B, n):
    Ap, Ai, Ax = A; Bp, Bi, Bx = B
    cap = max(1, len(Ai) + len(Bi))
    Cp = np.empty(n + 1, np.int64); Ci = np.empty(cap, np.int64); Cx = np.empty(cap, np.float64)
    Ai = np.ascontiguousarray(Ai, np.int64); Bi = np.ascontiguousarray(Bi, np.int64)
    Ax = np.ascontiguousarray(Ax, np.float64); Bx = np.ascontiguousarray(Bx, np.float64)
    nnz = lib().orc_spadd(C.c_int64(n), _i(np.ascontiguousarray(Ap, np.int64)), _i(Ai), _d(Ax),
                          _i(np.ascontiguousarray(Bp, np.int64)), _i(Bi), _d(Bx), _i(Cp), _i(Ci), _d(Cx))
    return Cp, Ci[:nnz].copy(), Cx[:nnz].copy()


def omp_threads():
    """Host threads the multi-threaded variant uses for the column-parallel adds (OMP_NUM_THREADS / all cores)."""
    return int(lib().orc_omp_threads())


def transportmatrix(phi, gm, idx, rho, mlotst, kappaH=500.0, kappaVML=0.1, kappaVdeep=1.0e-5, upwind=True,
                    tight=False, parallel=False):
    """Whole reference path (matrixbuilding.jl:128-150).  gm: gridmetrics NT/dict; idx: makeindices dict.
    Returns {name: (colptr, rowval, nzval)} for T, Tadv, TκH, TκVML, TκVdeep (1-based Int64).
    parallel=True: orc_transportmatrix_omp (four concurrent operator builds, column-parallel adds; same bits)."""
    v3D = _f(gm["v3D"]); thk = _f(gm["thkcello"])
    ph = [_f(phi[k]) for k in PHI_ORDER]
    e = [_f(gm["edge_length_2D"][d]) for d in HDIRS]; dd = [_f(gm["distance_to_neighbour_2D"][d]) for d in HDIRS]
    area2D = _f(gm["area2D"]); zt = np.ascontiguousarray(gm["zt"], dtype=np.float64); ml = _f(mlotst)
    Lwet = np.ascontiguousarray(idx["Lwet"], dtype=np.int64); Lwet3D = np.asfortranarray(idx["Lwet3D"], dtype=np.int64)
    N = len(Lwet)
    rho_a, rho_s = _rho_args(rho, v3D.shape)
    a = _TmArgs()
    a.phi = (_dp * 6)(*[_d(p) for p in ph]); a.v3D = _d(v3D); a.thk = _d(thk); a.rho = _d(rho_a); a.rho_scalar = rho_s
    a.Lwet = _i(Lwet); a.Lwet3D = _i(Lwet3D); a.N = N
    topo = gm["gridtopology"]["kind"] if isinstance(gm["gridtopology"], dict) else int(gm["gridtopology"])
    a.g = _grid(v3D.shape, topo)
    a.edge = (_dp * 4)(*[_d(x) for x in e]); a.dist = (_dp * 4)(*[_d(x) for x in dd])
    a.area2D = _d(area2D); a.zt = _d(zt); a.mlotst = _d(ml)
    a.kappaH = kappaH; a.kappaVML = kappaVML; a.kappaVdeep = kappaVdeep; a.upwind = int(upwind)
    n1 = max(N, 1)
    caps = [8 * n1] * 5 if tight else [28 * n1, 12 * n1, 8 * n1, 4 * n1, 4 * n1]
    cp = [np.empty(N + 1, np.int64) for _ in range(5)]
    rv = [np.empty(c, np.int64) for c in caps]
    nz = [np.empty(c, np.float64) for c in caps]
    nnz = (C.c_int64 * 5)()
    fn = lib().orc_transportmatrix_omp if parallel else lib().orc_transportmatrix
    rc = fn(C.byref(a), (_ip * 5)(*[_i(x) for x in cp]), (_ip * 5)(*[_i(x) for x in rv]),
            (_dp * 5)(*[_d(x) for x in nz]), nnz)
    if rc:
        raise OracleError(rc)
    return {m: (cp[k], rv[k][: nnz[k]].copy(), nz[k][: nnz[k]].copy()) for k, m in enumerate(MATS)}


def lump_and_spray(wet3D, vol, T, mask=None, di=2, dj=2, dk=1):
    """extratools.jl:38-119.  T = (colptr, rowval, nzval) 1-based.  Returns LUMP, SPRAY as (colptr, rowval, nzval) and vol_c."""
    wet = np.asfortranarray(np.asarray(wet3D) != 0).astype(np.uint8)
    nx, ny, nz = wet.shape
    m = None if mask is None else np.asfortranarray(np.asarray(mask) != 0).astype(np.uint8)
    vol = np.ascontiguousarray(vol, dtype=np.float64)
    N = len(vol)
    Tp = np.ascontiguousarray(T[0], dtype=np.int64); Ti = np.ascontiguousarray(T[1], dtype=np.int64)
    lrow = np.empty(max(N, 1), np.int64); lval = np.empty(max(N, 1), np.float64)
    scp = np.empty(N + 2, np.int64); srow = np.empty(max(N, 1), np.int64); vc = np.empty(max(N, 1), np.float64)
    Nc = C.c_int64(0)
    fn = lib().orc_lump_and_spray
    fn.restype = C.c_int32
    rc = fn(wet.ctypes.data_as(C.c_void_p), None if m is None else m.ctypes.data_as(C.c_void_p), C.c_int64(nx), C.c_int64(ny),
            C.c_int64(nz), _d(vol), _i(Tp), _i(Ti), C.c_int64(di), C.c_int64(dj), C.c_int64(dk), _i(lrow), _d(lval), _i(scp),
            _i(srow), _d(vc), C.byref(Nc))
    if rc:
        raise OracleError(rc)
    n = Nc.value
    LUMP = (np.arange(1, N + 2, dtype=np.int64), lrow[:N].copy(), lval[:N].copy())
    SPRAY = (scp[: n + 1].copy(), srow[:N].copy(), np.ones(N))
    return LUMP, SPRAY, vc[:n].copy()


def haversine(lon1, lat1, lon2, lat2):
    return lib().orc_haversine(float(lon1), float(lat1), float(lon2), float(lat2))


def to_scipy(csc, N):
    """(colptr,rowval,nzval) 1-based -> scipy.sparse.csc_matrix (0-based); keeps stored zeros."""
    import scipy.sparse as sp

    colptr, rowval, nzval = csc
    return sp.csc_matrix((nzval, rowval - 1, colptr - 1), shape=(N, N))


def velocity_flux(a_i, a_j, rho, thk, edge_east, edge_north, topo, to_velocity=False):
    """velocity2fluxes (to_velocity=False, velocities.jl:10-39) / fluxes2velocity (True, :50-74), C-grid."""
    a_i = _f(a_i); a_j = _f(a_j); thk = _f(thk)
    rho_a, rho_s = _rho_args(rho, thk.shape)
    ee = _f(edge_east); en = _f(edge_north)
    oi = np.empty(thk.shape, order="F"); oj = np.empty(thk.shape, order="F")
    g = _grid(thk.shape, topo)
    rc = lib().orc_velocity_flux(_d(a_i), _d(a_j), _d(rho_a), C.c_double(rho_s), _d(thk), _d(ee), _d(en), C.byref(g),
                                 C.c_int32(int(to_velocity)), _d(oi), _d(oj))
    if rc:
        raise OracleError(rc)
    return oi, oj


def bolus_gm_velocity(rho, Z3D, wet3D, dist_east, dist_north, topo, kappaGM=600.0, maxslope=0.01):
    """bolus_GM_velocity(ρ, gridmetrics, indices; κGM, maxslope) -> (u, v)  (RediGM.jl:46-79; unpinned)."""
    rho = _f(rho); Z3D = _f(Z3D)
    wet3D = np.asfortranarray(wet3D, dtype=np.uint8)
    de = _f(dist_east); dn = _f(dist_north)
    u = np.empty(rho.shape, order="F"); v = np.empty(rho.shape, order="F")
    g = _grid(rho.shape, topo)
    rc = lib().orc_bolus_gm_velocity(_d(rho), _d(Z3D), _b(wet3D), _d(de), _d(dn), C.byref(g), C.c_double(kappaGM),
                                     C.c_double(maxslope), _d(u), _d(v))
    if rc:
        raise OracleError(rc)
    return u, v


# ---- makegridmetrics / topology detection / Arakawa detection / B-grid interpolation (src/gridcellgeometry.jl) ----------
ARAKAWA_POS = ("C", "SW", "SE", "NE", "NW", "S", "N", "W", "E")  # field order of the reference's `cell` NamedTuple (:65)


def _data_and_fill(x):
    """(array, _FillValue or None) of a plain array or of an object with .data/.properties like the package's Cube."""
    props = getattr(x, "properties", None) or {}
    data = getattr(x, "data", x)
    return np.asarray(data), (float(props["_FillValue"]) if "_FillValue" in props else None)


def makegridmetrics(*, areacello, volcello, lon, lat, lev, lon_vertices, lat_vertices):
    """makegridmetrics (gridcellgeometry.jl:265-311) -> dict with the reference's field names; per-direction dicts keyed by
    direction name; gridtopology = dict(kind=0|1).  Raises OracleError(-7) for an unknown topology."""
    area, fa = _data_and_fill(areacello)
    vol, fv = _data_and_fill(volcello)
    area = np.asfortranarray(np.where(np.isnan(area.astype(np.float64)), np.nan, area), dtype=np.float64)
    vol = np.asfortranarray(vol, dtype=np.float64)
    nx, ny, nz = vol.shape
    lon = _f(np.asarray(getattr(lon, "data", lon)))
    lat = _f(np.asarray(getattr(lat, "data", lat)))
    lonv_in = _f(np.asarray(getattr(lon_vertices, "data", lon_vertices)))
    latv_in = _f(np.asarray(getattr(lat_vertices, "data", lat_vertices)))
    f2 = lambda: np.empty((nx, ny), order="F")
    f3 = lambda: np.empty((nx, ny, nz), order="F")
    area2D, v3D, thk, Z3D = f2(), f3(), f3(), f3()
    lonv, latv = np.empty((4, nx, ny), order="F"), np.empty((4, nx, ny), order="F")
    edge, dedge, dnbr = [f2() for _ in range(4)], [f2() for _ in range(4)], [f2() for _ in range(4)]
    pa = lambda arrs: (_dp * 4)(*[_d(a) for a in arrs])
    fn = lib().orc_makegridmetrics
    fn.restype = C.c_int32
    rc = fn(_d(vol), _d(area), C.c_double(fa if fa is not None else 0.0), C.c_int32(fa is not None),
            C.c_double(fv if fv is not None else 0.0), C.c_int32(fv is not None), _d(lon), _d(lat), _d(lonv_in), _d(latv_in),
            C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), _d(area2D), _d(v3D), _d(thk), _d(Z3D), _d(lonv), _d(latv), pa(edge),
            pa(dedge), pa(dnbr))
    if rc < 0:
        raise OracleError(rc)
    return dict(area2D=area2D, v3D=v3D, thkcello=thk, lon_vertices=lonv, lat_vertices=latv, lon=lon, lat=lat, Z3D=Z3D,
                zt=np.asarray(getattr(lev, "data", lev), dtype=np.float64),
                edge_length_2D=dict(zip(HDIRS, edge)), distance_to_edge_2D=dict(zip(HDIRS, dedge)),
                distance_to_neighbour_2D=dict(zip(HDIRS, dnbr)), gridtopology=dict(kind=int(rc)))


def getgridtopology(lon_vertices, lat_vertices):
    """gridtopology.jl:33-53 on vertices in the default order: 0 bipolar, 1 tripolar, 2 unknown."""
    lonv, latv = _f(lon_vertices), _f(lat_vertices)
    fn = lib().orc_getgridtopology
    fn.restype = C.c_int32
    return int(fn(_d(lonv), _d(latv), C.c_int64(lonv.shape[1]), C.c_int64(lonv.shape[2])))


def vertexpermutation(lon_vertices, lat_vertices):
    lonv, latv = _f(lon_vertices), _f(lat_vertices)
    perm = (C.c_int32 * 4)()
    fn = lib().orc_vertexpermutation
    fn.restype = C.c_int32
    if fn(_d(lonv), _d(latv), C.c_int64(lonv.shape[1]), C.c_int64(lonv.shape[2]), perm):
        raise OracleError(-21)
    return [int(x) for x in perm]


def getarakawagrid(u_lon, u_lat, v_lon, v_lat, gridmetrics):
    """gridcellgeometry.jl:50-95 -> ("A"|"B"|"C", u_pos, v_pos, relerr)."""
    lon, lat = _f(gridmetrics["lon"]), _f(gridmetrics["lat"])
    lonv, latv = _f(gridmetrics["lon_vertices"]), _f(gridmetrics["lat_vertices"])
    kind, up, vp, rel = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_double(0)
    fn = lib().orc_getarakawagrid
    fn.restype = C.c_int32
    rc = fn(C.c_double(float(np.asarray(u_lon)[0, 0])), C.c_double(float(np.asarray(u_lat)[0, 0])),
            C.c_double(float(np.asarray(v_lon)[0, 0])), C.c_double(float(np.asarray(v_lat)[0, 0])), _d(lon), _d(lat), _d(lonv), _d(latv),
            C.byref(kind), C.byref(up), C.byref(vp), C.byref(rel))
    if rc:
        raise OracleError(-22)  # "Unknown Arakawa grid type"
    return "ABC"[kind.value], ARAKAWA_POS[up.value], ARAKAWA_POS[vp.value], rel.value


def bgrid_to_cgrid(u, v, fill, gridmetrics):
    """interpolateontodefaultCgrid(…, ::BGridCell) (:106-140) -> u2, u2_lon, u2_lat, v2, v2_lon, v2_lat."""
    u = np.asfortranarray(np.asarray(u), dtype=np.float64)
    v = np.asfortranarray(np.asarray(v), dtype=np.float64)
    nx, ny, nz = u.shape
    lonv, latv = _f(gridmetrics["lon_vertices"]), _f(gridmetrics["lat_vertices"])
    u2, v2 = np.empty(u.shape, order="F"), np.empty(u.shape, order="F")
    pts = [np.empty((nx, ny), order="F") for _ in range(4)]
    fn = lib().orc_bgrid_to_cgrid
    fn.restype = None
    fn(_d(u), _d(v), C.c_double(float(fill)), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), _d(lonv), _d(latv), _d(u2), _d(v2),
       _d(pts[0]), _d(pts[1]), _d(pts[2]), _d(pts[3]))
    return u2, pts[0], pts[1], v2, pts[2], pts[3]
