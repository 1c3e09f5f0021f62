"""Synthetic CMIP-like ocean grids.

The reference's tests read the author's private NetCDF files or download CMIP6 Zarr
(test/LocalBuiltMatrix.jl:18, test/online.jl:19-38); neither exists offline, so every test
and benchmark here runs on a seeded synthetic grid with the same array names, shapes,
units and missing-value conventions as the CMIP variables the reference consumes:
`areacello, volcello, lon, lat, lev, lon_vertices, lat_vertices, umo, vmo, mlotst`.

Shapes follow Julia: (nx,ny[,nz]) Fortran-ordered, vertices (4,nx,ny).
"""
import numpy as np

from ._nt import NT, Cube

FILL = 1.0e20
R = 6371000.0

PRESETS = {
    # name: (nx, ny, nz, land_fraction)
    "tiny": (12, 10, 6, 0.35),
    "small": (36, 30, 10, 0.30),
    "access1deg": (360, 300, 50, 0.30),  # BASELINE.json configs[0], configs[1]
    "quarterdeg": (1440, 1080, 75, 0.30),  # configs[2], configs[3]
    "tenthdeg": (3600, 2700, 75, 0.30),  # configs[4]
}


def levels(nz, dz0=10.0, zmax=5800.0):
    """nz stretched levels: thickness grows geometrically from dz0 so that the column is ~zmax deep.
    Returns (zt, dz): nominal level-centre depths `lev` and thicknesses."""
    if nz == 1:
        return np.array([zmax / 2]), np.array([zmax])
    lo, hi = 1.0, 2.0
    for _ in range(200):  # bisection on the growth ratio
        r = 0.5 * (lo + hi)
        tot = dz0 * (r**nz - 1) / (r - 1) if abs(r - 1) > 1e-14 else dz0 * nz
        if tot > zmax:
            hi = r
        else:
            lo = r
    dz = dz0 * r ** np.arange(nz)
    zt = np.cumsum(dz) - dz / 2
    return zt, dz


def _smooth_field(rng, nx, ny, scale):
    """Periodic-in-i smooth random field in [0,1] by bilinear upsampling of coarse white noise."""
    cx, cy = max(2, nx // scale), max(2, ny // scale)
    coarse = rng.random((cx, cy))
    xi = (np.arange(nx) + 0.5) / nx * cx
    yj = (np.arange(ny) + 0.5) / ny * (cy - 1)
    i0 = np.floor(xi).astype(int) % cx
    i1 = (i0 + 1) % cx
    fx = xi - np.floor(xi)
    j0 = np.minimum(np.floor(yj).astype(int), cy - 2)
    fy = yj - j0
    a = coarse[i0][:, j0] * (1 - fx)[:, None] + coarse[i1][:, j0] * fx[:, None]
    b = coarse[i0][:, j0 + 1] * (1 - fx)[:, None] + coarse[i1][:, j0 + 1] * fx[:, None]
    return a * (1 - fy)[None, :] + b * fy[None, :]


def vertices(nx, ny, topology="tripolar", lat_south=-78.0, lat_top=64.0, lat_pole=67.0, lon0=80.0):
    """lon/lat vertices (4,nx,ny) in the default order SW,SE,NE,NW (gridcellgeometry.jl:150-155).
    Rows are regular in lon/lat; for "tripolar" the north edge of the top row lies on a seam
    running from a pole at (lon0,lat_pole) over the North Pole to (lon0+180,lat_pole) so that
    vertex NE of cell i equals vertex NW of cell nx+1-i (getgridtopology's test,
    gridtopology.jl:44); for "bipolar" the top row's north vertices sit at lat == 90 (:41)."""
    dx = 360.0 / nx
    west = lon0 + dx * np.arange(nx)
    east = lon0 + dx * (np.arange(nx) + 1)
    if topology == "bipolar":
        late = np.linspace(lat_south, 90.0, ny + 1)
    else:
        late = np.concatenate([np.linspace(lat_south, lat_top, ny), [np.nan]])
    lonv = np.empty((4, nx, ny), order="F")
    latv = np.empty((4, nx, ny), order="F")
    lonv[0] = west[:, None]
    lonv[3] = west[:, None]
    lonv[1] = east[:, None]
    lonv[2] = east[:, None]
    latv[0] = late[None, :-1]
    latv[1] = late[None, :-1]
    latv[2] = late[None, 1:]
    latv[3] = late[None, 1:]
    if topology != "bipolar":
        m = np.arange(nx + 1)
        mm = np.minimum(m, nx - m)  # seam vertex m coincides with vertex nx-m
        s = mm / (nx / 2.0)  # 0 at pole A ... 1 at pole B
        seam_lon = np.where(s < 0.5, lon0, lon0 + 180.0)
        seam_lat = np.where(s < 0.5, lat_pole + (90.0 - lat_pole) * 2 * s, 90.0 - (90.0 - lat_pole) * 2 * (s - 0.5))
        lonv[3, :, -1] = seam_lon[:-1]
        latv[3, :, -1] = seam_lat[:-1]
        lonv[2, :, -1] = seam_lon[1:]
        latv[2, :, -1] = seam_lat[1:]
    return lonv, latv


def make_grid(nx, ny, nz, *, seed=20260501, land_fraction=0.30, topology="tripolar", rho="scalar",
              vertex_order=(0, 1, 2, 3), dtype_flux=np.float64):
    """Build a synthetic grid + forcing.  Returns NT with the CMIP-like raw inputs
    (`areacello, volcello, lon, lat, lev, lon_vertices, lat_vertices, umo, vmo, mlotst`, the 3-D/2-D
    ones as `Cube`s carrying `_FillValue`) and the parameter set of SURVEY.md section 8d."""
    rng = np.random.default_rng(seed)
    zt, dz = levels(nz)
    zbot = np.cumsum(dz)
    lonv, latv = vertices(nx, ny, topology)
    # cell centres: vertex mean (the seam row included); only finite positive distances matter
    lon = np.asfortranarray(lonv.mean(axis=0))
    lat = np.asfortranarray(latv.mean(axis=0))
    # cell area ~ R^2 dlon (sin lat_n - sin lat_s), from the SW/NW..NE corner latitudes
    lat_s = np.deg2rad(latv[0])
    lat_n = np.deg2rad(np.maximum(latv[3], latv[2]))
    area = R * R * np.deg2rad(360.0 / nx) * np.abs(np.sin(lat_n) - np.sin(lat_s))
    area = np.asfortranarray(np.maximum(area, 1.0e6))

    # bathymetry: land where a smooth field is below its land_fraction quantile
    scale = max(3, min(nx, ny) // 12)
    f = _smooth_field(rng, nx, ny, scale)
    land = f < np.quantile(f, land_fraction)
    # keep the fold row and its approach partly open so the tripolar duplicates are exercised
    mid = nx // 2
    land[max(0, mid - 3) : mid + 3, -3:] = False
    land[:2, -2:] = False
    land[-2:, -2:] = False
    depth = zbot[-1] * (0.05 + 0.95 * _smooth_field(rng, nx, ny, max(2, scale // 2)))
    # a few single-level columns (wet only at k=1) and very shallow shelves
    shallow = rng.random((nx, ny)) < 0.02
    depth = np.where(shallow, 0.6 * dz[0], depth)
    depth = np.where(land, 0.0, depth)

    ztop = zbot - dz
    thk = np.clip(depth[:, :, None] - ztop[None, None, :], 0.0, dz[None, None, :])  # partial bottom cells
    thk = np.where(thk < 0.2 * dz[None, None, :], 0.0, thk)  # drop slivers at the bottom
    vol = np.asfortranarray(thk * area[:, :, None])  # 0 on land -> NaN in makegridmetrics
    wet = vol > 0

    thkmax = dz.max()
    sig = 1.0e9 * thk / thkmax
    umo = np.asfortranarray(rng.standard_normal((nx, ny, nz)) * sig)
    vmo = np.asfortranarray(rng.standard_normal((nx, ny, nz)) * sig)
    umo[~wet] = FILL
    vmo[~wet] = FILL
    if dtype_flux == np.float32:
        umo = np.asfortranarray(umo.astype(np.float32))
        vmo = np.asfortranarray(vmo.astype(np.float32))
    mlotst = np.asfortranarray(np.exp(rng.uniform(np.log(10.0), np.log(1000.0), (nx, ny))))
    mlotst[~wet[:, :, 0]] = np.nan  # `missing` on land columns

    if rho == "scalar":
        rho_val = 1035.0
    else:
        zc = np.cumsum(thk, axis=2) - 0.5 * thk
        rho_val = 1025.0 + 0.004 * zc + 0.1 * rng.standard_normal((nx, ny, nz))
        rho_val = np.asfortranarray(np.where(wet, rho_val, np.nan))

    p = list(vertex_order)
    fill32 = float(np.float32(FILL)) if dtype_flux == np.float32 else FILL
    return NT(
        nx=nx, ny=ny, nz=nz, seed=seed, topology=topology,
        areacello=Cube(area, _FillValue=FILL), volcello=Cube(vol, _FillValue=FILL),
        lon=lon, lat=lat, lev=zt,
        lon_vertices=np.asfortranarray(lonv[p]), lat_vertices=np.asfortranarray(latv[p]),
        umo=Cube(umo, _FillValue=fill32), vmo=Cube(vmo, _FillValue=fill32),
        mlotst=mlotst, rho=rho_val,
        kappaH=500.0, kappaVML=0.1, kappaVdeep=1.0e-5,
    )


def preset(name, **kw):
    nx, ny, nz, lf = PRESETS[name]
    kw.setdefault("land_fraction", lf)
    return make_grid(nx, ny, nz, **kw)


# ---- level-sliced generation (multi-GPU depth slabs) ------------------------------------------------
def _grid2d(nx, ny, nz, seed, land_fraction, topology):
    """The 2-D part of a synthetic grid (identical on every rank for a given seed)."""
    rng = np.random.default_rng([seed, 7])
    zt, dz = levels(nz)
    lonv, latv = vertices(nx, ny, topology)
    lon = np.asfortranarray(lonv.mean(axis=0))
    lat = np.asfortranarray(latv.mean(axis=0))
    lat_s = np.deg2rad(latv[0])
    lat_n = np.deg2rad(np.maximum(latv[3], latv[2]))
    area = np.asfortranarray(np.maximum(R * R * np.deg2rad(360.0 / nx) * np.abs(np.sin(lat_n) - np.sin(lat_s)), 1.0e6))
    import os

    scale = int(os.environ.get("OTMB_SYN_SCALE", "0")) or max(3, min(nx, ny) // 12)  # (experiments: coarser / finer coastlines)
    f = _smooth_field(rng, nx, ny, scale)
    land = f < np.quantile(f, land_fraction)
    mid = nx // 2
    land[max(0, mid - 3): mid + 3, -3:] = False
    land[:2, -2:] = False
    land[-2:, -2:] = False
    zbot = np.cumsum(dz)
    depth = zbot[-1] * (0.05 + 0.95 * _smooth_field(rng, nx, ny, max(2, scale // 2)))
    shallow = rng.random((nx, ny)) < 0.02
    depth = np.where(shallow, 0.6 * dz[0], depth)
    depth = np.where(land, 0.0, depth)
    mlotst = np.asfortranarray(np.exp(rng.uniform(np.log(10.0), np.log(1000.0), (nx, ny))))
    mlotst[depth < 0.2 * dz[0]] = np.nan
    return NT(zt=zt, dz=dz, lonv=lonv, latv=latv, lon=lon, lat=lat, area=area, depth=depth, mlotst=mlotst)


def _level_thickness(g2, k0, k1):
    dz = g2.dz[k0:k1]
    ztop = (np.cumsum(g2.dz) - g2.dz)[k0:k1]
    thk = np.clip(g2.depth[:, :, None] - ztop[None, None, :], 0.0, dz[None, None, :])
    return np.where(thk < 0.2 * dz[None, None, :], 0.0, thk)


def level_wet_counts(nx, ny, nz, *, seed=20260501, land_fraction=0.30, topology="tripolar"):
    """Wet cells per level of the grid make_slab() slices (cheap: from the 2-D bathymetry only)."""
    g2 = _grid2d(nx, ny, nz, seed, land_fraction, topology)
    return np.array([(_level_thickness(g2, k, k + 1) > 0).sum() for k in range(nz)], dtype=np.int64)


def make_slab(nx, ny, nz, k0, k1, *, seed=20260501, land_fraction=0.30, topology="tripolar", rho="scalar"):
    """Levels [k0,k1) of a synthetic (nx,ny,nz) grid.  Every random field of level k comes from its own
    stream (seed, k), so any rank can generate any level and the slabs of all ranks tile one consistent
    global grid: make_slab(..., 0, nz) IS that grid.  Same fields and conventions as make_grid()."""
    g2 = _grid2d(nx, ny, nz, seed, land_fraction, topology)
    thk = _level_thickness(g2, k0, k1)
    vol = np.asfortranarray(thk * g2.area[:, :, None])
    wet = vol > 0
    nl = k1 - k0
    umo = np.empty((nx, ny, nl), order="F")
    vmo = np.empty((nx, ny, nl), order="F")
    noise = np.empty((nx, ny, nl), order="F")
    thkmax = g2.dz.max()
    for q, k in enumerate(range(k0, k1)):
        r = np.random.default_rng([seed, 1000 + k])
        sig = 1.0e9 * thk[:, :, q] / thkmax
        umo[:, :, q] = r.standard_normal((nx, ny)) * sig
        vmo[:, :, q] = r.standard_normal((nx, ny)) * sig
        noise[:, :, q] = r.standard_normal((nx, ny))
    umo[~wet] = FILL
    vmo[~wet] = FILL
    if rho == "scalar":
        rho_val = 1035.0
    else:
        ztop = (np.cumsum(g2.dz) - g2.dz)[k0:k1]
        zc = ztop[None, None, :] + 0.5 * thk
        rho_val = np.asfortranarray(np.where(wet, 1025.0 + 0.004 * zc + 0.1 * noise, np.nan))
    return NT(
        nx=nx, ny=ny, nz=nz, k0=k0, k1=k1, seed=seed, topology=topology,
        areacello=Cube(g2.area, _FillValue=FILL), volcello=Cube(vol, _FillValue=FILL),
        lon=g2.lon, lat=g2.lat, lev=g2.zt, lon_vertices=g2.lonv, lat_vertices=g2.latv,
        umo=Cube(umo, _FillValue=FILL), vmo=Cube(vmo, _FillValue=FILL), mlotst=g2.mlotst, rho=rho_val,
        kappaH=500.0, kappaVML=0.1, kappaVdeep=1.0e-5,
    )
