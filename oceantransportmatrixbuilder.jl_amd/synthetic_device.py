"""Synthetic grids whose 3-D fields are generated ON the GPU (SURVEY.md section 8d workloads that are too large
to build on the host and copy: the 0.1 degree grid has 729 M cells, 5.8 GB per Float64 array).

2-D geometry comes from the host generator (synthetic.py) and the host makegridmetrics; thickness, volume, density
and the mass transports are the same formulas as synthetic.make_grid evaluated with torch on the device (torch's own
random stream, so the VALUES differ from make_grid's: what is checked on these grids are size-independent
properties and oracle comparisons of sub-slabs copied back to the host, never a host-generated twin)."""
import numpy as np
import torch

from . import synthetic
from ._nt import NT, Cube
from .capi import HDIRS


def make_device_grid(name_or_shape, device, *, seed=20260501, land_fraction=None, rho="array", k0=0, k1=None):
    """Returns NT(gm=host gridmetrics of the 2-D part, tensors...) with flat float64 device tensors in Julia's
    column-major (nx,ny,nz) order: v3d, thkcello, umo, vmo, rho (or a float), mlotst, area2d, zt, edge_length[4],
    dist_nbr[4].  k0/k1 select a depth slab [k0,k1) of the grid (multi-GPU strong scaling): every level's random
    fields come from a generator seeded with (seed, level), so the slabs of all ranks tile one global grid."""
    from .gridmetrics import makegridmetrics

    if isinstance(name_or_shape, str):
        nx, ny, nz, lf = synthetic.PRESETS[name_or_shape]
    else:
        nx, ny, nz = name_or_shape
        lf = 0.30
    if land_fraction is not None:
        lf = land_fraction
    k1 = nz if k1 is None else k1
    FILL = synthetic.FILL
    g2 = synthetic._grid2d(nx, ny, nz, seed, lf, "tripolar")
    zt, dz = g2.zt, g2.dz
    zbot = np.cumsum(dz)
    # geometry: makegridmetrics' 2-D outputs do not depend on volcello, so a one-level volume is enough here
    vol1 = np.asfortranarray((np.clip(g2.depth, 0.0, dz[0]) * g2.area)[:, :, None])
    gm = makegridmetrics(areacello=Cube(g2.area, _FillValue=FILL), volcello=Cube(vol1, _FillValue=FILL), lon=g2.lon,
                         lat=g2.lat, lev=zt[:1], lon_vertices=g2.lonv, lat_vertices=g2.latv)

    def dev2(a):  # (nx, ny) host -> (ny, nx) device view of the same column-major data
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)).to(device)

    d_depth, d_area = dev2(g2.depth), dev2(g2.area)
    d_dz = torch.from_numpy(dz[k0:k1].copy()).to(device)[:, None, None]
    d_ztop = torch.from_numpy((zbot - dz)[k0:k1].copy()).to(device)[:, None, None]
    # torch shape (nz, ny, nx) is Julia's (nx, ny, nz) column-major
    thk = torch.minimum((d_depth[None] - d_ztop).clamp_min_(0.0), d_dz)
    thk = torch.where(thk < 0.2 * d_dz, torch.zeros((), dtype=torch.float64, device=device), thk)
    wet = thk > 0
    nanv = torch.full((), float("nan"), dtype=torch.float64, device=device)
    fillv = torch.full((), FILL, dtype=torch.float64, device=device)
    v3d = torch.where(wet, thk * d_area[None], nanv)  # volcello == 0 -> NaN (gridcellgeometry.jl:270-276)
    thkc = torch.where(wet, thk, nanv)                # thkcello = v3D / area2D
    sig = 1.0e9 / float(dz.max())
    nl = k1 - k0
    umo = torch.empty((nl, ny, nx), dtype=torch.float64, device=device)
    vmo = torch.empty_like(umo)
    noise = torch.empty_like(umo) if rho == "array" else None
    gen = torch.Generator(device=device)
    for q in range(nl):
        gen.manual_seed(int(seed) * 1000003 + k0 + q)
        umo[q] = torch.randn((ny, nx), generator=gen, dtype=torch.float64, device=device)
        vmo[q] = torch.randn((ny, nx), generator=gen, dtype=torch.float64, device=device)
        if noise is not None:
            noise[q] = torch.randn((ny, nx), generator=gen, dtype=torch.float64, device=device)
    umo = torch.where(wet, umo * thk * sig, fillv)
    vmo = torch.where(wet, vmo * thk * sig, fillv)
    if rho == "array":
        rho_t = torch.where(wet, 1025.0 + 0.004 * (d_ztop + 0.5 * thk) + 0.1 * noise, nanv).reshape(-1)
    else:
        rho_t = 1035.0
    ml = dev2(g2.mlotst)
    del thk, wet, noise
    return NT(nx=nx, ny=ny, nz=nz, k0=k0, k1=k1, gm=gm, topology=int(gm.gridtopology.kind), fill=FILL, zt_host=zt, mlotst_host=g2.mlotst,
              v3d=v3d.reshape(-1), thkcello=thkc.reshape(-1), umo=umo.reshape(-1), vmo=vmo.reshape(-1), rho=rho_t,
              mlotst=ml.reshape(-1), area2d=d_area.reshape(-1), zt=torch.from_numpy(zt[k0:k1].copy()).to(device),
              edge_length=[dev2(gm.edge_length_2D[d]).reshape(-1) for d in HDIRS],
              dist_nbr=[dev2(gm.distance_to_neighbour_2D[d]).reshape(-1) for d in HDIRS],
              kappaH=500.0, kappaVML=0.1, kappaVdeep=1.0e-5)


def assembler_for(dg, device_index=0, upwind=True):
    """DeviceAssembler on a whole device-generated grid (k0 == 0, k1 == nz)."""
    from .device import DeviceAssembler

    asm = DeviceAssembler(device_index)
    asm.set_grid_tensors(shape=(dg.nx, dg.ny, dg.k1 - dg.k0), topology=dg.topology, v3d=dg.v3d, thkcello=dg.thkcello,
                         edge_length=dg.edge_length, dist_nbr=dg.dist_nbr, area2d=dg.area2d, zt=dg.zt, mlotst=dg.mlotst,
                         rho=dg.rho, kappaH=dg.kappaH, kappaVML=dg.kappaVML, kappaVdeep=dg.kappaVdeep, upwind=upwind)
    return asm


def host_copy(dg):
    """The device-generated grid as HOST arrays in the shapes the host API takes (bench.py's end_to_end leg on the 0.25 degree grid, where
    building the 3-D fields on the host would take minutes): (g, gm) with g.umo / g.vmo Cubes, g.rho, g.mlotst, the κ and gm the gridmetrics
    NamedTuple whose 3-D members are copied down once.  Whole grids only (k0 == 0, k1 == nz)."""
    assert dg.k0 == 0 and dg.k1 == dg.nz
    shape = (dg.nx, dg.ny, dg.nz)

    def h3(t):
        return np.asfortranarray(t.cpu().numpy().reshape(shape, order="F"))

    gm = NT(**{k: v for k, v in dg.gm.items()})
    gm["v3D"], gm["thkcello"], gm["zt"] = h3(dg.v3d), h3(dg.thkcello), np.ascontiguousarray(dg.zt_host, dtype=np.float64)
    g = NT(umo=Cube(h3(dg.umo), _FillValue=dg.fill), vmo=Cube(h3(dg.vmo), _FillValue=dg.fill),
           rho=h3(dg.rho) if torch.is_tensor(dg.rho) else float(dg.rho), mlotst=np.asfortranarray(dg.mlotst_host, dtype=np.float64),
           kappaH=dg.kappaH, kappaVML=dg.kappaVML, kappaVdeep=dg.kappaVdeep)
    return g, gm
