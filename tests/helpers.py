"""Shared test helpers: build the reference pipeline inputs for a synthetic grid."""
import os

import numpy as np

import otmb_amd
from otmb_amd import synthetic

MATS = ("T", "Tadv", "TκH", "TκVML", "TκVdeep")


def gridmetrics_of(g):
    return otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                    lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)


def randomize_metrics(gm, seed=1):
    """Replace the geometric 2-D metrics by positive random fields.  The kernels take the metric
    arrays as data, so this exercises the stencil logic on grids whose geometry is degenerate
    (odd nx tripolar: the fold-centre cell is its own north neighbour and the real distance is 0)."""
    rng = np.random.default_rng(seed)
    nx, ny = gm.area2D.shape
    for d in gm.edge_length_2D:
        gm.edge_length_2D[d] = np.asfortranarray(rng.uniform(5e4, 2e5, (nx, ny)))
        keep_nan = np.isnan(gm.distance_to_neighbour_2D[d])
        dn = rng.uniform(5e4, 2e5, (nx, ny))
        dn[keep_nan] = np.nan
        gm.distance_to_neighbour_2D[d] = np.asfortranarray(dn)
    return gm


CASES = {
    # name: (make_grid kwargs, randomize metrics?)
    "tiny_tripolar": (dict(nx=12, ny=10, nz=6), False),
    "tiny_rho3d": (dict(nx=12, ny=10, nz=6, rho="array", seed=3), False),
    "tiny_bipolar": (dict(nx=12, ny=10, nz=6, topology="bipolar", seed=5), False),
    "odd_nx_fold": (dict(nx=7, ny=5, nz=4, seed=7, land_fraction=0.2), True),
    "nx2": (dict(nx=2, ny=3, nz=3, seed=8, land_fraction=0.0), True),
    "even_fold_open": (dict(nx=8, ny=4, nz=3, seed=12, land_fraction=0.05), True),
    "small_rho3d": (dict(nx=36, ny=30, nz=10, seed=9, rho="array"), False),
    "float32_flux": (dict(nx=12, ny=10, nz=6, seed=13, dtype_flux=np.float32), False),
}


def make_case(name):
    kw, rnd = CASES[name]
    kw = dict(kw)
    g = synthetic.make_grid(kw.pop("nx"), kw.pop("ny"), kw.pop("nz"), **kw)
    gm = gridmetrics_of(g)
    if rnd:
        randomize_metrics(gm)
    return g, gm


def assert_csc_equal(a, b, what="", rtol=0.0):
    """a, b: (colptr,rowval,nzval).  Pattern must be bit-exact; values bit-exact when rtol == 0."""
    assert np.array_equal(a[0], b[0]), f"{what}: colptr differs"
    assert np.array_equal(a[1], b[1]), f"{what}: rowval differs"
    if rtol == 0.0:
        same = (a[2] == b[2]) & (np.signbit(a[2]) == np.signbit(b[2]))
        assert same.all(), f"{what}: nzval differs at {np.flatnonzero(~same)[:5]}: {a[2][~same][:5]} vs {b[2][~same][:5]}"
    else:
        np.testing.assert_allclose(a[2], b[2], rtol=rtol, atol=0.0, err_msg=what)


# The suite is also run under the library's experiment switches (OTMB_COUNT_IN_FF=0, ...): tests ABOUT a switched-off feature skip, tests
# that only note which kernels ran adapt
COUNTS_ON = os.environ.get("OTMB_COUNT_IN_FF", "1") != "0"
