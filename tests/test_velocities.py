"""velocity2fluxes / fluxes2velocity / facefluxesfromvelocities (src/velocities.jl:10-108,132-151):
oracle vs the Python transliteration on CPU, HIP vs oracle on GPU (bit-exact), and the reference's own
round-trip test u -> ϕ -> u (test/local_full.jl:300-304) on wet cells."""
import numpy as np
import pytest

from helpers import make_case
from oracle import pyref


def _velocities(g, gm, seed=5, dtype=np.float64):
    """C-grid velocity fields: u on east-face midpoints, v on north-face midpoints, fill on land."""
    rng = np.random.default_rng(seed)
    wet = ~np.isnan(gm.v3D)
    u = np.where(wet, rng.standard_normal(gm.v3D.shape) * 0.1, 1e20)
    v = np.where(wet, rng.standard_normal(gm.v3D.shape) * 0.1, 1e20)
    lv, tv = gm.lon_vertices, gm.lat_vertices
    u_lon, u_lat = (lv[1] + lv[2]) / 2, (tv[1] + tv[2]) / 2  # E = mid(SE, NE)
    v_lon, v_lat = (lv[2] + lv[3]) / 2, (tv[2] + tv[3]) / 2  # N = mid(NE, NW)
    return np.asfortranarray(u.astype(dtype)), u_lon, u_lat, np.asfortranarray(v.astype(dtype)), v_lon, v_lat


@pytest.mark.parametrize("name", ["tiny_tripolar", "tiny_rho3d", "odd_nx_fold"])
def test_oracle_velocity_flux_matches_transliteration(oracle, name):
    g, gm = make_case(name)
    u, _, _, v, _, _ = _velocities(g, gm)
    kind = gm.gridtopology.kind
    topo = pyref.Topo(kind, g.nx, g.ny, g.nz)
    fi, fj = oracle.velocity_flux(u, v, g.rho, gm.thkcello, gm.edge_length_2D["east"], gm.edge_length_2D["north"], kind)
    pi, pj = pyref.velocity_flux(u, v, g.rho, gm, topo)
    assert np.array_equal(fi, pi, equal_nan=True) and np.array_equal(fj, pj, equal_nan=True)
    ui, uj = oracle.velocity_flux(fi, fj, g.rho, gm.thkcello, gm.edge_length_2D["east"], gm.edge_length_2D["north"], kind, True)
    qi, qj = pyref.velocity_flux(pi, pj, g.rho, gm, topo, True)
    assert np.array_equal(ui, qi, equal_nan=True) and np.array_equal(uj, qj, equal_nan=True)


def test_oracle_velocity_flux_bipolar_raises(oracle):
    g, gm = make_case("tiny_bipolar")
    u, _, _, v, _, _ = _velocities(g, gm)
    with pytest.raises(oracle.OracleError):
        oracle.velocity_flux(u, v, 1035.0, gm.thkcello, gm.edge_length_2D["east"], gm.edge_length_2D["north"], 0)


def _bgrid_fields(gm, dtype, seed=8):
    rng = np.random.default_rng(seed)
    wet = ~np.isnan(gm.v3D)
    fill = float(dtype(1e20))
    u = np.asfortranarray(np.where(wet, rng.standard_normal(wet.shape) * 0.1, fill).astype(dtype))
    v = np.asfortranarray(np.where(wet, rng.standard_normal(wet.shape) * 0.1, fill).astype(dtype))
    return u, v, fill


@pytest.mark.parametrize("name", ["tiny_tripolar", "tiny_bipolar", "odd_nx_fold"])
def test_getarakawagrid_host_detection_matches_oracle_and_transliteration(oracle, name):
    """getarakawagrid (gridcellgeometry.jl:50-95): the product's host detection, the oracle's C restatement and the
    literal Python transliteration agree on A-, B- (all four corners) and C-grid placements, and on what is none of them."""
    from otmb_amd.gridmetrics import getarakawagrid

    g, gm = make_case(name)
    u, u_lon, u_lat, v, v_lon, v_lat = _velocities(g, gm)
    lv, tv = gm.lon_vertices, gm.lat_vertices
    placements = {"C": (u_lon, u_lat, v_lon, v_lat), "A": (gm.lon, gm.lat, gm.lon, gm.lat),
                  "C_WS": ((lv[0] + lv[3]) / 2, (tv[0] + tv[3]) / 2, (lv[0] + lv[1]) / 2, (tv[0] + tv[1]) / 2)}
    for q, corner in enumerate(("SW", "SE", "NE", "NW")):
        placements["B_" + corner] = (lv[q], tv[q], lv[q], tv[q])
    for label, (ul, ut, vl, vt) in placements.items():
        want = oracle.getarakawagrid(ul, ut, vl, vt, gm)
        assert want[:3] == pyref.getarakawagrid(ul, ut, vl, vt, gm.lon, gm.lat, lv, tv)[:3], label
        assert getarakawagrid(ul, ut, vl, vt, gm) == want[:3], label
        assert want[0] == label[0] and want[3] < 1e-9
    assert oracle.getarakawagrid(u_lon, u_lat, v_lon, v_lat, gm)[:3] == ("C", "E", "N")
    # u on a corner, v in the centre: none of the three
    with pytest.raises(oracle.OracleError):
        oracle.getarakawagrid(lv[2], tv[2], gm.lon, gm.lat, gm)
    with pytest.raises(RuntimeError, match="Unknown Arakawa grid type"):
        getarakawagrid(lv[2], tv[2], gm.lon, gm.lat, gm)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_bgrid_interpolation_matches_transliteration(oracle, dtype):
    g, gm = make_case("tiny_tripolar")
    u, v, fill = _bgrid_fields(gm, dtype)
    got = oracle.bgrid_to_cgrid(u, v, fill, gm)
    want = pyref.bgrid_to_cgrid(u.astype(np.float64), v.astype(np.float64), fill, gm.lon_vertices, gm.lat_vertices)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # the interpolated points are the east / north face midpoints: a C-grid as far as getarakawagrid is concerned
    assert oracle.getarakawagrid(got[1], got[2], got[4], got[5], gm)[:3] == ("C", "E", "N")


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype", [("tiny_tripolar", np.float64), ("tiny_rho3d", np.float64), ("small_rho3d", np.float32),
                                        ("odd_nx_fold", np.float64)])
def test_hip_velocity_flux_matches_oracle_and_round_trips(oracle, name, dtype):
    import otmb_amd.api as api

    g, gm = make_case(name)
    u, u_lon, u_lat, v, v_lon, v_lat = _velocities(g, gm, dtype=dtype)
    kind = gm.gridtopology.kind
    fi, fj = api.velocity2fluxes(u, u_lon, u_lat, v, v_lon, v_lat, gm, g.rho)
    ri, rj = oracle.velocity_flux(u, v, g.rho, gm.thkcello, gm.edge_length_2D["east"], gm.edge_length_2D["north"], kind)
    assert np.array_equal(fi, ri, equal_nan=True) and np.array_equal(fj, rj, equal_nan=True)
    ui, uj = api.fluxes2velocity(fi, fj, gm, g.rho)
    qi, qj = oracle.velocity_flux(ri, rj, g.rho, gm.thkcello, gm.edge_length_2D["east"], gm.edge_length_2D["north"], kind, True)
    assert np.array_equal(ui, qi, equal_nan=True) and np.array_equal(uj, qj, equal_nan=True)
    # test/local_full.jl:300-304: the round trip returns the velocities on faces between two wet cells
    wet = ~np.isnan(gm.v3D)
    both_e = wet & np.roll(wet, -1, axis=0)
    np.testing.assert_allclose(ui[both_e], u.astype(np.float64)[both_e], rtol=1e-12)
    # facefluxesfromvelocities == facefluxes(velocity2fluxes(...))
    from otmb_amd import Cube

    idx = api.makeindices(gm.v3D)
    phi = api.facefluxesfromvelocities(uo=Cube(u, _FillValue=1e20), uo_lon=u_lon, uo_lat=u_lat, vo=Cube(v, _FillValue=1e20),
                                       vo_lon=v_lon, vo_lat=v_lat, gridmetrics=gm, indices=idx, ρ=g.rho)
    ref = oracle.facefluxes(ri, rj, idx.wet3D.view(np.uint8), 1e20, kind)
    for k in ref:
        assert np.array_equal(phi[k], ref[k]), k


@pytest.mark.gpu
def test_hip_velocity2fluxes_bipolar_is_an_error():
    import otmb_amd.api as api
    from otmb_amd.capi import OtmbError

    g, gm = make_case("tiny_bipolar")
    u, u_lon, u_lat, v, v_lon, v_lat = _velocities(g, gm)
    with pytest.raises(OtmbError):
        api.velocity2fluxes(u, u_lon, u_lat, v, v_lon, v_lat, gm, 1035.0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hip_bgrid_interpolation(oracle, dtype):
    """interpolateontodefaultCgrid(…, ::BGridCell) (gridcellgeometry.jl:106-140): the HIP path against the oracle's
    restatement -- fields and the six returned arrays, bit for bit."""
    import otmb_amd.api as api
    from otmb_amd import Cube

    g, gm = make_case("small_rho3d")
    u, v, fill = _bgrid_fields(gm, dtype)
    ne_lon, ne_lat = gm.lon_vertices[2], gm.lat_vertices[2]  # B-grid: both velocity points on the NE corner
    assert oracle.getarakawagrid(ne_lon, ne_lat, ne_lon, ne_lat, gm)[:3] == ("B", "NE", "NE")
    u2, u2_lon, u2_lat, v2, v2_lon, v2_lat = api.interpolateontodefaultCgrid(Cube(u, _FillValue=fill), ne_lon, ne_lat,
                                                                              Cube(v, _FillValue=fill), ne_lon, ne_lat, gm)
    want = oracle.bgrid_to_cgrid(u, v, fill, gm)
    for name, a, b in zip(("u2", "u2_lon", "u2_lat", "v2", "v2_lon", "v2_lat"), (u2, u2_lon, u2_lat, v2, v2_lon, v2_lat), want):
        assert np.array_equal(np.asarray(a), b), name
    assert oracle.getarakawagrid(u2_lon, u2_lat, v2_lon, v2_lat, gm)[:3] == ("C", "E", "N")
    # and the whole chain runs: B-grid velocities -> fluxes, same as feeding the interpolated fields directly
    fi, fj = api.velocity2fluxes(Cube(u, _FillValue=fill), ne_lon, ne_lat, Cube(v, _FillValue=fill), ne_lon, ne_lat, gm, g.rho)
    ri, rj = oracle.velocity_flux(u2, v2, g.rho, gm.thkcello, gm.edge_length_2D["east"], gm.edge_length_2D["north"], gm.gridtopology.kind)
    assert np.array_equal(fi, ri, equal_nan=True) and np.array_equal(fj, rj, equal_nan=True)
