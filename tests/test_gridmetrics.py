"""makegridmetrics (src/gridcellgeometry.jl:265-311): the oracle's C restatement vs the scalar Python transliteration
and vs the product's host numpy mirror (CPU), vertex permutation / topology detection, and the device version vs
the ORACLE (GPU).  Exact arithmetic (replace, divisions, cumsum, vertex sorting) is compared bit for bit; the
haversine results (sin/cos/asin from three different math libraries) at 1e-12 relative, the north star's tolerance."""
import numpy as np
import pytest

import otmb_amd
from helpers import gridmetrics_of
from oracle import pyref
from otmb_amd import gridtopology as gt
from otmb_amd import synthetic


@pytest.mark.parametrize("topology", ["tripolar", "bipolar"])
def test_host_metrics_match_scalar_transliteration(topology):
    g = synthetic.make_grid(12, 10, 4, topology=topology, seed=3)
    gm = gridmetrics_of(g)
    el, de, dn = pyref.gridmetrics_2d(gm.lon, gm.lat, gm.lon_vertices, gm.lat_vertices, gm.gridtopology.kind)
    for d in ("south", "east", "north", "west"):
        np.testing.assert_allclose(gm.edge_length_2D[d], el[d], rtol=1e-13)
        np.testing.assert_allclose(gm.distance_to_edge_2D[d], de[d], rtol=1e-13)
        np.testing.assert_allclose(gm.distance_to_neighbour_2D[d], dn[d], rtol=1e-13, equal_nan=True)
    assert gm.gridtopology.kind == (gt.TRIPOLAR if topology == "tripolar" else gt.BIPOLAR)
    wet = ~np.isnan(gm.v3D)
    assert np.array_equal(wet, g.volcello.data > 0)  # zero volume -> NaN -> land (:269-280)
    np.testing.assert_array_equal(gm.thkcello[wet], (g.volcello.data / g.areacello.data[:, :, None])[wet])


def _compare_with_oracle(got, ref, exact_transcendental=False):
    for k in ("area2D", "v3D", "thkcello", "Z3D", "lon_vertices", "lat_vertices"):
        assert np.array_equal(got[k], ref[k], equal_nan=True), k
    for grp in ("edge_length_2D", "distance_to_edge_2D", "distance_to_neighbour_2D"):
        for d in ("west", "east", "south", "north"):
            np.testing.assert_allclose(got[grp][d], ref[grp][d], rtol=1e-12, atol=0.0, equal_nan=True, err_msg=f"{grp}[{d}]")


@pytest.mark.parametrize("kw", [dict(), dict(topology="bipolar"), dict(vertex_order=(2, 3, 0, 1)), dict(vertex_order=(3, 2, 1, 0))])
def test_oracle_makegridmetrics_matches_transliteration_and_host_mirror(oracle, kw):
    g = synthetic.make_grid(12, 10, 4, seed=3, **kw)
    ref = oracle.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                 lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    gm = gridmetrics_of(g)
    assert ref["gridtopology"]["kind"] == gm.gridtopology.kind == oracle.getgridtopology(gm.lon_vertices, gm.lat_vertices)
    assert oracle.vertexpermutation(g.lon_vertices, g.lat_vertices) == list(np.argsort(kw.get("vertex_order", (0, 1, 2, 3))))
    _compare_with_oracle(gm, ref)
    el, de, dn = pyref.gridmetrics_2d(ref["lon"], ref["lat"], ref["lon_vertices"], ref["lat_vertices"], ref["gridtopology"]["kind"])
    for d in ("south", "east", "north", "west"):  # same libm underneath: equal to the last bit
        assert np.array_equal(ref["edge_length_2D"][d], el[d])
        assert np.array_equal(ref["distance_to_edge_2D"][d], de[d])
        assert np.array_equal(ref["distance_to_neighbour_2D"][d], dn[d], equal_nan=True)
    # the permutation sorts the vertices back whatever the input order
    base = oracle.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                  lon_vertices=synthetic.make_grid(12, 10, 4, seed=3, topology=kw.get("topology", "tripolar")).lon_vertices,
                                  lat_vertices=synthetic.make_grid(12, 10, 4, seed=3, topology=kw.get("topology", "tripolar")).lat_vertices)
    assert np.array_equal(base["lon_vertices"], ref["lon_vertices"]) and np.array_equal(base["lat_vertices"], ref["lat_vertices"])


def test_oracle_replace_rules_zero_fill_missing(oracle):
    """:269-280: 0, the two _FillValues and missing become NaN; replace() matches with isequal, so -0.0 does not."""
    from otmb_amd import Cube

    g = synthetic.make_grid(12, 10, 3, seed=4)
    vol = g.volcello.data.copy(order="F")
    wet = np.argwhere(vol > 0)
    (a, b, c), (d, e, f), (p, q, r) = wet[3], wet[40], wet[77]
    vol[a, b, c] = 1e20   # areacello's _FillValue in volcello is replaced too (one common set)
    vol[d, e, f] = -0.0
    vol[p, q, r] = np.nan
    kw = dict(lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    ref = oracle.makegridmetrics(areacello=Cube(g.areacello.data, _FillValue=1e20), volcello=Cube(vol, _FillValue=-9.0), **kw)
    got = otmb_amd.makegridmetrics(areacello=Cube(g.areacello.data, _FillValue=1e20), volcello=Cube(vol, _FillValue=-9.0), **kw)
    assert np.isnan(ref["v3D"][a, b, c]) and np.isnan(ref["v3D"][p, q, r])
    assert ref["v3D"][d, e, f] == 0 and np.signbit(ref["v3D"][d, e, f])
    _compare_with_oracle(got, ref)
    assert np.array_equal(np.signbit(got["v3D"]), np.signbit(ref["v3D"]))


def test_oracle_topology_detection(oracle):
    g = synthetic.make_grid(12, 10, 3, seed=4)
    assert oracle.getgridtopology(g.lon_vertices, g.lat_vertices) == 1
    lonv = g.lon_vertices.copy(order="F")
    lonv[2, 3, -1] += 7.0  # break the seam symmetry
    assert oracle.getgridtopology(lonv, g.lat_vertices) == 2 == gt.getgridtopology(lonv, g.lat_vertices)
    with pytest.raises(oracle.OracleError):
        oracle.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=lonv,
                               lat_vertices=g.lat_vertices)
    # a longitude off by a whole turn is the same longitude (isapprox_lon, gridtopology.jl:23-26)
    lonv = g.lon_vertices.copy(order="F")
    lonv[3, :, -1] += 360.0
    assert oracle.getgridtopology(lonv, g.lat_vertices) == 1 == gt.getgridtopology(lonv, g.lat_vertices)
    gb = synthetic.make_grid(12, 10, 3, seed=4, topology="bipolar")
    assert oracle.getgridtopology(gb.lon_vertices, gb.lat_vertices) == 0


def test_vertex_permutation_is_undone():
    """test/local_fast.jl:125-132 restated: whatever the vertex order on input, adjacent cells share vertices
    in the default orientation afterwards, and the metrics do not depend on the input order."""
    base = gridmetrics_of(synthetic.make_grid(12, 10, 3, seed=4))
    for order in [(1, 2, 3, 0), (3, 2, 1, 0), (2, 0, 3, 1)]:
        g = synthetic.make_grid(12, 10, 3, seed=4, vertex_order=order)
        gm = gridmetrics_of(g)
        assert np.array_equal(gm.lon_vertices, base.lon_vertices) and np.array_equal(gm.lat_vertices, base.lat_vertices)
        lv, tv = gm.lon_vertices, gm.lat_vertices
        assert np.array_equal(lv[1, :-1, :], lv[0, 1:, :]) and np.array_equal(tv[2, :-1, :-1], tv[3, 1:, :-1])  # SE(i) == SW(i+1), NE(i) == NW(i+1)
        assert np.array_equal(tv[3, :, :-2], tv[0, :, 1:-1])  # NW(j) == SW(j+1) below the seam row
        for d in base.edge_length_2D:
            assert np.array_equal(gm.edge_length_2D[d], base.edge_length_2D[d])


def test_unknown_topology_detected():
    g = synthetic.make_grid(12, 10, 3, seed=4)
    lonv = g.lon_vertices.copy(order="F")
    lonv[2, 3, -1] += 7.0  # break the seam symmetry
    assert gt.getgridtopology(lonv, g.lat_vertices) == gt.UNKNOWN
    with pytest.raises(RuntimeError, match="Unknown grid type"):
        otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                 lon_vertices=lonv, lat_vertices=g.lat_vertices)


def _haversine_answers():
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "known_answer", "haversine.json")))["pairs"]


def test_haversine_converts_the_latitudes_before_subtracting(oracle):
    """Distances.jl 0.10: φ₁ = deg2rad(x[2]); φ₂ = deg2rad(y[2]); Δφ = φ₂ - φ₁ (rounds 1-4 had deg2rad(lat₂ - lat₁)).  Known answers made by
    tests/golden/known_answer/make_haversine_answer.py without oracle or product.  The C oracle and the Python transliteration share the
    generator's libm: bit for bit, including the pairs where the two forms differ in the last place only; numpy's sin / cos may differ
    from libm's by an ulp, so the host product is held to 1e-12 -- on the pairs whose forms differ by 1e-7."""
    from otmb_amd import gridmetrics as hostgm

    pairs = _haversine_answers()
    assert sum(p["relative_difference_of_the_two_forms"] > 1e-9 for p in pairs) >= 2
    for p in pairs:
        want, other = p["convert_then_subtract"]["distance"], p["subtract_then_convert"]["distance"]
        args = (p["lon1"], p["lat1"], p["lon2"], p["lat2"])
        assert oracle.haversine(*args) == want != other
        assert pyref.haversine(args[:2], args[2:]) == want
        got = float(hostgm.haversine(*(np.float64(a) for a in args)))
        assert abs(got / want - 1) <= 1e-12
        if p["relative_difference_of_the_two_forms"] > 1e-9:
            assert abs(got / other - 1) > 1e-9


@pytest.mark.gpu
def test_device_haversine_converts_the_latitudes_before_subtracting():
    """The device kernel's copy of the formula, through otmb_makegridmetrics_dev: a 3 x 3 grid whose cell centres in column i = 1 sit at the
    latitudes of a known-answer pair with NEARLY EQUAL latitudes -- distance_to_neighbour_2D[north] of the lower cell is that pair's
    distance, and the two forms of Δφ differ by 1e-7 there."""
    from otmb_amd.device import DeviceAssembler
    from otmb_amd._nt import Cube

    for p in [q for q in _haversine_answers() if q["relative_difference_of_the_two_forms"] > 1e-9]:
        nx, ny, nz = 3, 3, 1
        lon = np.asfortranarray(np.tile(np.array([p["lon1"] - 1.0, p["lon1"], p["lon1"] + 1.0])[:, None], (1, ny)))
        lat = np.asfortranarray(np.tile(np.array([p["lat1"] - 1.0, p["lat1"], p["lat2"]])[None, :], (nx, 1)))
        # a consistent vertex lattice (neighbouring cells share their corners: vertexpermutation needs that); bipolar: the top row's north
        # vertices at the pole (gridtopology.jl:41).  The cell CENTRES -- all that distance_to_neighbour_2D reads -- are the known-answer points.
        le = np.array([p["lon1"] - 1.5, p["lon1"] - 0.5, p["lon1"] + 0.5, p["lon1"] + 1.5])
        te = np.array([p["lat1"] - 2.0, p["lat1"] - 0.5, min(p["lat1"], p["lat2"]) + abs(p["lat2"] - p["lat1"]) / 2, 90.0])
        lonv = np.zeros((4, nx, ny), order="F")
        latv = np.zeros((4, nx, ny), order="F")
        for i in range(nx):
            for j in range(ny):
                lonv[:, i, j] = (le[i], le[i + 1], le[i + 1], le[i])
                latv[:, i, j] = (te[j], te[j], te[j + 1], te[j + 1])
        area = Cube(np.ones((nx, ny), order="F"))
        vol = Cube(np.ones((nx, ny, nz), order="F"))
        asm = DeviceAssembler(0)
        asm.set_grid_from_raw(areacello=area, volcello=vol, lon=lon, lat=lat, lev=np.array([5.0]), lon_vertices=lonv, lat_vertices=latv,
                              mlotst=np.ones((nx, ny), order="F"), rho=1035.0)
        got = float(asm.dist[3].cpu().numpy().reshape((nx, ny), order="F")[1, 1])  # north neighbour of cell (2, 2)
        want, other = p["convert_then_subtract"]["distance"], p["subtract_then_convert"]["distance"]
        assert abs(got / want - 1) <= 1e-12, (got, want)
        assert abs(got / other - 1) > 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(topology="bipolar"), dict(vertex_order=(2, 3, 0, 1))])
def test_device_makegridmetrics_matches_oracle(oracle, kw):
    from otmb_amd.device import DeviceAssembler

    g = synthetic.make_grid(36, 30, 10, seed=6, rho="array", **kw)
    ref = oracle.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                 lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    from otmb_amd import NT

    gm = NT(**{k: v for k, v in ref.items() if k != "gridtopology"}, gridtopology=NT(kind=ref["gridtopology"]["kind"]))
    asm = DeviceAssembler(0)
    asm.set_grid_from_raw(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                          lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices, mlotst=g.mlotst, rho=g.rho)
    shp = gm.v3D.shape
    back = lambda t, s: t.cpu().numpy().reshape(s, order="F")
    assert asm.topology == gm.gridtopology.kind
    assert np.array_equal(back(asm.v3d, shp), gm.v3D, equal_nan=True)          # exact
    assert np.array_equal(back(asm.thk, shp), gm.thkcello, equal_nan=True)     # one IEEE division
    assert np.array_equal(back(asm.z3d, shp), gm.Z3D, equal_nan=True)          # same sequential sum
    assert np.array_equal(back(asm.area, shp[:2]), gm.area2D, equal_nan=True)
    for k, d in enumerate(("west", "east", "south", "north")):
        np.testing.assert_allclose(back(asm.edge[k], shp[:2]), gm.edge_length_2D[d], rtol=1e-12)
        np.testing.assert_allclose(back(asm.dist_edge[k], shp[:2]), gm.distance_to_edge_2D[d], rtol=1e-12)
        np.testing.assert_allclose(back(asm.dist[k], shp[:2]), gm.distance_to_neighbour_2D[d], rtol=1e-12, equal_nan=True)
    assert asm.N == int((~np.isnan(gm.v3D)).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(topology="bipolar"), dict(vertex_order=(3, 2, 1, 0))])
def test_host_pointer_makegridmetrics_entry_point_matches_oracle(oracle, kw):
    """otmb_makegridmetrics (host arrays in, host arrays out): what `makegridmetrics(...; gpu = true)` of the Julia shim calls, through its
    Python mirror api.makegridmetrics_gpu.  Exact arithmetic bit for bit against the oracle, haversines to 1e-12; and the result drives
    the rest of the path: the transportmatrix built from it equals the one built from the host metrics to 1e-12."""
    import otmb_amd.api as api

    g = synthetic.make_grid(36, 30, 10, seed=6, rho="array", **kw)
    args = dict(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev, lon_vertices=g.lon_vertices,
                lat_vertices=g.lat_vertices)
    ref = oracle.makegridmetrics(**args)
    got = api.makegridmetrics_gpu(**args)
    assert got.gridtopology.kind == ref["gridtopology"]["kind"]
    for k in ("v3D", "thkcello", "Z3D", "area2D", "lon_vertices", "lat_vertices", "lon", "lat", "zt"):
        assert np.array_equal(np.asarray(got[k]), np.asarray(ref[k]), equal_nan=True), k
    for group in ("edge_length_2D", "distance_to_edge_2D", "distance_to_neighbour_2D"):
        for d in ("west", "east", "south", "north"):
            np.testing.assert_allclose(got[group][d], ref[group][d], rtol=1e-12, equal_nan=True, err_msg=f"{group}[{d}]")
    idx = api.makeindices(got.v3D)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=got, indices=idx)
    tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=got, indices=idx, ρ=g.rho)
    hgm = otmb_amd.makegridmetrics(**args)
    tm0 = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=hgm, indices=idx, ρ=g.rho)
    for m in ("T", "Tadv", "TκH", "TκVML", "TκVdeep"):
        a, b = tuple(tm[m]), tuple(tm0[m])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), m
        np.testing.assert_allclose(a[2], b[2], rtol=1e-12, err_msg=m)
