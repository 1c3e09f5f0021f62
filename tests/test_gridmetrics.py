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
