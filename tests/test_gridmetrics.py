"""makegridmetrics (src/gridcellgeometry.jl:265-311): host numpy mirror vs the scalar Python transliteration
(CPU), vertex permutation / topology detection, and the device version vs the host one (GPU)."""
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
def test_device_makegridmetrics_matches_host(kw):
    from otmb_amd.device import DeviceAssembler

    g = synthetic.make_grid(36, 30, 10, seed=6, rho="array", **kw)
    gm = gridmetrics_of(g)
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
