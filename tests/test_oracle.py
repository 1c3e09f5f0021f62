"""CPU tests of the oracle (C restatement) -- run without a GPU.

The reference holds no golden vectors for this path, so the oracle is
pinned by (1) an independent pure-Python transliteration that must agree bit for bit,
(2) scipy for the sparse()/+ contracts, (3) the physical properties the reference's own
tests assert (test/online.jl:93-123), (4) golden fixtures generated from it (regression).
"""
import math

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import CASES, MATS, assert_csc_equal, make_case
from oracle import pyref


def _pipeline(orc, name, upwind=True):
    g, gm = make_case(name)
    idx = orc.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    phi = orc.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], fill, gm.gridtopology.kind)
    tm = orc.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)
    return g, gm, idx, phi, tm


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("upwind", [True, False])
def test_oracle_matches_python_transliteration(oracle, name, upwind):
    g, gm, idx, phi, tm = _pipeline(oracle, name, upwind)
    pidx = pyref.makeindices(gm.v3D)
    assert pidx["Lwet"] == list(idx["Lwet"])
    topo = pyref.Topo(gm.gridtopology.kind, g.nx, g.ny, g.nz)
    pphi = pyref.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], g.umo.properties["_FillValue"], topo)
    for k in phi:
        assert np.array_equal(phi[k], pphi[k]), k
    ptm = pyref.transportmatrix(phi, gm, pidx, topo, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)
    for m in MATS:
        assert_csc_equal(tm[m], ptm[m], f"{name}/{m}")


def test_coo_generators_emission_order(oracle):
    """The three *_sparse_entries generators, triplet by triplet in the reference's push order."""
    g, gm = make_case("tiny_rho3d")
    idx = oracle.makeindices(gm.v3D)
    kind = gm.gridtopology.kind
    phi = oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], 1e20, kind)
    pidx = pyref.makeindices(gm.v3D)
    topo = pyref.Topo(kind, g.nx, g.ny, g.nz)
    for upwind in (True, False):
        a = oracle.advection_entries(phi, gm.v3D, g.rho, idx["Lwet"], idx["Lwet3D"], kind, upwind)
        b = pyref.advection_entries(phi, gm.v3D, g.rho, pidx, topo, upwind)
        for x, y in zip(a, b):
            assert np.array_equal(x, np.array(y))
    a = oracle.hdiff_entries(gm.v3D, gm.thkcello, gm.edge_length_2D, gm.distance_to_neighbour_2D, idx["Lwet"], idx["Lwet3D"], kind, 500.0)
    b = pyref.hdiff_entries(gm, pidx, topo, 500.0)
    for x, y in zip(a, b):
        assert np.array_equal(x, np.array(y))
    Om = oracle.ml_mask(gm.zt, g.mlotst, idx["Lwet"], gm.v3D.shape)
    assert list(Om.astype(bool)) == pyref.ml_mask(gm.zt, g.mlotst, pidx)
    for om in (Om, None):
        a = oracle.vdiff_entries(gm.v3D, gm.area2D, gm.zt, idx["Lwet"], idx["Lwet3D"], kind, 0.1, om)
        b = pyref.vdiff_entries(gm, pidx, topo, 0.1, None if om is None else list(om))
        for x, y in zip(a, b):
            assert np.array_equal(x, np.array(y))


def test_sparse_contract_vs_scipy(oracle):
    """sparse(I,J,V,m,n): pattern = distinct pairs (explicit zeros kept), rows ascending, values = in-order sums."""
    rng = np.random.default_rng(0)
    m = n = 40
    ln = 600
    I = rng.integers(1, m + 1, ln)
    J = rng.integers(1, n + 1, ln)
    V = rng.standard_normal(ln)
    V[::7] = 0.0  # stored zeros must survive
    I[-2:] = I[:2]; J[-2:] = J[:2]; V[-2:] = -V[:2]  # exact cancellation must survive too
    cp, rv, nz = oracle.sparse(I, J, V, m, n)
    ref = sp.coo_matrix((V, (I - 1, J - 1)), shape=(m, n)).tocsc()
    ref.sum_duplicates(); ref.sort_indices()
    assert np.array_equal(cp - 1, ref.indptr) and np.array_equal(rv - 1, ref.indices)
    np.testing.assert_allclose(nz, ref.data, rtol=1e-13, atol=1e-15)
    # in-order left fold, bit for bit
    acc = {}
    for i, j, v in zip(I, J, V):
        acc[(j, i)] = acc[(j, i)] + v if (j, i) in acc else v
    assert np.array_equal(nz, np.array([acc[k] for k in sorted(acc)]))
    for c in range(n):
        rows = rv[cp[c] - 1: cp[c + 1] - 1]
        assert np.all(np.diff(rows) > 0)
    # empty input
    cp, rv, nz = oracle.sparse(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0), 5, 5)
    assert list(cp) == [1] * 6 and len(rv) == 0


def test_spadd_contract_vs_scipy(oracle):
    rng = np.random.default_rng(1)
    n = 30
    def rand():
        ln = 200
        return oracle.sparse(rng.integers(1, n + 1, ln), rng.integers(1, n + 1, ln), rng.integers(-2, 3, ln).astype(float), n, n)
    A, B = rand(), rand()
    C = oracle.spadd(A, B, n)
    ref = (oracle.to_scipy(A, n) + oracle.to_scipy(B, n)).tocsc()
    ref.eliminate_zeros(); ref.sort_indices()
    assert np.array_equal(C[0] - 1, ref.indptr) and np.array_equal(C[1] - 1, ref.indices) and np.array_equal(C[2], ref.data)
    assert not np.any(C[2] == 0.0)  # + drops exact zeros (stored zeros of A or B included)
    assert_csc_equal(C, pyref.spadd(A, B, n))


@pytest.mark.parametrize("name", ["tiny_tripolar", "tiny_bipolar", "small_rho3d"])
def test_reference_physical_properties(oracle, name):
    """test/online.jl:93-123 restated: divergence-free diffusion, volume conservation, sign structure."""
    g, gm, idx, phi, tm = _pipeline(oracle, name, True)
    N = idx["N"]
    v = np.asarray(gm.v3D).ravel(order="F")[idx["Lwet"] - 1]
    e1 = np.ones(N)
    Myr = 365.25 * 86400 * 1e6
    for m in MATS:
        M = oracle.to_scipy(tm[m], N)
        assert tm[m][0].dtype == np.int64 and tm[m][1].dtype == np.int64 and tm[m][2].dtype == np.float64
        if m not in ("T", "Tadv"):
            tau_div = np.linalg.norm(e1) / max(np.linalg.norm(M @ e1), 1e-300) / Myr
            assert tau_div > 1e6, (m, tau_div)  # online.jl:110-111
        if np.ndim(g.rho) == 0 or m not in ("T", "Tadv"):
            tau_vol = np.linalg.norm(v) / max(np.linalg.norm(M.T @ v), 1e-300) / Myr
            assert tau_vol > 1e6, (m, tau_vol)  # online.jl:114-115
    T = oracle.to_scipy(tm["T"], N)
    d = T.diagonal()
    assert np.all(d > 0)  # online.jl:122
    off = (T - sp.diags(d)).tocsc(); off.eliminate_zeros()
    assert np.all(off.data < 0)  # online.jl:123
    # T == ((Tadv+TκH)+TκVML)+TκVdeep bit for bit, no stored zeros
    S = oracle.spadd(oracle.spadd(oracle.spadd(tm["Tadv"], tm["TκH"], N), tm["TκVML"], N), tm["TκVdeep"], N)
    assert_csc_equal(tm["T"], S)


def test_facefluxes_properties(oracle):
    g, gm = make_case("tiny_tripolar")
    idx = oracle.makeindices(gm.v3D)
    u0, v0 = g.umo.data.copy(), g.vmo.data.copy()
    phi = oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], 1e20, gm.gridtopology.kind)
    assert np.array_equal(u0, g.umo.data) and np.array_equal(v0, g.vmo.data)  # caller's arrays untouched
    wet = idx["wet3D"].astype(bool)
    for k in phi:
        assert np.all(np.isfinite(phi[k]))
    assert np.all(phi["east"][~wet] == 0) and np.all(phi["north"][~wet] == 0)
    assert np.all(phi["bottom"][:, :, -1] == 0)
    assert np.array_equal(phi["bottom"][:, :, :-1], phi["top"][:, :, 1:])
    assert np.array_equal(phi["west"], np.roll(phi["east"], 1, axis=0))
    assert np.all(phi["south"][:, 0, :] == 0) and np.array_equal(phi["south"][:, 1:, :], phi["north"][:, :-1, :])
    # no flux through a face whose other side is land: east face of (i) with land at i+1
    land_e = ~np.roll(wet, -1, axis=0)
    assert np.all(phi["east"][land_e] == 0)


def test_error_paths(oracle):
    g, gm = make_case("tiny_rho3d")
    idx = oracle.makeindices(gm.v3D)
    kind = gm.gridtopology.kind
    phi = oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], 1e20, kind)
    rho = g.rho.copy(order="F")
    L = idx["Lwet"][5] - 1
    rho.ravel(order="F")[L] = np.nan
    with pytest.raises(oracle.OracleError, match="ρ contains NaNs"):
        oracle.transportmatrix(phi, gm, idx, rho, g.mlotst)
    with pytest.raises(oracle.OracleError, match="ρ contains NaNs"):
        oracle.transportmatrix(phi, gm, idx, float("nan"), g.mlotst)
    # a flux pointing into land (user bypassed facefluxes)
    bad = {k: v.copy(order="F") for k, v in phi.items()}
    wet = idx["wet3D"].astype(bool)
    cand = np.argwhere(wet & ~np.roll(wet, 1, axis=0))  # wet cell whose west neighbour is land
    i, j, k = cand[0]
    bad["west"][i, j, k] = 5.0
    with pytest.raises(oracle.OracleError, match="flux into land"):
        oracle.transportmatrix(bad, gm, idx, g.rho, g.mlotst)
    # bottom flux at the deepest level -> k₊₁ is nothing
    bad = {k: v.copy(order="F") for k, v in phi.items()}
    deep = np.argwhere(wet[:, :, -1])
    if len(deep):
        i, j = deep[0]
        bad["bottom"][i, j, -1] = 1.0
        with pytest.raises(oracle.OracleError, match="flux into land"):
            oracle.transportmatrix(bad, gm, idx, g.rho, g.mlotst)
    # NaN metric -> "TκH contains NaNs."
    gm2 = dict(gm); gm2["edge_length_2D"] = {d: a.copy(order="F") for d, a in gm.edge_length_2D.items()}
    ii, jj = np.argwhere(wet[:, :, 0] & np.roll(wet[:, :, 0], 1, axis=0))[0]
    gm2["edge_length_2D"]["west"][ii, jj] = np.nan
    with pytest.raises(oracle.OracleError, match="TκH contains NaNs"):
        oracle.transportmatrix(phi, gm2, idx, g.rho, g.mlotst)
    with pytest.raises(oracle.OracleError, match="Unknown grid type"):
        oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], 1e20, 2)
    with pytest.raises(oracle.OracleError, match="all fluxes missing"):
        # nofluxboundaries! zeroes land cells first, so the assert only fires on an all-wet grid
        allwet = np.ones((4, 3, 2), np.uint8)
        oracle.facefluxes(np.full((4, 3, 2), np.nan), np.ones((4, 3, 2)), allwet, 1e20, 1)


def test_haversine_known_answers(oracle):
    R = 6371000.0
    assert oracle.haversine(0, 0, 0, 0) == 0.0
    assert math.isclose(oracle.haversine(0, 0, 0, 90), math.pi * R / 2, rel_tol=1e-15)
    assert math.isclose(oracle.haversine(0, 0, 180, 0), math.pi * R, rel_tol=1e-15)
    assert math.isclose(oracle.haversine(10, 0, 11, 0), math.pi * R / 180, rel_tol=1e-13)
    assert math.isclose(oracle.haversine(-73.97, 40.78, 2.35, 48.86), pyref.haversine((-73.97, 40.78), (2.35, 48.86)), rel_tol=1e-15)


# ---- golden fixtures (regression vectors made by tests/golden/make_golden.py) ------------------
from golden_util import GOLDEN, load  # noqa: E402


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_reproduces_golden_fixtures(oracle, name):
    gd = load(name)
    idx = oracle.makeindices(gd["gm"]["v3D"])
    assert np.array_equal(idx["Lwet"], gd["z"]["Lwet"])
    phi = oracle.facefluxes(gd["umo"], gd["vmo"], idx["wet3D"], gd["fill"], gd["gm"]["gridtopology"]["kind"])
    for k in phi:
        assert np.array_equal(phi[k], gd["phi"][k]), k
    kH, kML, kD = gd["kappa"]
    for upwind in (True, False):
        tm = oracle.transportmatrix(gd["phi"], gd["gm"], idx, gd["rho"], gd["mlotst"], kH, kML, kD, upwind)
        for q, m in enumerate(MATS):
            assert_csc_equal(tm[m], gd["tm"](upwind)[q], f"{name}/{m}")


@pytest.mark.parametrize("name", ["tiny_tripolar", "small_rho3d", "odd_nx_fold"])
def test_multithreaded_oracle_variant_is_bit_identical(oracle, name):
    """orc_transportmatrix_omp (bench.py's multi-thread CPU figure): concurrent operator builds and column-parallel adds
    must not change a bit, and errors keep the reference's order."""
    from helpers import MATS, assert_csc_equal, make_case

    g, gm = make_case(name)
    idx = oracle.makeindices(gm.v3D)
    phi = oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], g.umo.properties["_FillValue"], gm.gridtopology.kind)
    for upwind in (True, False):
        a = oracle.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)
        b = oracle.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind, parallel=True)
        for m in MATS:
            assert_csc_equal(b[m], a[m], f"{name}/{m}/upwind={upwind}")
    assert oracle.omp_threads() >= 1
    rho = np.full(gm.v3D.shape, np.nan)
    with pytest.raises(oracle.OracleError) as e1:
        oracle.transportmatrix(phi, gm, idx, rho, g.mlotst, parallel=False)
    with pytest.raises(oracle.OracleError) as e2:
        oracle.transportmatrix(phi, gm, idx, rho, g.mlotst, parallel=True)
    assert e1.value.args == e2.value.args


# ---- lump_and_spray (src/extratools.jl:38-119) -------------------------------------------------------------------
LUMP_SETTINGS = [(2, 2, 1, False), (3, 2, 2, True), (1, 1, 1, False), (4, 5, 1, True), (2, 3, 3, True), (10, 10, 1, False)]


def lump_inputs(oracle, name):
    from helpers import make_case

    g, gm = make_case(name)
    idx = oracle.makeindices(gm.v3D)
    phi = oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], g.umo.properties["_FillValue"], gm.gridtopology.kind)
    tm = oracle.transportmatrix(phi, gm, idx, g.rho, g.mlotst)
    wet = idx["wet3D"].astype(bool)
    vol = gm.v3D.reshape(-1, order="F")[wet.reshape(-1, order="F")]
    return wet, vol, tm, idx["N"]


def lump_mask(wet, seed):
    rng = np.random.default_rng(seed)
    mask = rng.random(wet.shape) < 0.6
    mask[:, : wet.shape[1] // 3, :] = False
    return mask


@pytest.mark.parametrize("name", ["tiny_tripolar", "tiny_bipolar", "odd_nx_fold"])
def test_lump_and_spray_oracle_matches_transliteration(oracle, name):
    wet, vol, tm, N = lump_inputs(oracle, name)
    for q, (di, dj, dk, usemask) in enumerate(LUMP_SETTINGS):
        mask = lump_mask(wet, q) if usemask else None
        a = oracle.lump_and_spray(wet, vol, tm["T"], mask, di, dj, dk)
        b = pyref.lump_and_spray(wet, vol, tm["T"], mask, di, dj, dk)
        for x, y in zip(a[0] + a[1] + (a[2],), b[0] + b[1] + (b[2],)):
            assert np.array_equal(x, y), (name, di, dj, dk, usemask)
        # what the reference's docstring promises: LUMP is volume conserving and LUMP * SPRAY = I
        LUMP, SPRAY, vol_c = a
        Nc = len(vol_c)
        Lm = sp.csc_matrix((LUMP[2], LUMP[1] - 1, LUMP[0] - 1), shape=(Nc, N))
        Sm = sp.csc_matrix((SPRAY[2], SPRAY[1] - 1, SPRAY[0] - 1), shape=(N, Nc))
        assert abs(Lm @ Sm - sp.identity(Nc)).max() < 1e-14
        assert np.allclose(Lm.T @ vol_c, vol, rtol=1e-14)
        if (di, dj, dk) == (1, 1, 1) and mask is None:
            assert Nc == N


def test_lump_and_spray_rejects_asymmetric_connectivity(oracle):
    """An advection-only (upwind) matrix has a one-directional pattern: Graphs.SimpleGraph throws ArgumentError."""
    wet, vol, tm, N = lump_inputs(oracle, "tiny_tripolar")
    with pytest.raises(oracle.OracleError, match="symmetric"):
        oracle.lump_and_spray(wet, vol, tm["Tadv"], None, 2, 2, 1)
    with pytest.raises(pyref.AsymmetricConnectivity):
        pyref.lump_and_spray(wet, vol, tm["Tadv"], None, 2, 2, 1)


def test_as2d_as3d(oracle):
    wet, vol, tm, N = lump_inputs(oracle, "tiny_tripolar")
    x = np.arange(1.0, N + 1)
    x3 = pyref.as3D(x, wet)
    assert np.array_equal(x3.reshape(-1, order="F")[wet.reshape(-1, order="F")], x) and np.isnan(x3[~wet]).all()
    n2 = int(wet[:, :, 0].sum())
    x2 = pyref.as2D(np.arange(1.0, n2 + 1), wet)
    assert np.array_equal(x2.reshape(-1, order="F")[wet[:, :, 0].reshape(-1, order="F")], np.arange(1.0, n2 + 1))


def lump_fixtures():
    import glob
    import os

    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lump", "*.npz")))


@pytest.mark.parametrize("path", lump_fixtures(), ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_reproduces_lump_fixtures(oracle, path):
    z = np.load(path)
    di, dj, dk = (int(x) for x in z["block"])
    N = len(z["vol"])
    T = (z["T_colptr"], z["T_rowval"], np.ones(len(z["T_rowval"])))
    LUMP, SPRAY, vol_c = oracle.lump_and_spray(z["wet3D"], z["vol"], T, z["mask"], di, dj, dk)
    assert np.array_equal(LUMP[1], z["lump_rowval"]) and np.array_equal(LUMP[2], z["lump_nzval"])
    assert np.array_equal(SPRAY[0], z["spray_colptr"]) and np.array_equal(SPRAY[1], z["spray_rowval"])
    assert np.array_equal(vol_c, z["vol_c"]) and np.array_equal(LUMP[0], np.arange(1, N + 2))


def test_static_capacity_bounds_every_matrix_and_is_exact_where_the_mask_decides(oracle):
    """otmb_static_capacity (pure host arithmetic in the product library: runs without a GPU): from the wet mask alone, counts that every
    build on that grid stays under -- whatever the fluxes, κ and mixed layer -- and that ARE TκH's, TκVdeep's and the union pattern's counts
    wherever no two row-mates coincide (no tripolar seam aliasing, nx >= 3)."""
    import ctypes as C

    from helpers import CASES, make_case
    from otmb_amd import capi

    lib = capi.lib()
    for name in CASES:
        g, gm = make_case(name)
        idx = oracle.makeindices(gm.v3D)
        wet = np.asfortranarray(idx["wet3D"]).view(np.uint8)
        cap = (C.c_int64 * 5)()
        assert lib.otmb_static_capacity(wet.ctypes.data, *wet.shape, int(gm.gridtopology.kind), C.byref(cap)) == 0
        phi = oracle.facefluxes(np.asarray(g.umo.data, dtype=np.float64), np.asarray(g.vmo.data, dtype=np.float64), idx["wet3D"],
                                float(g.umo.properties["_FillValue"]), gm.gridtopology.kind)
        for upwind in (True, False):
            tm = oracle.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)
            got = [len(tm[m][1]) for m in ("T", "Tadv", "TκH", "TκVML", "TκVdeep")]
            assert all(c >= n for c, n in zip(cap, got)), (name, list(cap), got)
            assert cap[4] == got[4], (name, "TκVdeep")
            if name in ("tiny_bipolar",):  # (no fold: nothing coincides)
                assert cap[2] == got[2], (name, "TκH")
                if upwind is False:
                    assert cap[0] >= got[0]
    bad = (C.c_int64 * 5)()
    assert lib.otmb_static_capacity(None, 3, 3, 3, 0, C.byref(bad)) != 0
    assert lib.otmb_static_capacity(wet.ctypes.data, *wet.shape, 2, C.byref(bad)) != 0  # UnknownGridTopology has no neighbours to count
