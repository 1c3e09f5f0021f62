"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.  Bar: nonzero pattern (colptr,rowval) bit-exact;
values bit-exact as well (the path uses only + - * / min max abs with contraction off), which is
stricter than the 1e-12 relative tolerance BASELINE.json states."""
import numpy as np
import pytest

from helpers import CASES, MATS, assert_csc_equal, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    import otmb_amd.api as api

    return api


def _fill(g):
    return g.umo.properties["_FillValue"]


@pytest.mark.parametrize("name", list(CASES))
def test_makeindices_and_facefluxes_match_oracle(api, oracle, name):
    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    idx = api.makeindices(gm.v3D)
    assert idx.N == ref["N"]
    assert np.array_equal(idx.Lwet, ref["Lwet"])
    assert np.array_equal(idx.Lwet3D, ref["Lwet3D"])
    assert np.array_equal(idx.wet3D.view(np.uint8), ref["wet3D"])
    u0 = np.array(g.umo.data, copy=True)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
    assert np.array_equal(u0, g.umo.data, equal_nan=True)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    for k in rphi:
        assert phi[k].dtype == np.float64
        same = (phi[k] == rphi[k]) & (np.signbit(phi[k]) == np.signbit(rphi[k]))
        assert same.all(), (name, k, np.argwhere(~same)[:3])


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("upwind", [True, False])
def test_transportmatrix_matches_oracle_bit_for_bit(api, oracle, name, upwind):
    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)
    idx = api.makeindices(gm.v3D)
    tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH,
                             κVML=g.kappaVML, κVdeep=g.kappaVdeep, upwind=upwind)
    for m in MATS:
        assert tm[m].shape == (ref["N"], ref["N"])
        assert tm[m].colptr.dtype == np.int64 and tm[m].rowval.dtype == np.int64 and tm[m].nzval.dtype == np.float64
        assert_csc_equal(tuple(tm[m]), rtm[m], f"{name}/{m}/upwind={upwind}")


def test_exact_cancellation_is_dropped_from_T_but_kept_in_operators(api, oracle):
    """sparse() keeps stored zeros, + drops exact zeros: craft fluxes whose advective and diffusive
    off-diagonals cancel exactly in one entry."""
    g, gm = make_case("tiny_tripolar")
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    base = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, 0.0, 0.0, 0.0, True)
    # κ = 0 makes every diffusive value an explicit 0.0 (kept in TκH/TκV*, invisible in T)
    idx = api.makeindices(gm.v3D)
    tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=0.0, κVML=0.0, κVdeep=0.0)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), base[m], m)
    assert tm["TκH"].nnz > 0 and np.all(tm["TκH"].nzval == 0.0)
    assert not np.any(tm["T"].nzval == 0.0)
    assert tm["T"].nnz == tm["Tadv"].nnz


def test_error_strings_match_reference(api, oracle):
    from otmb_amd.capi import OtmbError

    g, gm = make_case("tiny_rho3d")
    ref = oracle.makeindices(gm.v3D)
    idx = api.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    kw = dict(mlotst=g.mlotst, gridmetrics=gm, indices=idx)
    rho = g.rho.copy(order="F")
    rho.ravel(order="F")[ref["Lwet"][5] - 1] = np.nan
    with pytest.raises(OtmbError, match="ρ contains NaNs"):
        api.transportmatrix(ϕ=rphi, ρ=rho, **kw)
    with pytest.raises(OtmbError, match="ρ contains NaNs"):
        api.transportmatrix(ϕ=rphi, ρ=float("nan"), **kw)
    wet = ref["wet3D"].astype(bool)
    bad = {k: v.copy(order="F") for k, v in rphi.items()}
    i, j, k = np.argwhere(wet & ~np.roll(wet, 1, axis=0))[0]
    bad["west"][i, j, k] = 5.0
    with pytest.raises(OtmbError, match="flux into a land cell") as e:
        api.transportmatrix(ϕ=bad, ρ=g.rho, **kw)
    assert e.value.name == "FLUX_INTO_LAND"
    bad = {k: v.copy(order="F") for k, v in rphi.items()}
    i, j = np.argwhere(wet[:, :, -1])[0]
    bad["bottom"][i, j, -1] = 1.0
    with pytest.raises(OtmbError, match="flux into a land cell"):
        api.transportmatrix(ϕ=bad, ρ=g.rho, **kw)
    gm2 = dict(gm)
    gm2["edge_length_2D"] = {d: a.copy(order="F") for d, a in gm.edge_length_2D.items()}
    ii, jj = np.argwhere(wet[:, :, 0] & np.roll(wet[:, :, 0], 1, axis=0))[0]
    gm2["edge_length_2D"]["west"][ii, jj] = np.nan
    with pytest.raises(OtmbError, match="TκH contains NaNs."):
        api.transportmatrix(ϕ=rphi, ρ=g.rho, mlotst=g.mlotst, gridmetrics=gm2, indices=idx)
    gm3 = dict(gm)
    gm3["gridtopology"] = dict(kind=2)
    with pytest.raises(OtmbError, match="Unknown grid type"):
        api.transportmatrix(ϕ=rphi, ρ=g.rho, mlotst=g.mlotst, gridmetrics=gm3, indices=idx)
    with pytest.raises(OtmbError, match="Unknown grid type"):
        api.facefluxes(g.umo.data, g.vmo.data, gm3, idx, FillValue=1e20)
    allwet = dict(wet3D=np.ones((4, 3, 2), np.bool_))
    with pytest.raises(OtmbError, match="AssertionError"):
        api.facefluxes(np.full((4, 3, 2), np.nan), np.ones((4, 3, 2)), dict(gridtopology=dict(kind=1)), allwet, FillValue=1e20)
    # Lwet3D that is not makeindices' ranking
    idx2 = dict(idx)
    lw = idx.Lwet3D.copy(order="F")
    flat = lw.ravel(order="F")
    a, b = ref["Lwet"][3] - 1, ref["Lwet"][4] - 1
    flat[a], flat[b] = flat[b], flat[a]
    idx2["Lwet3D"] = lw
    with pytest.raises(OtmbError, match="Lwet3D"):
        api.transportmatrix(ϕ=rphi, ρ=g.rho, mlotst=g.mlotst, gridmetrics=gm, indices=idx2)


# ---- golden fixtures ---------------------------------------------------------------------------
from golden_util import GOLDEN, load  # noqa: E402


@pytest.mark.parametrize("name", GOLDEN)
def test_hip_reproduces_golden_fixtures(api, name):
    gd = load(name)
    idx = api.makeindices(gd["gm"]["v3D"])
    assert np.array_equal(idx.Lwet, gd["z"]["Lwet"])
    phi = api.facefluxes(gd["umo"], gd["vmo"], gd["gm"], idx, FillValue=gd["fill"])
    for k in gd["phi"]:
        assert np.array_equal(phi[k], gd["phi"][k]), k
    kH, kML, kD = gd["kappa"]
    for upwind in (True, False):
        tm = api.transportmatrix(ϕ=gd["phi"], mlotst=gd["mlotst"], gridmetrics=gd["gm"], indices=idx, ρ=gd["rho"],
                                 κH=kH, κVML=kML, κVdeep=kD, upwind=upwind)
        for q, m in enumerate(MATS):
            assert_csc_equal(tuple(tm[m]), gd["tm"](upwind)[q], f"{name}/{m}")


# ---- device-resident driver: one-pass (look-back) and two-phase protocols -------------------------
@pytest.mark.parametrize("name", ["tiny_tripolar", "small_rho3d", "odd_nx_fold", "nx2", "tiny_bipolar"])
def test_device_onepass_and_twophase_match_oracle(oracle, name):
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    assert asm.N == ref["N"]
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    for onepass in (True, False, True):
        asm.step(umo, vmo, _fill(g), onepass=onepass)
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], rtm[m], f"{name}/{m}/onepass={onepass}")


def test_onepass_many_tiles(oracle):
    """A grid with several hundred tiles: more than one wave of the tile scan, the march order in use."""
    import torch

    from helpers import gridmetrics_of
    from otmb_amd import synthetic
    from otmb_amd.device import DeviceAssembler

    g = synthetic.make_grid(90, 80, 20, seed=77, rho="array")
    gm = gridmetrics_of(g)
    ref = oracle.makeindices(gm.v3D)
    assert ref["N"] > 64 * 256 * 2
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True, tight=True)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    for _ in range(3):
        asm.step(umo, vmo, 1e20, onepass=True)
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)


# ---- sparse add and precomputed operators (matrixbuilding.jl:133-147) ------------------------------
def test_spadd_matches_oracle(api, oracle):
    from otmb_amd.api import SparseMatrixCSC

    rng = np.random.default_rng(3)
    n = 700

    def rand():
        ln = 5000
        cp, rv, nz = oracle.sparse(rng.integers(1, n + 1, ln), rng.integers(1, n + 1, ln), rng.integers(-2, 3, ln).astype(float), n, n)
        return SparseMatrixCSC(n, n, cp, rv, nz)

    A, B = rand(), rand()
    Cm = api.spadd(A, B)
    ref = oracle.spadd(tuple(A), tuple(B), n)
    assert_csc_equal(tuple(Cm), ref, "spadd")
    assert not np.any(Cm.nzval == 0.0)
    E = SparseMatrixCSC(n, n, np.ones(n + 1, np.int64), np.zeros(0, np.int64), np.zeros(0))
    assert_csc_equal(tuple(api.spadd(A, E)), oracle.spadd(tuple(A), tuple(E), n), "spadd with empty")


def test_spadd_staged_and_unstaged_blocks(api, oracle):
    """Round 4's sparse add stages the entries of a workgroup's 256 columns in LDS when they fit (2040 per operand) and walks global memory
    otherwise: a matrix whose first columns are dense (unstaged blocks) and whose other columns hold a handful of entries (staged blocks), both
    ways round, with exact cancellations, against the oracle; and n not a multiple of 256."""
    from otmb_amd.api import SparseMatrixCSC

    rng = np.random.default_rng(12)
    n = 1100

    def rand(dense_cols, per_dense, ln_sparse):
        J = np.concatenate([rng.integers(1, dense_cols + 1, per_dense * dense_cols), rng.integers(dense_cols + 1, n + 1, ln_sparse)])
        I = rng.integers(1, n + 1, J.size)
        V = rng.integers(-2, 3, J.size).astype(float)
        cp, rv, nz = oracle.sparse(I, J, V, n, n)
        return SparseMatrixCSC(n, n, cp, rv, nz)

    A, B = rand(300, 40, 4000), rand(200, 3, 5000)
    assert int(A.colptr[256]) - 1 > 2040 and int(B.colptr[256]) - 1 <= 2040  # block 0: A does not fit, B does
    for X, Y, what in ((A, B, "A + B"), (B, A, "B + A"), (A, A, "A + A"), (B, B, "B + B")):
        assert_csc_equal(tuple(api.spadd(X, Y)), oracle.spadd(tuple(X), tuple(Y), n), what)
    neg = SparseMatrixCSC(n, n, A.colptr, A.rowval, -A.nzval)
    Z = api.spadd(A, neg)  # everything cancels exactly: an empty matrix
    assert Z.nzval.size == 0 and (Z.colptr == 1).all()


def test_transportmatrix_with_precomputed_operators(api, oracle):
    g, gm = make_case("tiny_rho3d")
    ref = oracle.makeindices(gm.v3D)
    idx = api.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    kw = dict(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho)
    base = api.transportmatrix(**kw)
    other = api.transportmatrix(κH=123.0, **kw)  # a different TκH, handed in as precomputed
    mixed = api.transportmatrix(TκH=other.TκH, **kw)
    assert mixed.TκH is other.TκH
    assert_csc_equal(tuple(mixed.Tadv), tuple(base.Tadv), "Tadv")
    want = oracle.spadd(oracle.spadd(oracle.spadd(tuple(base.Tadv), tuple(other.TκH), ref["N"]), tuple(base.TκVML), ref["N"]),
                        tuple(base.TκVdeep), ref["N"])
    assert_csc_equal(tuple(mixed.T), want, "T from precomputed TκH")
    assert_csc_equal(tuple(mixed.T), tuple(other.T), "same as building with κH=123")
    for m in ("TκVML", "TκVdeep"):
        assert_csc_equal(tuple(mixed[m]), tuple(base[m]), m)
    # only the MISSING operators are built (matrixbuilding.jl:140-143): with Tadv handed in, ϕ and ρ are never looked at,
    # so a NaN in ρ or a flux into land is not an error, exactly as in the reference
    bad_phi = {k: v.copy(order="F") for k, v in rphi.items()}
    wet = ref["wet3D"].astype(bool)
    i, j, k = np.argwhere(wet & ~np.roll(wet, 1, axis=0))[0]
    bad_phi["west"][i, j, k] = 5.0
    rho_nan = g.rho.copy(order="F")
    rho_nan.ravel(order="F")[ref["Lwet"][5] - 1] = np.nan
    kw2 = dict(mlotst=g.mlotst, gridmetrics=gm, indices=idx)
    with_adv = api.transportmatrix(ϕ=bad_phi, ρ=rho_nan, Tadv=base.Tadv, **kw2)
    assert with_adv.Tadv is base.Tadv
    assert_csc_equal(tuple(with_adv.T), tuple(base.T), "T with Tadv given")
    all_given = api.transportmatrix(ϕ=None, ρ=None, mlotst=None, gridmetrics=gm, indices=idx, Tadv=base.Tadv, TκH=base.TκH,
                                    TκVML=base.TκVML, TκVdeep=base.TκVdeep)
    assert_csc_equal(tuple(all_given.T), tuple(base.T), "T from four given operators")
    from otmb_amd.capi import OtmbError

    with pytest.raises(OtmbError, match="ρ contains NaNs"):  # TκH given, Tadv not: ρ is checked (matrixbuilding.jl:233)
        api.transportmatrix(ϕ=rphi, ρ=rho_nan, TκH=base.TκH, **kw2)
    # a given TκH is never built either: a NaN edge length is not "TκH contains NaNs." -- but it is once TκH must be built
    gm_nan = dict(gm)
    gm_nan["edge_length_2D"] = {d: a.copy(order="F") for d, a in gm.edge_length_2D.items()}
    ii, jj = np.argwhere(wet[:, :, 0] & np.roll(wet[:, :, 0], 1, axis=0))[0]
    gm_nan["edge_length_2D"]["west"][ii, jj] = np.nan
    with_h = api.transportmatrix(ϕ=rphi, ρ=g.rho, mlotst=g.mlotst, gridmetrics=gm_nan, indices=idx, TκH=base.TκH)
    assert_csc_equal(tuple(with_h.T), tuple(base.T), "T with TκH given and a NaN metric that only TκH reads")
    with pytest.raises(OtmbError, match="TκH contains NaNs."):
        api.transportmatrix(ϕ=rphi, ρ=g.rho, mlotst=g.mlotst, gridmetrics=gm_nan, indices=idx, TκVdeep=base.TκVdeep)


# ---- edge cases: empty, single level, single wet cell ------------------------------------------------
def _edge_grid(kind):
    from helpers import gridmetrics_of, randomize_metrics
    from otmb_amd import synthetic

    if kind == "nz1":
        g = synthetic.make_grid(10, 8, 1, seed=41, land_fraction=0.3)
        gm = gridmetrics_of(g)
    else:
        g = synthetic.make_grid(8, 6, 4, seed=42, land_fraction=0.3)
        gm = gridmetrics_of(g)
        v = gm.v3D
        if kind == "all_land":
            v[:] = np.nan
        elif kind == "one_wet":
            keep = np.argwhere(~np.isnan(v))[7]
            val = v[tuple(keep)]
            v[:] = np.nan
            v[tuple(keep)] = val
        gm.thkcello[np.isnan(v)] = np.nan
    return g, gm


@pytest.mark.parametrize("kind", ["nz1", "all_land", "one_wet"])
def test_edge_grids(api, oracle, kind):
    g, gm = _edge_grid(kind)
    ref = oracle.makeindices(gm.v3D)
    idx = api.makeindices(gm.v3D)
    assert idx.N == ref["N"] and np.array_equal(idx.Lwet, ref["Lwet"]) and np.array_equal(idx.Lwet3D, ref["Lwet3D"])
    if kind == "all_land":
        assert idx.N == 0
        # facefluxes' assertion fires in the reference as well only if nothing is valid; land is zeroed first
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    phi = api.facefluxes(g.umo.data, g.vmo.data, gm, idx, FillValue=_fill(g))
    for k in rphi:
        assert np.array_equal(phi[k], rphi[k]), k
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho)
    for m in MATS:
        assert tm[m].shape == (ref["N"], ref["N"])
        assert_csc_equal(tuple(tm[m]), rtm[m], f"{kind}/{m}")


def test_plan_survives_other_calls_before_fill(oracle):
    """plan -> (other library calls that use scratch scans) -> fill must still be correct."""
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm = make_case("small_rho3d")
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    phi = asm.facefluxes(umo, vmo, _fill(g))
    asm.plan(phi)
    keep = (asm.lwet3d, asm.lwet, asm.wet3d)  # the plan points at these: they must outlive fill
    asm.makeindices()  # same context: runs the scratch tile scans between plan and fill
    assert keep[0].data_ptr() != asm.lwet3d.data_ptr()
    asm.fill()
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)


def test_async_pipeline_of_steps(oracle):
    """step_async x3 without host synchronisation, then finish(): same result, errors still surface."""
    import torch

    from otmb_amd.capi import OtmbError
    from otmb_amd.device import DeviceAssembler

    g, gm = make_case("small_rho3d")
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    for _ in range(3):
        asm.step_async(umo, vmo, _fill(g))
    asm.finish()
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)
    bad = torch.full_like(umo, float("nan"))
    asm.wet3d.fill_(1)  # all wet + all NaN -> the reference's assertion
    asm.facefluxes_async(bad, vmo, _fill(g))
    asm.transportmatrix_onepass(asm.phi, sync=False)
    with pytest.raises(OtmbError, match="AssertionError"):
        asm.finish()


# ---- randomised sweep: many small grids, both topologies, odd/even nx, zero and missing fluxes --------------
@pytest.mark.parametrize("seed", range(24))
def test_random_small_grids(api, oracle, seed):
    from helpers import gridmetrics_of, randomize_metrics
    from otmb_amd import synthetic

    rng = np.random.default_rng(1000 + seed)
    nx, ny, nz = int(rng.integers(3, 11)), int(rng.integers(2, 8)), int(rng.integers(1, 7))
    topo = "tripolar" if rng.random() < 0.7 else "bipolar"
    rho = "array" if rng.random() < 0.5 else "scalar"
    g = synthetic.make_grid(nx, ny, nz, seed=int(rng.integers(1 << 30)), land_fraction=float(rng.uniform(0.0, 0.6)), topology=topo, rho=rho)
    gm = randomize_metrics(gridmetrics_of(g), seed=seed)
    umo, vmo = g.umo.data.copy(order="F"), g.vmo.data.copy(order="F")
    wet = ~np.isnan(gm.v3D)
    z = rng.random(umo.shape) < 0.15  # exact zeros: faces without any flux
    umo[z & wet] = 0.0
    vmo[(rng.random(umo.shape) < 0.15) & wet] = 0.0
    umo[(rng.random(umo.shape) < 0.05) & wet] = np.nan  # missing inside the ocean
    if not np.any(wet):
        pytest.skip("all land")
    ref = oracle.makeindices(gm.v3D)
    idx = api.makeindices(gm.v3D)
    assert np.array_equal(idx.Lwet3D, ref["Lwet3D"])
    try:
        rphi = oracle.facefluxes(umo, vmo, ref["wet3D"], 1e20, gm.gridtopology.kind)
    except oracle.OracleError:
        with pytest.raises(Exception):
            api.facefluxes(umo, vmo, gm, idx, FillValue=1e20)
        return
    phi = api.facefluxes(umo, vmo, gm, idx, FillValue=1e20)
    for k in rphi:
        assert np.array_equal(phi[k], rphi[k]), k
    upwind = bool(rng.random() < 0.6)
    kw = dict(mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=float(rng.choice([0.0, 500.0])), κVML=0.1, κVdeep=1e-5, upwind=upwind)
    try:
        rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, kw["κH"], 0.1, 1e-5, upwind)
    except oracle.OracleError as e:
        from otmb_amd.capi import OtmbError

        with pytest.raises(OtmbError) as ei:
            api.transportmatrix(ϕ=rphi, **kw)
        assert {-1: 1, -2: 2, -3: 3, -4: 4, -5: 5, -6: 6}[e.code] == ei.value.status
        return
    tm = api.transportmatrix(ϕ=rphi, **kw)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], f"seed {seed} {nx}x{ny}x{nz} {topo} {m}")


def test_three_launch_tile_scan_path(oracle, monkeypatch):
    """Grids of up to 1024 tiles take the single-launch tile scan; force the three-launch path of the larger grids
    (otmb_scan.hip) on a test-sized grid."""
    import torch

    from helpers import gridmetrics_of
    from otmb_amd import synthetic
    from otmb_amd.device import DeviceAssembler

    monkeypatch.setenv("OTMB_SCAN_SINGLE_MAX", "0")
    g = synthetic.make_grid(90, 80, 20, seed=78, rho="array")
    gm = gridmetrics_of(g)
    ref = oracle.makeindices(gm.v3D)
    assert 256 < ref["N"] < 1024 * 256  # several tiles, one scan group: the single-launch path would be taken
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True, tight=True)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    assert asm.N == ref["N"]
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    for onepass in (True, False):
        asm.step(umo, vmo, 1e20, onepass=onepass)
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], rtm[m], f"{m}/onepass={onepass}")


def test_rho_nan_is_reported_by_both_device_protocols(oracle):
    """The asynchronous protocol checks ρ in the fill pass (which loads it anyway), the two-phase protocol in the
    counting pass (plan must already report it): same error, and it wins over a flux into land (reference order:
    src/matrixbuilding.jl:233 comes before the loop)."""
    import torch

    from otmb_amd.capi import OtmbError
    from otmb_amd.device import DeviceAssembler

    g, gm = make_case("small_rho3d")
    ref = oracle.makeindices(gm.v3D)
    rho = g.rho.copy(order="F")
    rho.ravel(order="F")[ref["Lwet"][ref["N"] // 2] - 1] = np.nan
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    for onepass in (True, False):
        with pytest.raises(OtmbError, match="ρ contains NaNs"):
            asm.step(umo, vmo, _fill(g), onepass=onepass)
    # double fault: also push a flux into land
    phi = asm.facefluxes(umo, vmo, _fill(g))
    wet = ref["wet3D"].astype(bool)
    i, j, k = np.argwhere(wet & ~np.roll(wet, 1, axis=0))[0]
    L = i + wet.shape[0] * (j + wet.shape[1] * k)
    phi[1][L] = 5.0  # OTMB_WEST; in place through torch, so the assembler drops its push mask
    for twophase in (False, True):
        with pytest.raises(OtmbError, match="ρ contains NaNs"):
            asm.transportmatrix(phi) if twophase else asm.transportmatrix_onepass(phi)


def test_set_grid_tensors_equals_set_grid(oracle):
    """Device-resident callers hand over tensors (tools/large_run_device.py does, for the 0.1 degree grid): same result
    as uploading the host arrays."""
    import torch

    from otmb_amd.capi import HDIRS
    from otmb_amd.device import DeviceAssembler

    g, gm = make_case("small_rho3d")
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)

    def dev(a):
        return torch.from_numpy(np.asfortranarray(a, dtype=np.float64).ravel(order="F").copy()).cuda()

    asm = DeviceAssembler(0)
    asm.set_grid_tensors(shape=gm.v3D.shape, topology=gm.gridtopology.kind, v3d=dev(gm.v3D), thkcello=dev(gm.thkcello),
                         edge_length=[dev(gm.edge_length_2D[d]) for d in HDIRS],
                         dist_nbr=[dev(gm.distance_to_neighbour_2D[d]) for d in HDIRS], area2d=dev(gm.area2D),
                         zt=dev(gm.zt), mlotst=dev(g.mlotst), rho=dev(g.rho), kappaH=g.kappaH, kappaVML=g.kappaVML,
                         kappaVdeep=g.kappaVdeep)
    assert asm.N == ref["N"]
    asm.step(dev(g.umo.data), dev(g.vmo.data), _fill(g))
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)
    with pytest.raises(ValueError):
        asm.set_grid_tensors(shape=gm.v3D.shape, topology=1, v3d=dev(gm.v3D).float(), thkcello=dev(gm.thkcello),
                             edge_length=[dev(gm.edge_length_2D[d]) for d in HDIRS],
                             dist_nbr=[dev(gm.distance_to_neighbour_2D[d]) for d in HDIRS], area2d=dev(gm.area2D),
                             zt=dev(gm.zt), mlotst=dev(g.mlotst), rho=1035.0)


@pytest.mark.parametrize("name", ["tiny_tripolar", "small_rho3d", "odd_nx_fold"])
def test_t_only_extension_gives_the_same_T(api, oracle, name):
    """operators=False / otmb_tm_args.only_t: the four operators are evaluated but not materialised; T is unchanged."""
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], _fill(g), gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    idx = api.makeindices(gm.v3D)
    tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH, κVML=g.kappaVML,
                             κVdeep=g.kappaVdeep, operators=False)
    assert_csc_equal(tuple(tm["T"]), rtm["T"], name)
    assert all(tm[m] is None for m in MATS[1:])
    asm = DeviceAssembler(0)
    asm.only_T = True
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    for onepass in (True, False):
        asm.step(umo, vmo, _fill(g), onepass=onepass)
        assert asm.nnz[1:] == [0, 0, 0, 0]
        cp, rv, nz = asm.out["T"]
        n = asm.nnz[0]
        assert_csc_equal((cp.cpu().numpy(), rv[:n].cpu().numpy(), nz[:n].cpu().numpy()), rtm["T"], f"{name}/onepass={onepass}")
    asm.only_T = False
    asm.step(umo, vmo, _fill(g))
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)
