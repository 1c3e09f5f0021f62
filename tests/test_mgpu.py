"""Depth slabs over several GPUs of ONE process, behind the C ABI (otmb_mgpu_*, include/otmb.h): what a Julia
caller reaches with `transportmatrix(...; devices = 0:7)`.  On a one-GPU box N contexts share GPU 0 and the facefluxes chain's
planes are handed over by device-to-device copies; with every slab on its own GPU the same code hands them over with grouped
ncclSend / ncclRecv on a single-process RCCL communicator (the last test: skipped below 2 GPUs).  Whole-grid oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from helpers import MATS, assert_csc_equal, gridmetrics_of


@pytest.fixture(autouse=True)
def _close_mgpu_objects():
    """api.mgpu keeps one otmb_mgpu per device list for the life of the process; a test session that goes through a dozen lists
    would keep ~50 contexts (each with its pinned ring and copy threads) alive: close them after every test."""
    yield
    import sys

    api = sys.modules.get("otmb_amd.api") or sys.modules.get("oceantransportmatrixbuilder.jl_amd.api")
    if api is not None:
        for key in list(api._mgpu):
            api._mgpu.pop(key).close()


def _py_partition(level_counts, world):
    """The rule as rounds 1-3 had it in dist.py (floating-point target); the library restates it in exact integer arithmetic."""
    counts = np.asarray(level_counts, dtype=np.int64)
    nz = len(counts)
    total = int(counts.sum())
    cum = np.concatenate([[0], np.cumsum(counts)])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, nz)] - target):
            k -= 1
        k = max(k, bounds[-1] + 1)
        k = min(k, nz - (world - r))
        bounds.append(k)
    bounds.append(nz)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def test_balanced_partition_is_host_arithmetic_and_matches_the_rule():
    """No GPU: the library cuts levels into slabs (>= 1 level each, consecutive, covering) as the Python rule did."""
    from otmb_amd import capi, dist

    rng = np.random.default_rng(5)
    for trial in range(200):
        nz = int(rng.integers(1, 80))
        world = int(rng.integers(1, min(nz, 9) + 1))
        counts = np.sort(rng.integers(0, 100000, nz))[::-1] if trial % 2 else rng.integers(0, 50, nz)
        got = capi.balanced_partition(counts, world)
        assert got[0][0] == 0 and got[-1][1] == nz and all(a < b for a, b in got)
        assert all(got[r][1] == got[r + 1][0] for r in range(world - 1))
        if counts.sum() * world < 2 ** 52:  # where the floating-point target is exact enough to compare
            assert got == _py_partition(counts, world), (counts, world)
        assert dist.balanced_partition(counts, world) == got
    with pytest.raises(ValueError):
        capi.balanced_partition([1, 2], 3)


def _reference(oracle, g, gm, upwind=True, kappa=None):
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    kH, kML, kD = kappa if kappa is not None else (g.kappaH, g.kappaVML, g.kappaVdeep)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, kH, kML, kD, upwind)
    return ref, rphi, rtm


@pytest.mark.gpu
@pytest.mark.parametrize("ndev,rho,upwind", [(1, "array", True), (2, "array", True), (3, "scalar", False), (8, "array", True)])
def test_mgpu_on_one_gpu_matches_whole_grid_oracle(oracle, ndev, rho, upwind):
    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(24, 18, 11, seed=33 + ndev, rho=rho)
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm, upwind)
    devices = [0] * ndev
    idx = api.makeindices(gm.v3D)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
    for k in rphi:
        same = (phi[k] == rphi[k]) & (np.signbit(phi[k]) == np.signbit(rphi[k]))
        assert same.all(), (k, np.argwhere(~same)[:3])
    mg = api.mgpu(devices)
    assert mg.transport == "same-device copy"
    bounds = mg.partition()
    assert bounds[0] == 0 and bounds[-1] == 11 and len(bounds) == ndev + 1 and all(a < b for a, b in zip(bounds, bounds[1:]))
    for operators in (True, False):
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, upwind=upwind, devices=devices,
                                 operators=operators)
        for m in MATS if operators else MATS[:1]:
            assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/ndev={ndev}")
            assert tm[m].colptr[-1] == len(tm[m].rowval) + 1


@pytest.mark.gpu
def test_mgpu_compacts_T_across_slabs_when_entries_cancel(oracle):
    """κ = 0: every diffusive value is an explicit 0.0 -- kept in the operators, dropped from T (src/matrixbuilding.jl:147) -- so
    EVERY slab's T is shorter than its planned (union) bound and every slab below the first must shift its offsets."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(20, 16, 9, seed=41, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm, True, (0.0, 0.0, 0.0))
    idx = api.makeindices(gm.v3D)
    tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=0.0, κVML=0.0, κVdeep=0.0, devices=[0, 0, 0])
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], m)
    assert tm["T"].nnz == tm["Tadv"].nnz < tm["TκH"].nnz + tm["Tadv"].nnz


@pytest.mark.gpu
def test_mgpu_reports_the_reference_error_of_the_failing_slab(oracle):
    import otmb_amd.api as api
    from otmb_amd import synthetic
    from otmb_amd.capi import OtmbError

    g = synthetic.make_grid(24, 18, 11, seed=36, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    idx = api.makeindices(gm.v3D)
    devices = [0, 0, 0]
    kw = dict(mlotst=g.mlotst, gridmetrics=gm, indices=idx, devices=devices)
    # a NaN density in a cell of the deepest wet level: the LAST slab's check (src/matrixbuilding.jl:233)
    rho = g.rho.copy(order="F")
    rho.ravel(order="F")[ref["Lwet"][-1] - 1] = np.nan
    with pytest.raises(OtmbError, match="ρ contains NaNs") as e:
        api.transportmatrix(ϕ=rphi, ρ=rho, **kw)
    assert "slab 3 of 3" in str(e.value)
    # a NaN edge length (TκH, :61) and a NaN density in different slabs: the reference checks ρ first
    gm2 = dict(gm)
    gm2["edge_length_2D"] = {d: a.copy(order="F") for d, a in gm.edge_length_2D.items()}
    wet0 = ref["wet3D"].astype(bool)[:, :, 0]
    ii, jj = np.argwhere(wet0 & np.roll(wet0, 1, axis=0))[0]
    gm2["edge_length_2D"]["west"][ii, jj] = np.nan
    with pytest.raises(OtmbError, match="TκH contains NaNs."):
        api.transportmatrix(ϕ=rphi, ρ=g.rho, mlotst=g.mlotst, gridmetrics=gm2, indices=idx, devices=devices)
    with pytest.raises(OtmbError, match="ρ contains NaNs"):
        api.transportmatrix(ϕ=rphi, ρ=rho, mlotst=g.mlotst, gridmetrics=gm2, indices=idx, devices=devices)
    # facefluxes: a field without a single valid value is an assertion over the WHOLE grid (src/velocities.jl:199-200; on an
    # all-wet grid: next to land nofluxboundaries! has already written zeros, which count as values, as in the reference)
    allwet = dict(wet3D=np.ones((4, 3, 6), np.bool_))
    topo = dict(gridtopology=dict(kind=1))
    with pytest.raises(OtmbError, match="AssertionError"):
        api.facefluxes(np.full((4, 3, 6), np.nan), np.ones((4, 3, 6)), topo, allwet, FillValue=1e20, devices=devices)
    # ... and ONE slab with values is enough: u valid on the deepest level only (the last slab's)
    u1 = np.full((4, 3, 6), np.nan, order="F")
    u1[:, :, -1] = 2.0
    got1 = api.facefluxes(u1, np.ones((4, 3, 6)), topo, allwet, FillValue=1e20, devices=devices)
    want1 = oracle.facefluxes(u1, np.ones((4, 3, 6)), allwet["wet3D"], 1e20, 1)
    for k in want1:
        assert np.array_equal(got1[k], want1[k]), k
    # valid values in ONE slab only are enough
    u = np.full(g.umo.data.shape, 1e20, order="F")
    u[:, :, -1] = g.umo.data[:, :, -1]
    got = api.facefluxes(u, g.vmo.data, gm, idx, FillValue=1e20, devices=devices)
    want = oracle.facefluxes(u, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    # and the object is usable after every one of these errors
    tm = api.transportmatrix(ϕ=rphi, ρ=g.rho, **kw)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], m)
    # more devices than levels, a device twice among others
    with pytest.raises(OtmbError):
        api.transportmatrix(ϕ=rphi, ρ=g.rho, mlotst=g.mlotst, gridmetrics=gm, indices=idx, devices=[0] * 12)


@pytest.mark.gpu
def test_mgpu_facefluxes_on_grids_of_alternating_size(oracle):
    """One otmb_mgpu (the host layer keeps one per device list) used on a small and a larger grid in turn: the plane buffer of slab s is written
    by the thread of slab s + 1, so it has to exist at its final size before any slab thread runs (round 4: it was reserved inside its owner's
    thread, and on a grid larger than the previous one the slab below could copy into a buffer that was missing or being replaced -- an
    intermittent "plane hand-off" error).  Ten alternations, four slabs, every ϕ against the oracle."""
    import otmb_amd.api as api

    rng = np.random.default_rng(21)
    devices = [0, 0, 0, 0]
    topo = dict(gridtopology=dict(kind=1))
    for rep in range(10):
        nx, ny, nz = ((5, 4, 8), (40 + rep, 30, 9))[rep % 2]
        wet = rng.random((nx, ny, nz)) > 0.2
        wet[:, :, 0] |= True
        u = np.asfortranarray(np.where(wet, rng.standard_normal((nx, ny, nz)), 1e20))
        v = np.asfortranarray(np.where(wet, rng.standard_normal((nx, ny, nz)), 1e20))
        got = api.facefluxes(u, v, topo, dict(wet3D=wet), FillValue=1e20, devices=devices)
        want = oracle.facefluxes(u, v, wet, 1e20, 1)
        for k in want:
            assert np.array_equal(got[k], want[k]), (rep, k)


@pytest.mark.gpu
def test_mgpu_at_the_headline_grid_matches_the_single_device_path():
    """BASELINE.json configs[0/1]'s grid (360x300x50) cut into 4 slabs on GPU 0 against the single-context host path of the same
    library: identical bytes (the single-context path is compared with the oracle on this grid by tests/test_baseline_configs.py)."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    nx, ny, nz, lf = synthetic.PRESETS["access1deg"]
    g = synthetic.make_grid(nx, ny, nz, seed=20260501, land_fraction=lf, rho="array")
    gm = gridmetrics_of(g)
    idx = api.makeindices(gm.v3D)
    phi1 = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
    phi4 = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=[0, 0, 0, 0])
    for k in phi1:
        assert np.array_equal(phi1[k].view(np.int64), phi4[k].view(np.int64)), k
    tm1 = api.transportmatrix(ϕ=phi1, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho)
    tm4 = api.transportmatrix(ϕ=phi4, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=[0, 0, 0, 0])
    for m in MATS:
        assert_csc_equal(tuple(tm4[m]), tuple(tm1[m]), m)


@pytest.mark.gpu
def test_mgpu_over_rccl_one_gpu_per_slab(oracle):
    """Every slab on its own GPU: the chain's planes cross xGMI as grouped ncclSend / ncclRecv pairs on the single-process
    communicator.  Skipped on a one-GPU box."""
    import torch

    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs at least 2 GPUs ({n} visible)")
    import otmb_amd.api as api
    from otmb_amd import synthetic

    devices = list(range(min(n, 8)))
    g = synthetic.make_grid(24, 18, 12, seed=37, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    idx = api.makeindices(gm.v3D)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
    assert api.mgpu(devices).transport == "rccl"
    for k in rphi:
        assert np.array_equal(phi[k], rphi[k]), k
    tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], m)


@pytest.mark.gpu
def test_mgpu_reuse_flags_keep_grid_and_fluxes_on_the_devices(oracle):
    """otmb_mgpu_set_reuse: with reuse_fluxes the ϕ that otmb_mgpu_facefluxes left on the devices (owned levels + the one flux each halo
    level pushes into an owned cell) is used in place; with reuse_grid the grid constants are uploaded once.  Same matrices bit for bit,
    fewer bytes up -- and nothing stale: other ϕ arrays, or modified host ϕ WITHOUT the promise, are uploaded."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(28, 20, 13, seed=45, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    devices = [0, 0, 0, 0]
    idx = api.makeindices(gm.v3D)
    mg = api.mgpu(devices)

    def run(fresh_phi=True, **kw):
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices) if fresh_phi else run.phi
        run.phi = phi
        b0 = mg.uploaded_bytes()
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices, **kw)
        for m in MATS:
            assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/{kw}")
        return mg.uploaded_bytes() - b0

    full = run()
    assert run(reuse_fluxes=True) < full                      # six ϕ arrays (with their halo levels) stay where they are
    first = run(reuse_grid=True)
    again = run(reuse_grid=True)
    assert again < first == full                              # the first call with the promise still uploads the grid
    both = run(reuse_grid=True, reuse_fluxes=True)
    assert both < again and both < run(reuse_fluxes=True)
    nx, ny, nz = gm.v3D.shape
    P = nx * ny
    nze = nz + 2 * (len(devices) - 1)                        # every inner boundary adds a halo level on both sides
    assert both == nze * P * 8 + len(devices) * P * 8         # ρ (extended levels) and mlotst per slab: what changes between time slices
    # the promise is about THESE host arrays: ϕ from elsewhere (the oracle's, equal in value) is uploaded
    run.phi = rphi
    assert run(fresh_phi=False, reuse_fluxes=True) > both
    # no promise, modified host ϕ: the modification is what is assembled
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
    z = {k: np.zeros_like(v) for k, v in phi.items()}
    tm0 = api.transportmatrix(ϕ=z, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices)
    assert tm0["Tadv"].nnz == 0 and tm0["TκH"].nnz == rtm["TκH"][1].size


@pytest.mark.gpu
@pytest.mark.parametrize("ndev", [2, 3, 8])
def test_mgpu_facefluxes_chain_in_row_bands(oracle, ndev):
    """the chain over depth slabs is handed over piece by piece (row bands), slab s piece c waiting for slab s + 1 piece c
    only.  Any number of pieces -- 1, 2, 5 (bands of unequal height), one row per piece -- gives the oracle's six arrays bit for bit,
    float32 inputs included; a hand-off that fails in the middle of the chain releases every slab (no deadlock), is reported, and
    leaves the object usable."""
    import os

    import otmb_amd.api as api
    from otmb_amd import synthetic
    from otmb_amd.capi import OtmbError

    g = synthetic.make_grid(28, 20, 13, seed=47, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, _ = _reference(oracle, g, gm)
    devices = [0] * ndev
    idx = api.makeindices(gm.v3D)
    mg = api.mgpu(devices)
    try:
        for pieces in (1, 2, 5, 20):
            mg.set_chain_pieces(pieces)
            phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
            for k in rphi:
                assert np.array_equal(phi[k], rphi[k]) and np.array_equal(np.signbit(phi[k]), np.signbit(rphi[k])), (k, pieces)
        mg.set_chain_pieces(5)
        os.environ["OTMB_TEST_FAIL_PIECE"] = f"{ndev - 1}:2"  # the deepest slab's third piece never leaves
        try:
            with pytest.raises(OtmbError) as e:
                api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
            assert "injected" in str(e.value)
        finally:
            del os.environ["OTMB_TEST_FAIL_PIECE"]
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
        for k in rphi:
            assert np.array_equal(phi[k], rphi[k]), k
    finally:
        mg.set_chain_pieces(0)


@pytest.mark.gpu
def test_facefluxes_row_bands_on_one_device_equal_the_whole_plane(oracle):
    """otmb_facefluxes_rows_dev: the plane in row bands (any cut, one-row and four-row workgroups) -- the same six arrays and the same
    validity flags as one call over the whole plane."""
    import ctypes as C

    import torch

    from otmb_amd import capi, synthetic

    for rows_env, shape in (("1", (40, 23, 7)), ("4", (150, 23, 7))):
        os_env = {"OTMB_FF_ROWS": rows_env}
        import os

        os.environ.update(os_env)
        try:
            ctx = capi.Context(0)
        finally:
            del os.environ["OTMB_FF_ROWS"]
        lib = capi.lib()
        g = synthetic.make_grid(*shape, seed=48, rho="array")
        gm = gridmetrics_of(g)
        idx = oracle.makeindices(gm.v3D)
        rphi = oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], 1e20, gm.gridtopology.kind)
        nx, ny, nz = shape
        G = nx * ny * nz
        flat = lambda a, dt=np.float64: torch.from_numpy(np.asfortranarray(a, dtype=dt).ravel(order="F")).cuda()
        umo, vmo, wet = flat(g.umo.data), flat(g.vmo.data), flat(idx["wet3D"], np.uint8)
        for cuts in ([0, ny], [0, 1, 2, ny], [0, 7, 8, 19, ny], list(range(ny + 1))):
            phi = [torch.full((G,), np.nan, dtype=torch.float64, device="cuda") for _ in range(6)]
            ptrs = capi.ptr_array(6, [p.data_ptr() for p in phi])
            for c in range(len(cuts) - 1):
                ctx.check(lib.otmb_facefluxes_rows_dev(ctx.handle, umo.data_ptr(), vmo.data_ptr(), 0, wet.data_ptr(), 1e20, nx, ny, nz,
                                                       gm.gridtopology.kind, C.byref(ptrs), None, None, cuts[c], cuts[c + 1], int(c == 0)))
            u, v = C.c_int32(0), C.c_int32(0)
            ctx.check(lib.otmb_facefluxes_slab_flags(ctx.handle, C.byref(u), C.byref(v)))
            assert u.value == 1 and v.value == 1
            for k, name in enumerate(capi.PHI_ORDER):
                got = phi[k].cpu().numpy().reshape(shape, order="F")
                assert np.array_equal(got, rphi[name]), (name, cuts, rows_env)
        ctx.close()


@pytest.mark.gpu
def test_mgpu_forgets_uploads_that_never_reached_the_devices(oracle):
    """a slab's residency keys are written when an upload is QUEUED.  A plan whose transfer fails must take them back, or a
    retry on the same handle with reuse_grid on would treat arrays that were never copied as resident and build matrices from
    uninitialised device memory (and skip the local-index shift of Lwet)."""
    import os

    import otmb_amd.api as api
    from otmb_amd import synthetic
    from otmb_amd.capi import OtmbError

    g = synthetic.make_grid(28, 20, 13, seed=46, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    devices = [0, 0, 0]
    idx = api.makeindices(gm.v3D)
    mg = api.mgpu(devices)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
    os.environ["OTMB_TEST_FAIL_UPLOAD"] = "1"
    try:
        with pytest.raises(OtmbError) as e:
            api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices, reuse_grid=True)
        assert "injected" in str(e.value)
    finally:
        del os.environ["OTMB_TEST_FAIL_UPLOAD"]
    b0 = mg.uploaded_bytes()
    tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices, reuse_grid=True)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], m)
    first = mg.uploaded_bytes() - b0
    b0 = mg.uploaded_bytes()
    tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices, reuse_grid=True)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], m)
    assert mg.uploaded_bytes() - b0 < first  # the retry uploaded the grid in full; only now is it resident


@pytest.mark.gpu
def test_mgpu_with_precomputed_operators(oracle):
    """transportmatrix(…; Tadv = …, devices = …) (src/matrixbuilding.jl:133-147): ignore_ops travels to every slab -- a NaN in ρ or a flux
    into land is no error when Tadv is handed in -- and T is the three sparse adds of the given and the built operators."""
    import otmb_amd.api as api
    from otmb_amd import synthetic
    from otmb_amd.capi import OtmbError

    g = synthetic.make_grid(20, 16, 9, seed=47, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    idx = api.makeindices(gm.v3D)
    devices = [0, 0, 0]
    kw = dict(mlotst=g.mlotst, gridmetrics=gm, indices=idx, devices=devices)
    base = api.transportmatrix(ϕ=rphi, ρ=g.rho, **kw)
    rho_nan = g.rho.copy(order="F")
    rho_nan.ravel(order="F")[ref["Lwet"][-1] - 1] = np.nan
    bad_phi = {k: v.copy(order="F") for k, v in rphi.items()}
    wet = ref["wet3D"].astype(bool)
    i, j, k = np.argwhere(wet & ~np.roll(wet, 1, axis=0))[-1]
    bad_phi["west"][i, j, k] = 5.0
    with_adv = api.transportmatrix(ϕ=bad_phi, ρ=rho_nan, Tadv=base.Tadv, **kw)
    assert with_adv.Tadv is base.Tadv
    for m in MATS:
        assert_csc_equal(tuple(with_adv[m]), rtm[m], m)
    with pytest.raises(OtmbError, match="ρ contains NaNs"):
        api.transportmatrix(ϕ=rphi, ρ=rho_nan, TκH=base.TκH, **kw)
    with pytest.raises(OtmbError, match="flux into a land cell"):
        api.transportmatrix(ϕ=bad_phi, ρ=g.rho, TκH=base.TκH, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    (12, 10, 6, 3, 101, "array", "bipolar", np.float64),
    (7, 5, 4, 4, 102, "scalar", "tripolar", np.float64),     # odd nx on the seam, as many slabs as levels
    (36, 30, 10, 5, 103, "array", "tripolar", np.float32),   # Float32 mass transports (the CMIP on-disk type)
    (2, 3, 3, 2, 104, "array", "tripolar", np.float64),      # nx = 2: every cell takes the generic column builder
    (64, 8, 13, 6, 105, "scalar", "bipolar", np.float32),
], ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}-{c[3]}slabs-{c[5]}-{c[6]}-{np.dtype(c[7]).name}")
def test_mgpu_sweep_of_grids_topologies_and_input_types(oracle, case):
    import otmb_amd.api as api
    from helpers import randomize_metrics
    from otmb_amd import synthetic

    nx, ny, nz, ndev, seed, rho, topo, dt = case
    g = synthetic.make_grid(nx, ny, nz, seed=seed, rho=rho, topology=topo, dtype_flux=dt, land_fraction=0.0 if nx == 2 else 0.3)
    gm = gridmetrics_of(g)
    if nx % 2 == 1 or nx == 2:
        randomize_metrics(gm)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], g.umo.properties["_FillValue"], gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    devices = [0] * ndev
    idx = api.makeindices(gm.v3D)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
    for k in rphi:
        same = (phi[k] == rphi[k]) & (np.signbit(phi[k]) == np.signbit(rphi[k]))
        assert same.all(), k
    for reuse in (False, True):
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices, reuse_grid=reuse, reuse_fluxes=reuse)
        for m in MATS:
            assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/reuse={reuse}")


@pytest.mark.gpu
def test_rccl_transport_loads_and_initialises_on_one_gpu(oracle, monkeypatch):
    """The RCCL transport cannot carry a plane on a one-GPU box, but its loading path can run: OTMB_MGPU_TRANSPORT=rccl makes a
    one-device otmb_mgpu dlopen librccl.so, resolve its seven entry points and bring up (and later destroy) a one-rank communicator."""
    import otmb_amd.api as api
    from otmb_amd import capi, synthetic

    monkeypatch.setenv("OTMB_MGPU_TRANSPORT", "rccl")
    mg = capi.Mgpu([0])
    try:
        assert mg.transport == "rccl", capi.lib().otmb_mgpu_last_error(mg.handle)
        g = synthetic.make_grid(12, 10, 6, seed=3, rho="array")
        gm = gridmetrics_of(g)
        ref, rphi, rtm = _reference(oracle, g, gm)
        api._mgpu[(0,)] = mg
        idx = api.makeindices(gm.v3D)
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=[0])
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=[0])
        for m in MATS:
            assert_csc_equal(tuple(tm[m]), rtm[m], m)
    finally:
        api._mgpu.pop((0,), None)
        mg.close()


# ---- the pipelined one-phase build (otmb_mgpu_transportmatrix_onepass; api.transportmatrix(..., slabs=S)) ----------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("slabs,rho,upwind", [(1, "array", True), (2, "array", True), (4, "scalar", False), (8, "array", True)])
def test_onepass_on_one_gpu_matches_whole_grid_oracle(oracle, slabs, rho, upwind):
    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(24, 18, 11, seed=53 + slabs, rho=rho)
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm, upwind)
    idx = api.makeindices(gm.v3D)
    for operators in (True, False):
        for rep in range(2):  # twice: the slabs' buffers and the pinned pool are reused
            tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, upwind=upwind, slabs=slabs, operators=operators)
            for m in MATS if operators else MATS[:1]:
                assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/slabs={slabs}/rep={rep}")
                assert tm[m].colptr[-1] == len(tm[m].rowval) + 1
            if not operators:
                assert all(tm[m] is None for m in MATS[1:])


@pytest.mark.gpu
def test_onepass_places_every_slab_behind_the_final_counts_of_the_slabs_above(oracle):
    """κ = 0: every slab's T loses entries (exact-zero sums, src/matrixbuilding.jl:147) -- a slab's offsets are the running sums of the
    FINAL counts above it, nothing is re-based afterwards."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(20, 16, 9, seed=41, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm, True, (0.0, 0.0, 0.0))
    idx = api.makeindices(gm.v3D)
    tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=0.0, κVML=0.0, κVdeep=0.0, slabs=3)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], m)
    assert tm["T"].nnz == tm["Tadv"].nnz < tm["TκH"].nnz + tm["Tadv"].nnz


@pytest.mark.gpu
def test_onepass_errors_release_every_slab(oracle):
    """A failing slab in the middle of the pipeline: the slabs below it store nothing and return, the reference's error comes back with
    the slab named, and the object builds the right matrices afterwards.  Too small an output array: OTMB_ERR_CAPACITY."""
    import otmb_amd.api as api
    from otmb_amd import capi, synthetic
    from otmb_amd.capi import OtmbError

    g = synthetic.make_grid(24, 18, 11, seed=36, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    idx = api.makeindices(gm.v3D)
    kw = dict(mlotst=g.mlotst, gridmetrics=gm, indices=idx, slabs=4)
    mg = api.mgpu([0, 0, 0, 0])
    for where in (0, len(ref["Lwet"]) // 2, -1):  # first, a middle and the last slab
        rho = g.rho.copy(order="F")
        rho.ravel(order="F")[ref["Lwet"][where] - 1] = np.nan
        with pytest.raises(OtmbError, match="ρ contains NaNs") as e:
            api.transportmatrix(ϕ=rphi, ρ=rho, **kw)
        assert "of 4" in str(e.value)
        tm = api.transportmatrix(ϕ=rphi, ρ=g.rho, **kw)
        for m in MATS:
            assert_csc_equal(tuple(tm[m]), rtm[m], m)
    # capacity: the C call with arrays one entry short for TκH
    keep, passthrough = [], []
    a = api._tm_args(rphi, g.mlotst, gm, idx, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, True, keep, passthrough)
    N = int(idx["N"])
    want = [len(rtm[m][1]) for m in MATS]
    cap = [N * 7 + 1, N * 7 + 1, want[2] - 1, N * 3 + 1, N * 3 + 1]
    cp = [np.zeros(N + 1, np.int64) for _ in range(5)]
    rv = [np.zeros(c + 1, np.int64) for c in cap]   # (one entry more than the first call is told: the second call gets exactly enough)
    nz = [np.zeros(c + 1, np.float64) for c in cap]
    final = (C.c_int64 * 5)()
    rc = capi.lib().otmb_mgpu_transportmatrix_onepass(mg.handle, C.byref(a), C.byref(capi.ptr_array(5, [x.ctypes.data for x in cp])),
                                                      C.byref(capi.ptr_array(5, [x.ctypes.data for x in rv])),
                                                      C.byref(capi.ptr_array(5, [x.ctypes.data for x in nz])), C.byref((C.c_int64 * 5)(*cap)), C.byref(final))
    assert rc == 14, rc  # OTMB_ERR_CAPACITY
    cap[2] = want[2]  # exactly enough
    rc = capi.lib().otmb_mgpu_transportmatrix_onepass(mg.handle, C.byref(a), C.byref(capi.ptr_array(5, [x.ctypes.data for x in cp])),
                                                      C.byref(capi.ptr_array(5, [x.ctypes.data for x in rv])),
                                                      C.byref(capi.ptr_array(5, [x.ctypes.data for x in nz])), C.byref((C.c_int64 * 5)(*cap)), C.byref(final))
    assert rc == capi.OK and list(final) == want, (rc, list(final), want)
    for q, m in enumerate(MATS):
        assert_csc_equal((cp[q], rv[q][: want[q]], nz[q][: want[q]]), rtm[m], m)


@pytest.mark.gpu
def test_onepass_with_the_reuse_promises_uploads_only_what_changed(oracle):
    """A time loop (facefluxes, then transportmatrix) on three slabs of one device with both promises: a slice uploads ρ and mlotst only."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(24, 18, 11, seed=37, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    idx = api.makeindices(gm.v3D)
    devices = [0, 0, 0]
    mg = api.mgpu(devices)

    def time_slice(**kw):
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, devices=devices)
        b0 = mg.uploaded_bytes()
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, devices=devices, slabs=3, **kw)
        for m in MATS:
            assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/{kw}")
        return mg.uploaded_bytes() - b0

    full = time_slice()
    first = time_slice(reuse_grid=True, reuse_fluxes=True)   # the first call with the promise still uploads the grid
    both = time_slice(reuse_grid=True, reuse_fluxes=True)
    nx, ny, nz = gm.v3D.shape
    P = nx * ny
    nze = nz + 2 * (len(devices) - 1)
    assert both == nze * P * 8 + len(devices) * P * 8 < first < full, (both, first, full)


@pytest.mark.gpu
def test_onepass_into_ordinary_host_memory_and_the_default_rule(oracle, monkeypatch):
    """pinned results off: the upper-bound arrays are pageable and every slab's columns come home through its context's staging ring.
    default_slabs: 4 from 2^18 wet cells and 8 levels on, never with reuse_fluxes, a device list or beyond 2^25 wet cells."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    monkeypatch.delenv("OTMB_HOST_SLABS", raising=False)
    assert api.default_slabs(3_000_000, 50, False, None) == 4 and api.default_slabs(3_000_000, 50, True, None) == 0
    assert api.default_slabs(3_000_000, 50, False, [0, 1]) == 0 and api.default_slabs(100_000, 50, False, None) == 0
    assert api.default_slabs(3_000_000, 7, False, None) == 0 and api.default_slabs(63_000_000, 75, False, None) == 4  # (round 6: no upper limit)
    monkeypatch.setenv("OTMB_HOST_SLABS", "0")
    assert api.default_slabs(3_000_000, 50, False, None) == 0
    monkeypatch.setenv("OTMB_HOST_SLABS", "6")
    assert api.default_slabs(3_000_000, 50, False, None) == 6
    monkeypatch.setattr(api, "PINNED_OUTPUTS", False)
    g = synthetic.make_grid(70, 40, 12, seed=58, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    idx = api.makeindices(gm.v3D)
    tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, slabs=3)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], m)


@pytest.mark.gpu
def test_onepass_result_arrays_follow_the_previous_slice_and_recover_when_it_is_outgrown(oracle):
    """The pipelined call's result arrays: the wet mask's bounds on the first slice (otmb_static_capacity), then the previous slice's counts
    + 25 / 50 % for Tadv / TκVML -- fewer pinned bytes, the same matrices -- and a slice that outgrows them (OTMB_ERR_CAPACITY inside) is
    built again at the mask's bounds without the caller noticing."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(70, 40, 12, seed=59, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    idx = api.makeindices(gm.v3D)
    kw = dict(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, slabs=3)
    key, bound = api._capacity_bounds(idx, gm)
    nnz = [len(rtm[m][1]) for m in MATS]
    assert all(b >= n for b, n in zip(bound, nnz)) and bound[4] == nnz[4] and sum(bound) < 25 * ref["N"]
    api._prev_nnz.pop(key, None)
    sizes = []
    for call in range(3):
        tm = api.transportmatrix(**kw)
        sizes.append((api.last_call_seconds["result_bytes_pinned"], api.last_call_seconds["result_bytes_used"]))
        for m in MATS:
            assert_csc_equal(tuple(tm[m]), rtm[m], f"call {call}: {m}")
    assert api._prev_nnz[key] == nnz
    assert sizes[1][0] < sizes[0][0] and sizes[1] == sizes[2] and sizes[1][0] >= sizes[1][1] and sizes[1][0] < 1.15 * sizes[1][1] + 5 * 16 * 4097, sizes
    # the previous slice is outgrown: pretend it had a tenth of the advective entries
    api._prev_nnz[key] = [nnz[0], max(1, (nnz[1] - 8000) // 10), nnz[2], nnz[3], nnz[4]]
    if api._capacities(key, bound)[1] < nnz[1]:
        tm = api.transportmatrix(**kw)
        for m in MATS:
            assert_csc_equal(tuple(tm[m]), rtm[m], f"after the retry: {m}")
        assert api._prev_nnz[key] == nnz


@pytest.mark.gpu
def test_onepass_at_the_headline_grid_matches_the_two_phase_path():
    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(360, 300, 50, seed=20260501, rho="array")
    gm = gridmetrics_of(g)
    idx = api.makeindices(gm.v3D)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
    kw = dict(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH, κVML=g.kappaVML, κVdeep=g.kappaVdeep)
    ref = api.transportmatrix(**kw)
    for slabs in (4, 6):
        tm = api.transportmatrix(slabs=slabs, **kw)
        for m in MATS:
            for a, b, what in zip(tuple(tm[m]), tuple(ref[m]), ("colptr", "rowval", "nzval")):
                assert np.array_equal(a, b), (m, what, slabs)


@pytest.mark.gpu
def test_index_arrays_cross_the_link_as_int32_and_come_back_the_same(oracle, monkeypatch):
    """otmb_xfer's `narrow` items (row indices and column offsets travel as Int32, the host threads widen them): forced onto every array
    (OTMB_XFER_NARROW_MIN_KB=1) with 1 MiB ring pieces, so that a matrix takes more pieces than the ring has slots -- the two-phase call on a
    fresh context, otmb_mgpu plan / fetch and the pipelined one-phase build, pinned and ordinary result arrays -- against the oracle and
    against the same calls with the arrays travelling as they are."""
    import otmb_amd.api as api
    from otmb_amd import capi, synthetic

    g = synthetic.make_grid(130, 90, 30, seed=61, rho="array")
    gm = gridmetrics_of(g)
    ref, rphi, rtm = _reference(oracle, g, gm)
    idx = api.makeindices(gm.v3D)
    assert len(rtm["T"][1]) * 4 > 5 * (1 << 20)  # T's row indices as Int32: more than four pieces of 1 MiB
    kw = dict(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho)
    monkeypatch.setenv("OTMB_XFER_CHUNK_MB", "1")
    for narrow in ("1", "0"):
        monkeypatch.setenv("OTMB_XFER_NARROW", narrow)
        monkeypatch.setenv("OTMB_XFER_NARROW_MIN_KB", "1")
        for pinned in (True, False):
            monkeypatch.setattr(api, "PINNED_OUTPUTS", pinned)
            old = api._ctx.pop(0, None)  # a fresh single-GPU context reads the switches
            api._ctx[0] = capi.Context(0)
            try:
                for name, extra in (("two-phase", dict(slabs=0)), ("mgpu", dict(devices=[0, 0, 0])), ("onepass", dict(slabs=3))):
                    tm = api.transportmatrix(**kw, **extra)
                    for m in MATS:
                        assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/{name}/narrow={narrow}/pinned={pinned}")
                    del tm
            finally:
                api._ctx.pop(0).close()
                if old is not None:
                    api._ctx[0] = old
            for key in list(api._mgpu):
                api._mgpu.pop(key).close()


def test_the_default_protocol_is_chosen_by_measurement():
    """api.Trial (no GPU): calls 1-4 pipelined (3 and 4 timed), 5-7 two-phase (6 and 7 timed), then the faster one by the minimum of its
    two samples; a trial whose timed calls raised stays with the pipelined build; every 64th call re-measures the loser; a protocol that
    turns slow three calls in a row starts the trial over; one trial per kind of call."""
    import otmb_amd.api as api

    for t_pipe, t_two, want in ((0.020, 0.025, True), (0.028, 0.024, False)):
        tr, seq = api.Trial(), []
        for call in range(12):
            p = tr.pipelined()
            seq.append(p)
            tr.record(t_pipe if p else t_two)
        assert seq[:7] == [True] * 4 + [False] * 3 and seq[7:] == [want] * 5, seq
    tr = api.Trial()
    for call in range(10):
        p = tr.pipelined()
        if call not in (5, 6):  # the timed two-phase calls raised: nothing recorded
            tr.record(0.02)
    assert tr.pipelined() is True
    # an outlier in one of the two samples does not decide: the minimum does
    tr = api.Trial()
    for call, t in enumerate((0.1, 0.03, 0.020, 0.031, 0.05, 0.025, 0.026)):
        tr.pipelined()
        tr.record(t)
    assert tr.t == {True: 0.020, False: 0.025} and tr.pipelined() is True
    # the verdict is revisited: call 64 runs the loser and refreshes its time -- which can change the choice
    tr = api.Trial()
    seq = []
    for call in range(1, 70):
        p = tr.pipelined()
        seq.append(p)
        tr.record((0.020 if p else 0.025) if call < 64 else (0.030 if p else 0.018))
    assert seq[7:63] == [True] * 56 and seq[63] is False and seq[64:] == [False] * 5, seq[60:]
    # a host that got busy: three slow calls in a row start the trial over (both protocols timed again, no warm-up calls)
    tr = api.Trial()
    for call in range(10):
        p = tr.pipelined()
        tr.record(0.020 if p else 0.025)
    for _ in range(3):
        assert tr.pipelined() is True
        tr.record(0.040)
    assert not tr.decided()
    seq = []
    for call in range(6):
        p = tr.pipelined()
        seq.append(p)
        tr.record(0.040 if p else 0.030)
    assert seq == [True, True, False, False, False, False], seq
    # one trial per kind of call
    T = api.Trial
    assert T.of(3, 12345) is T.of(3, 12345) and T.of(3, 12345) is not T.of(3, 12346)
    assert T.of(3, 12345) is not T.of(3, 12345, operators=False) and T.of(3, 12345) is not T.of(3, 12345, rho3d=True)
    assert T.of(3, 12345) is not T.of(3, 12345, reuse_grid=True) and T.of(3, 12345) is not T.of(3, 12345, given={"TκH": object()})
    assert T.peek(3, 999) is None and T.peek(3, 12345) is T.of(3, 12345)


def test_reuse_grid_is_not_forwarded_to_an_engine_that_did_not_serve_the_previous_call():
    """reuse_grid is the caller's promise about THE PREVIOUS CALL; an engine (single context / otmb_mgpu) checks it against its
    OWN previous call, so after a change of engine the promise stays behind (no GPU: the bookkeeping only)."""
    import otmb_amd.api as api

    api._last_engine.pop(7, None)
    assert api._reuse_grid_for(7, (7, 7, 7, 7), True) is False   # first call on the device: nothing can be resident
    assert api._reuse_grid_for(7, (7, 7, 7, 7), True) is True
    assert api._reuse_grid_for(7, "ctx", True) is False          # the Trial switched engines: the context's keys are from long ago
    assert api._reuse_grid_for(7, "ctx", True) is True
    assert api._reuse_grid_for(7, "ctx", False) is False
    assert api._reuse_grid_for(7, (7, 7, 7, 7), True) is False   # ... and back
    assert api._reuse_grid_for(7, (7, 7), True) is False         # another cut is another engine


@pytest.mark.gpu
def test_default_calls_at_the_headline_grid_switch_protocols_and_stay_right(monkeypatch):
    """Ten default calls on the 1 degree grid go through both protocols (api.Trial); every one returns the two-phase call's matrices."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    monkeypatch.delenv("OTMB_HOST_SLABS", raising=False)
    g = synthetic.make_grid(360, 300, 50, seed=20260501, rho="array")
    gm = gridmetrics_of(g)
    idx = api.makeindices(gm.v3D)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
    kw = dict(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH, κVML=g.kappaVML, κVdeep=g.kappaVdeep)
    ref = api.transportmatrix(slabs=0, **kw)
    api.Trial._all.pop(api.Trial.key(0, int(idx["N"]), rho3d=True), None)
    used = []
    for call in range(10):
        tm = api.transportmatrix(**kw)
        used.append(api.Trial.of(0, int(idx["N"]), rho3d=True).now)
        for m in MATS:
            for a, b, what in zip(tuple(tm[m]), tuple(ref[m]), ("colptr", "rowval", "nzval")):
                assert np.array_equal(a, b), (m, what, call)
        del tm
    assert used[:7] == [True] * 4 + [False] * 3 and used[7] == used[8] == used[9], used
