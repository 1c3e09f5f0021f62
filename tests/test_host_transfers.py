"""The host-pointer entry points move their arrays through the library's own staging (csrc/otmb_xfer.hip: pinned ring,
parallel host copies): results must be what they were, for arrays smaller than a chunk, spanning many chunks and with
odd sizes, and the reuse_grid option must not change results (run with -m gpu)."""
import numpy as np
import pytest

from helpers import MATS, assert_csc_equal, gridmetrics_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,chunk_mb", [((36, 30, 11), 8), ((90, 80, 20), 1), ((120, 100, 23), 1)])
def test_host_api_through_the_transfer_engine_matches_oracle(oracle, monkeypatch, shape, chunk_mb):
    import otmb_amd.api as api
    from otmb_amd import capi, synthetic

    monkeypatch.setenv("OTMB_XFER_CHUNK_MB", str(chunk_mb))  # read when a context creates its ring
    g = synthetic.make_grid(*shape, seed=51, rho="array")
    gm = gridmetrics_of(g)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    api._ctx.pop(7, None)
    api._ctx[7] = capi.Context(0)  # a fresh context (device key 7 is only a dictionary key here) with this chunk size
    try:
        ctx_kw = dict(device=7)
        idx = api.makeindices(gm.v3D, **ctx_kw)
        assert np.array_equal(idx.Lwet3D, ref["Lwet3D"]) and np.array_equal(idx.Lwet, ref["Lwet"])
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, **ctx_kw)
        for k in rphi:
            assert np.array_equal(phi[k], rphi[k]), k
        for reuse in (False, True, True, False):
            tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, reuse_grid=reuse, **ctx_kw)
            for m in MATS:
                assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/reuse={reuse}")
        # reuse_grid with ANOTHER grid behind other arrays: nothing stale may be used
        g2 = synthetic.make_grid(*shape, seed=52, rho="array")
        gm2 = gridmetrics_of(g2)
        ref2 = oracle.makeindices(gm2.v3D)
        rphi2 = oracle.facefluxes(g2.umo.data, g2.vmo.data, ref2["wet3D"], 1e20, gm2.gridtopology.kind)
        rtm2 = oracle.transportmatrix(rphi2, gm2, ref2, g2.rho, g2.mlotst, g2.kappaH, g2.kappaVML, g2.kappaVdeep, True)
        idx2 = api.makeindices(gm2.v3D, **ctx_kw)
        tm2 = api.transportmatrix(ϕ=rphi2, mlotst=g2.mlotst, gridmetrics=gm2, indices=idx2, ρ=g2.rho, reuse_grid=True, **ctx_kw)
        for m in MATS:
            assert_csc_equal(tuple(tm2[m]), rtm2[m], f"{m}/second grid")
    finally:
        api._ctx.pop(7).close()


def test_plan_is_invalidated_by_other_host_calls(oracle):
    """plan (host) -> makeindices of ANOTHER grid (host; reuses the staging slots) -> fetch must refuse, not fill garbage."""
    import ctypes as C

    import otmb_amd.api as api
    from otmb_amd import capi, synthetic
    from otmb_amd.capi import OtmbError

    g = synthetic.make_grid(20, 16, 6, seed=53, rho="array")
    gm = gridmetrics_of(g)
    idx = api.makeindices(gm.v3D)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
    ctx = api.context(0)
    keep = []
    a = api._tm_args(phi, g.mlotst, gm, idx, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, True, keep)
    nnz = (C.c_int64 * 5)()
    ctx.check(capi.lib().otmb_transportmatrix_plan(ctx.handle, C.byref(a), C.byref(nnz)))
    other = synthetic.make_grid(20, 16, 6, seed=54)
    api.makeindices(gridmetrics_of(other).v3D)  # overwrites the staged v3D / Lwet3D the plan pointed at
    N = int(idx["N"])
    colptr = [np.empty(N + 1, np.int64) for _ in range(5)]
    rowval = [np.empty(int(nnz[m]), np.int64) for m in range(5)]
    nzval = [np.empty(int(nnz[m]), np.float64) for m in range(5)]
    cp = capi.ptr_array(5, [x.ctypes.data for x in colptr])
    rv = capi.ptr_array(5, [x.ctypes.data for x in rowval])
    nz = capi.ptr_array(5, [x.ctypes.data for x in nzval])
    final = (C.c_int64 * 5)()
    with pytest.raises(OtmbError) as e:
        ctx.check(capi.lib().otmb_transportmatrix_fetch(ctx.handle, C.byref(cp), C.byref(rv), C.byref(nz), C.byref(final)))
    assert e.value.name == "NO_PLAN"


def test_facefluxes_never_trusts_the_address_of_a_wet_mask(oracle):
    """reuse_grid is sticky on the context (set by the last transportmatrix(...; reuse_grid=true)), facefluxes makes no
    promise about its wet mask, and callers hand over converted temporaries that an allocator may place at the address of
    the previous one: a mask at the SAME host address with OTHER content must be uploaded again."""
    import otmb_amd.api as api
    from otmb_amd import synthetic

    shape = (40, 32, 9)
    g = synthetic.make_grid(*shape, seed=61, rho="array")
    gm = gridmetrics_of(g)
    idx = api.makeindices(gm.v3D)
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
    api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, reuse_grid=True)  # the flag is on now
    wet = np.asfortranarray(np.array(idx.wet3D, copy=True))
    api.facefluxes(g.umo.data, g.vmo.data, gm, dict(wet3D=wet), FillValue=1e20)  # this address is what the slot remembers
    g2 = synthetic.make_grid(*shape, seed=62, rho="array")
    gm2 = gridmetrics_of(g2)
    ref2 = oracle.makeindices(gm2.v3D)
    assert not np.array_equal(ref2["wet3D"].astype(bool), wet)
    wet[...] = ref2["wet3D"].astype(bool)  # another grid's mask, same array object, same address
    got = api.facefluxes(g2.umo.data, g2.vmo.data, gm2, dict(wet3D=wet), FillValue=1e20)
    want = oracle.facefluxes(g2.umo.data, g2.vmo.data, ref2["wet3D"], 1e20, gm2.gridtopology.kind)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    api.context(0).set_reuse_grid(False)


def test_reuse_fluxes_and_pinned_outputs(oracle):
    """reuse_fluxes: the ϕ arrays facefluxes returned are not uploaded again -- same matrices; ϕ from somewhere else (other
    addresses) or modified-and-not-promised ϕ is uploaded as ever.  The results live in pinned memory of the context: views
    keep their block alive, dropped results return it to the pool."""
    import gc

    import otmb_amd.api as api
    from otmb_amd import synthetic

    g = synthetic.make_grid(90, 80, 20, seed=71, rho="array")
    gm = gridmetrics_of(g)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    idx = api.makeindices(gm.v3D)
    ctx = api.context(0)
    for rnd in range(3):
        phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
        for k in rphi:
            assert np.array_equal(phi[k], rphi[k]), k
        tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, reuse_fluxes=True, reuse_grid=rnd > 0)
        for m in MATS:
            assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/round {rnd}")
        keep = tm["T"].nzval[:5].copy()
        view = tm["T"].nzval[:5]  # a view keeps the pinned block alive after the matrices are dropped
        del tm, phi
        gc.collect()
        assert np.array_equal(view, keep)
    # the promise is per call: ϕ at other addresses (the oracle's arrays) with reuse_fluxes on is simply uploaded
    phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
    tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, reuse_fluxes=True)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), rtm[m], m)
    # ... and without the promise modified fluxes are what counts
    phi2 = {k: np.array(v, copy=True, order="F") for k, v in phi.items()}
    phi2["east"] *= 0.5
    phi2["west"] *= 0.5
    want = oracle.transportmatrix(phi2, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    phi["east"] *= 0.5  # in place: same addresses as what facefluxes returned
    phi["west"] *= 0.5
    tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), want[m], m)
    # pinned blocks are recycled: a dropped result's block serves the next request of its size
    a = ctx.pinned_empty(1 << 20, np.float64)
    addr = a.ctypes.data
    del a
    gc.collect()
    b = ctx.pinned_empty(1 << 20, np.float64)
    assert b.ctypes.data == addr
