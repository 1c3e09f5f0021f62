"""Pinned host memory handed to callers (otmb_host_alloc / otmb_host_free): lifetimes and threads as the Julia shim needs them
 -- a block outlives the context that allocated it, otmb_host_free works after otmb_ctx_destroy and from any
thread while a call is in flight on the context; and the two reuse flags are independent.  Run with -m gpu."""
import ctypes as C
import threading

import os

import numpy as np
import pytest

from helpers import MATS, assert_csc_equal, gridmetrics_of

pytestmark = pytest.mark.gpu
_vp = C.c_void_p


def _stats(lib):
    n, bu, bi = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    assert lib.otmb_host_pool_stats(C.byref(n), C.byref(bu), C.byref(bi)) == 0
    return n.value, bu.value, bi.value


def test_blocks_outlive_their_context_and_free_ignores_it():
    from otmb_amd import capi

    lib = capi.lib()
    ctx = capi.Context(0)
    n0 = _stats(lib)[0]
    ptrs = []
    for nbytes in (1, 4096, 3 << 20):
        p = _vp()
        ctx.check(lib.otmb_host_alloc(ctx.handle, nbytes, C.byref(p)))
        assert p.value
        (C.c_char * nbytes).from_address(p.value)[:] = b"\x5a" * nbytes  # usable memory
        ptrs.append(p)
    assert _stats(lib)[0] == n0 + 3
    stale = _vp(ctx.handle.value)  # what a finalizer would still hold
    ctx.close()                    # otmb_ctx_destroy: frees none of the blocks
    assert _stats(lib)[0] == n0 + 3
    assert bytes((C.c_char * 8).from_address(ptrs[2].value)[:]) == b"\x5a" * 8
    # free AFTER destroy: with a dangling context pointer (never dereferenced), and with NULL
    assert lib.otmb_host_free(stale, ptrs[0]) == 0
    assert lib.otmb_host_free(None, ptrs[1]) == 0
    assert lib.otmb_host_free(None, ptrs[2]) == 0
    assert _stats(lib)[0] == n0
    assert lib.otmb_host_free(None, ptrs[2]) == capi.OK + 11  # freed twice: OTMB_ERR_INVALID_ARG, nothing else happens
    assert lib.otmb_host_free(None, None) == 0
    # a new context gets the idle block back
    ctx2 = capi.Context(0)
    p = _vp()
    ctx2.check(lib.otmb_host_alloc(ctx2.handle, 3 << 20, C.byref(p)))
    assert p.value == ptrs[2].value
    assert lib.otmb_host_free(None, p) == 0
    ctx2.close()


def test_alloc_and_free_from_other_threads_during_a_transportmatrix_call(oracle):
    """Two threads allocate and free (as garbage-collector finalizers would) while the main thread runs transportmatrix on the
    same context with pinned result arrays; results bit-exact, no block lost."""
    import otmb_amd.api as api
    from otmb_amd import capi, synthetic

    lib = capi.lib()
    g = synthetic.make_grid(90, 80, 20, seed=61, rho="array")
    gm = gridmetrics_of(g)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    idx = api.makeindices(gm.v3D)
    ctx = api.context(0)
    n0 = _stats(lib)[0]
    stop = threading.Event()
    errors = []

    def churn(seed):
        rng = np.random.default_rng(seed)
        held = []
        try:
            while not stop.is_set():
                p = _vp()
                rc = lib.otmb_host_alloc(ctx.handle, int(rng.integers(1, 1 << 20)), C.byref(p))
                if rc != 0:
                    errors.append(("alloc", rc)); return
                held.append(p)
                if len(held) > 8:
                    q = held.pop(int(rng.integers(0, len(held))))
                    rc = lib.otmb_host_free(None, q)
                    if rc != 0:
                        errors.append(("free", rc)); return
            for q in held:
                if lib.otmb_host_free(None, q) != 0:
                    errors.append(("free at end", 1))
        except Exception as e:  # pragma: no cover
            errors.append(("exception", repr(e)))

    threads = [threading.Thread(target=churn, args=(s,)) for s in (1, 2)]
    for t in threads:
        t.start()
    try:
        for _ in range(6):
            tm = api.transportmatrix(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho)
            for m in MATS:
                assert_csc_equal(tuple(tm[m]), rtm[m], m)
            del tm
    finally:
        stop.set()
        for t in threads:
            t.join()
    assert not errors, errors[:3]
    import gc

    gc.collect()
    assert _stats(lib)[0] == n0


def test_reuse_flags_are_independent(oracle):
    """reuse_fluxes = True with reuse_grid = False must skip the six ϕ uploads (and nothing else); reuse_grid = True with
    reuse_fluxes = False the grid constants (and nothing else).  Counted in bytes (otmb_ctx_uploaded_bytes)."""
    import otmb_amd.api as api
    from otmb_amd import capi, synthetic

    lib = capi.lib()
    api._ctx.pop(9, None)
    api._ctx[9] = capi.Context(0)
    try:
        ctx = api._ctx[9]
        g = synthetic.make_grid(40, 30, 12, seed=62, rho="array")
        gm = gridmetrics_of(g)
        ref = oracle.makeindices(gm.v3D)
        rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
        rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
        idx = api.makeindices(gm.v3D, device=9)
        G, P, N = gm.v3D.size, gm.v3D.shape[0] * gm.v3D.shape[1], int(idx.N)
        phi_bytes = 6 * G * 8
        grid_bytes = 3 * G * 8 + N * 8 + 9 * P * 8 + gm.v3D.shape[2] * 8  # v3D, thkcello, Lwet3D | Lwet | 8 metrics + area | zt
        every_call = G * 8 + P * 8                                        # ρ, mlotst

        def run(inject=False, **kw):
            phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx, device=9)
            b0 = lib.otmb_ctx_uploaded_bytes(ctx.handle)
            if inject:  # (the uploads of transportmatrix fail: otmb_xfer.hip's test hook)
                os.environ["OTMB_TEST_FAIL_UPLOAD"] = "1"
            try:
                tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, device=9, **kw)
            finally:
                os.environ.pop("OTMB_TEST_FAIL_UPLOAD", None)
            for m in MATS:
                assert_csc_equal(tuple(tm[m]), rtm[m], f"{m}/{kw}")
            return lib.otmb_ctx_uploaded_bytes(ctx.handle) - b0

        assert run() == phi_bytes + grid_bytes + every_call
        assert run(reuse_fluxes=True) == grid_bytes + every_call            # the flag works WITHOUT reuse_grid
        assert run(reuse_grid=True) == phi_bytes + grid_bytes + every_call  # first call with the promise: everything still goes up
        assert run(reuse_grid=True) == phi_bytes + every_call
        assert run(reuse_grid=True, reuse_fluxes=True) == every_call
        assert run(reuse_fluxes=True) == grid_bytes + every_call            # reuse_grid off again: the grid is forgotten, ϕ is not
        assert run() == phi_bytes + grid_bytes + every_call
        # a batch that never reached the device is not remembered: the keys are written when an upload is queued, so a
        # failed transfer must take them back -- or the retry below would assemble from buffers nothing was ever copied into
        from otmb_amd.capi import OtmbError

        with pytest.raises(OtmbError) as e:
            run(inject=True, reuse_grid=True)
        assert "injected" in str(e.value)
        assert run(reuse_grid=True) == phi_bytes + grid_bytes + every_call  # everything goes up again, and the matrices are right
    finally:
        api._ctx.pop(9).close()
