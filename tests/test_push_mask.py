"""Push mask (include/otmb.h, otmb_push_mask_dev): the 16 bits per cell the counting pass of transportmatrix reads
instead of the six ϕ arrays.  Written by the facefluxes kernel, or derived from existing ϕ arrays."""
import ctypes as C

import numpy as np
import pytest

from helpers import CASES, MATS, assert_csc_equal, make_case

pytestmark = pytest.mark.gpu

W, E, S, N, B, T, WET = 1, 2, 4, 8, 16, 32, 64


def _expected_mask(phi, wet):
    """Straight from the definition: max(ϕ,0) / min(ϕ,0) / ϕ/2 non-zero (src/matrixbuilding.jl:244-289)."""
    def nz(x):
        return (x > 0) | (x < 0)

    lo = np.zeros(wet.shape, dtype=np.uint16)
    hi = np.zeros(wet.shape, dtype=np.uint16)
    for bit, key, positive in ((W, "west", True), (E, "east", False), (S, "south", True), (N, "north", False),
                               (B, "bottom", True), (T, "top", False)):
        x = phi[key]
        lo |= np.where((x > 0) if positive else (x < 0), bit, 0).astype(np.uint16)
        hi |= np.where(nz(x / 2), bit, 0).astype(np.uint16)
    w = np.where(wet != 0, WET, 0).astype(np.uint16)
    return (lo | w) | ((hi | w) << 8)


def _assembler(g, gm, upwind=True):
    import torch

    from otmb_amd.device import DeviceAssembler

    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=upwind)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    return asm, umo, vmo


@pytest.mark.parametrize("name", list(CASES))
def test_mask_written_by_facefluxes_equals_derived_mask_and_definition(oracle, name):
    import torch

    from otmb_amd import capi

    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    asm, umo, vmo = _assembler(g, gm)
    asm.count_in_ff = False  # (the counting variant of the kernel writes only the rows its seam-row pass reads: tests/test_counts_in_ff.py)
    asm.facefluxes(umo, vmo, fill)
    got = asm.push_mask.cpu().numpy().view(np.uint16)
    want = _expected_mask(rphi, ref["wet3D"]).ravel(order="F")
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:5]
    derived = torch.zeros_like(asm.push_mask)
    ptrs = capi.ptr_array(6, [p.data_ptr() for p in asm.phi])
    asm.ctx.check(asm.lib.otmb_push_mask_dev(asm.ctx.handle, C.byref(ptrs), asm.lwet3d.data_ptr(), 0, asm.G,
                                             derived.data_ptr()))
    asm.ctx.synchronize()
    assert np.array_equal(derived.cpu().numpy().view(np.uint16), want)


def test_push_mask_subrange_and_smallest_denormal():
    import torch

    from otmb_amd import capi

    ctx = capi.Context(0)
    n = 1000
    tiny = np.float64(5e-324)
    vals = np.array([0.0, -0.0, tiny, -tiny, 2 * tiny, -2 * tiny, 1.0, -1.0, np.nan, np.inf, -np.inf])
    rng = np.random.default_rng(5)
    host = {k: rng.choice(vals, size=n) for k in ("east", "west", "north", "south", "top", "bottom")}
    lw = rng.integers(0, 2, size=n).astype(np.int64) * np.arange(1, n + 1)
    order = ("east", "west", "north", "south", "top", "bottom")  # OTMB_EAST..OTMB_BOTTOM
    dev = [torch.from_numpy(host[k]).cuda() for k in order]
    dlw = torch.from_numpy(lw).cuda()
    mask = torch.full((n,), -1, dtype=torch.int16, device="cuda")
    ptrs = capi.ptr_array(6, [p.data_ptr() for p in dev])
    first, count = 137, 500
    ctx.check(capi.lib().otmb_push_mask_dev(ctx.handle, C.byref(ptrs), dlw.data_ptr(), first, count, mask.data_ptr()))
    ctx.synchronize()
    got = mask.cpu().numpy().view(np.uint16)
    want = _expected_mask(host, lw)
    assert np.array_equal(got[first:first + count], want[first:first + count])
    assert (got[:first] == 0xFFFF).all() and (got[first + count:] == 0xFFFF).all()  # outside the range: untouched
    # the smallest denormal halves to zero: pushes under upwind, not under centred weighting
    k = np.flatnonzero(host["west"] == tiny)
    assert len(k) and ((want[k] & W) != 0).all() and ((want[k] >> 8 & W) == 0).all()


@pytest.mark.parametrize("name", ["tiny_tripolar", "small_rho3d", "odd_nx_fold", "tiny_bipolar"])
def test_device_step_with_centred_weighting(oracle, name):
    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, False)
    asm, umo, vmo = _assembler(g, gm, upwind=False)
    for onepass in (True, False):
        asm.step(umo, vmo, fill, onepass=onepass)
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], rtm[m], f"{name}/{m}/onepass={onepass}")


def test_mask_is_dropped_when_fluxes_change_after_facefluxes(oracle):
    """ϕ modified in place after facefluxes (here: reversed flow, a different Tadv pattern): the assembler must not
    hand the now stale mask to the library."""
    g, gm = make_case("small_rho3d")
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    rev = {k: -v for k, v in rphi.items()}
    rtm_fwd = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    rtm_rev = oracle.transportmatrix(rev, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    assert not np.array_equal(rtm_fwd["Tadv"][1], rtm_rev["Tadv"][1])  # the patterns do differ
    asm, umo, vmo = _assembler(g, gm)
    phi = asm.facefluxes(umo, vmo, fill)
    assert asm._args(phi).push_mask is not None
    for p in phi:
        p.neg_()
    assert asm._args(phi).push_mask is None
    for twophase in (False, True):
        if twophase:
            asm.transportmatrix(phi)
        else:
            asm.transportmatrix_onepass(phi)
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], rtm_rev[m], f"{m}/twophase={twophase}")


def test_library_refuses_a_mask_that_does_not_describe_the_fluxes(oracle):
    """C-ABI misuse: the caller hands over the mask of other fluxes.  The fill pass compares its counts with the
    counting pass tile by tile, writes nothing for a tile that disagrees and reports OTMB_ERR_PUSH_MASK."""
    from otmb_amd.capi import OtmbError

    g, gm = make_case("small_rho3d")
    fill = g.umo.properties["_FillValue"]
    asm, umo, vmo = _assembler(g, gm)
    asm.count_in_ff = False  # (the same misuse with the counts that facefluxes makes itself: tests/test_counts_in_ff.py)
    phi = asm.facefluxes(umo, vmo, fill)
    for p in phi:
        p.neg_()
    asm._mask_key = asm._phi_key(phi)  # pretend the mask is still current
    assert asm._args(phi).push_mask is not None
    with pytest.raises(OtmbError) as e:
        asm.transportmatrix_onepass(phi)
    assert e.value.name == "PUSH_MASK"
    with pytest.raises(OtmbError) as e:
        asm.transportmatrix(phi)
    assert e.value.name == "PUSH_MASK"
    # and the context is still usable
    asm.step(umo, vmo, fill)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)
