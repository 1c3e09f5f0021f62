"""Counts in facefluxes (include/otmb.h, otmb_facefluxes_counts_dev): the facefluxes kernel accumulates, per tile of 256 matrix
columns, the five row counts of the transportmatrix that its fluxes are handed to -- a column's presence bits follow from the cell's OWN
six fluxes (src/velocities.jl:206-224, :238-240) -- so that the device-resident step has no counting pass.  Everything here is compared
with the oracle bit for bit, and with the same library run with the counting pass (count_in_ff = False)."""
import numpy as np
import pytest

from helpers import CASES, COUNTS_ON, MATS, assert_csc_equal, gridmetrics_of, make_case

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not COUNTS_ON, reason="OTMB_COUNT_IN_FF=0: the feature under test is switched off")]


def _assembler(g, gm, upwind=True, count_in_ff=True, only_T=False):
    import torch

    from otmb_amd.device import DeviceAssembler

    asm = DeviceAssembler(0)
    asm.count_in_ff = count_in_ff
    asm.only_T = only_T
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=upwind)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    return asm, umo, vmo


def _reference(oracle, g, gm, upwind=True):
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    return oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind), fill


def _kernels_of_one_step(asm, umo, vmo, fill, onepass=True):
    asm.ctx.timing_enable(True)
    asm.step(umo, vmo, fill, onepass=onepass)
    k = asm.ctx.timing_collect()
    asm.ctx.timing_enable(False)
    return k


@pytest.mark.parametrize("upwind", [True, False])
@pytest.mark.parametrize("name", list(CASES))
def test_step_without_a_counting_pass_equals_the_oracle(oracle, name, upwind):
    g, gm = make_case(name)
    rtm, fill = _reference(oracle, g, gm, upwind)
    asm, umo, vmo = _assembler(g, gm, upwind=upwind)
    for onepass in (True, False):  # asynchronous and two-phase protocol
        k = _kernels_of_one_step(asm, umo, vmo, fill, onepass)
        if g.umo.data.shape[0] >= 3:  # (nx < 3: row-mates coincide everywhere, the library counts with its own pass)
            assert "tm_count_kernel" not in k and "push_mask_kernel" not in k, sorted(k)
        else:
            assert "tm_count_kernel" in k
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], rtm[m], f"{name}/{m}/onepass={onepass}")


@pytest.mark.parametrize("rows", [1, 4])
@pytest.mark.parametrize("topology", ["tripolar", "bipolar"])
def test_both_wave_geometries_on_rows_that_are_no_multiple_of_a_wave(oracle, monkeypatch, rows, topology):
    """One-row waves (64 consecutive cells of the plane, crossing rows) and four-row workgroups (a wave = 64 cells of one row): rows of
    150 cells leave a partial wave at the end of every row / of the plane, levels of a few hundred wet cells put several tiles and
    many tile boundaries inside waves."""
    from otmb_amd import synthetic

    monkeypatch.setenv("OTMB_FF_ROWS", str(rows))
    g = synthetic.make_grid(150, 13, 7, seed=31, rho="array", topology=topology)
    gm = gridmetrics_of(g)
    for upwind in (True, False):
        rtm, fill = _reference(oracle, g, gm, upwind)
        asm, umo, vmo = _assembler(g, gm, upwind=upwind)
        k = _kernels_of_one_step(asm, umo, vmo, fill)
        assert "tm_count_kernel" not in k and "facefluxes_kernel" in k
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], rtm[m], f"{m}/rows={rows}/{topology}/upwind={upwind}")


def test_same_matrices_with_and_without_the_counting_pass_over_a_pipeline(oracle):
    """Ten asynchronous steps whose fields differ (every step its own pattern): the two count buffers alternate, each is left zeroed by
    the scan that consumed it; every step's matrices equal the run with the counting pass."""
    import torch

    g, gm = make_case("small_rho3d")
    fill = g.umo.properties["_FillValue"]
    a1, umo, vmo = _assembler(g, gm)
    a0, _, _ = _assembler(g, gm, count_in_ff=False)
    outs = {0: [], 1: []}
    fields = []
    gen = torch.Generator(device="cpu").manual_seed(5)
    for s in range(10):
        su = torch.where(torch.rand(umo.numel(), generator=gen) < 0.5, -1.0, 1.0).to(umo.dtype).cuda()
        fields.append((torch.where(umo == fill, umo, umo * su), torch.where(vmo == fill, vmo, vmo * su.flip(0))))
    for which, asm in ((1, a1), (0, a0)):
        sets = [asm.new_output_set() for _ in fields]
        for (u, v), o in zip(fields, sets):
            asm.transportmatrix_onepass(asm.facefluxes_async(u, v, fill), sync=False, out=o)
        asm.finish()
        for q, o in enumerate(sets):
            rc, nnz = asm.result_step(q)
            assert rc == 0
            outs[which].append([(o[m][0].cpu().numpy(), o[m][1][: nnz[k]].cpu().numpy(), o[m][2][: nnz[k]].cpu().numpy())
                                for k, m in enumerate(MATS)])
    assert not np.array_equal(outs[1][0][1][1], outs[1][1][1][1])  # the steps' Tadv patterns do differ
    for q in range(len(fields)):
        for k, m in enumerate(MATS):
            assert_csc_equal(outs[1][q][k], outs[0][q][k], f"step {q}/{m}")


def test_T_alone(oracle):
    g, gm = make_case("small_rho3d")
    rtm, fill = _reference(oracle, g, gm)
    asm, umo, vmo = _assembler(g, gm, only_T=True)
    k = _kernels_of_one_step(asm, umo, vmo, fill)
    assert "tm_count_kernel" not in k
    got = asm.result_to_host()
    assert_csc_equal(got["T"], rtm["T"], "T")
    assert asm.nnz[1:] == [0, 0, 0, 0]


def test_stale_counts_are_not_used(oracle):
    """What the counts were made for is part of their key: other mixed-layer depths, another weighting or fluxes modified in place
    between facefluxes and transportmatrix make the library count for itself (and the two-row mask of the counting kernel is never
    taken for a push mask)."""
    import torch

    g, gm = make_case("small_rho3d")
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    asm, umo, vmo = _assembler(g, gm)
    # (1) mlotst changes in place after facefluxes
    ml2 = np.asfortranarray(g.mlotst * 0.25)
    phi = asm.facefluxes(umo, vmo, fill)
    asm.mlotst.copy_(torch.from_numpy(ml2.ravel(order="F")).cuda())
    assert asm._args(phi).push_mask is None
    asm.transportmatrix_onepass(phi)
    want = oracle.transportmatrix(rphi, gm, ref, g.rho, ml2, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    assert not np.array_equal(want["TκVML"][1], oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)["TκVML"][1])
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], want[m], f"mlotst/{m}")
    # (2) the weighting changes between the two calls
    phi = asm.facefluxes(umo, vmo, fill)
    asm.upwind = False
    asm.transportmatrix_onepass(phi)
    want = oracle.transportmatrix(rphi, gm, ref, g.rho, ml2, g.kappaH, g.kappaVML, g.kappaVdeep, False)
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], want[m], f"centred/{m}")
    asm.upwind = True
    # (3) the same fluxes assembled twice: the counts are consumed by the first call, the second counts for itself
    phi = asm.facefluxes(umo, vmo, fill)
    want = oracle.transportmatrix(rphi, gm, ref, g.rho, ml2, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    for q in range(2):
        asm.ctx.timing_enable(True)
        asm.transportmatrix_onepass(phi)
        k = asm.ctx.timing_collect()
        asm.ctx.timing_enable(False)
        assert ("tm_count_kernel" in k) == (q == 1), (q, sorted(k))
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], want[m], f"twice[{q}]/{m}")


def test_counts_that_do_not_describe_the_fluxes_are_refused(oracle):
    """C-ABI misuse: fluxes modified behind the library's back but handed over with the pointers the counts are keyed to.  The fill pass
    compares every tile with what it builds and refuses (OTMB_ERR_PUSH_MASK); the context stays usable."""
    from otmb_amd.capi import OtmbError

    g, gm = make_case("small_rho3d")
    rtm, fill = _reference(oracle, g, gm)
    asm, umo, vmo = _assembler(g, gm)
    for twophase in (False, True):
        phi = asm.facefluxes(umo, vmo, fill)
        for p in phi:
            p.neg_()
        asm._mask_key = asm._phi_key(phi)  # pretend nothing happened
        with pytest.raises(OtmbError) as e:
            asm.transportmatrix(phi) if twophase else asm.transportmatrix_onepass(phi)
        assert e.value.name == "PUSH_MASK"
        asm.step(umo, vmo, fill)
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], rtm[m], m)


def test_vertical_flux_into_land_is_found_by_the_counting_kernel(oracle):
    """West / east / south / north fluxes are zero towards land by construction (src/velocities.jl:167-173); the vertical ones follow
    from continuity and can point into a land cell ABOVE a wet one (an overhang): the reference then indexes Lwet3D[land]
    (src/matrixbuilding.jl:289-296).  Here: one column's upper cell is made land after the fields were generated."""
    import otmb_amd
    from otmb_amd import synthetic
    from otmb_amd.capi import OtmbError

    g = synthetic.make_grid(36, 30, 10, seed=9, rho="array")
    from otmb_amd._nt import Cube

    vol = np.array(g.volcello.data, dtype=np.float64, order="F")
    wet = ~(np.isnan(vol) | (vol == 0))
    # a wet column of at least three levels, away from the seam row: make level 1 (0-based) land -> level 2's top flux meets land
    cand = np.argwhere(wet[:, 2:-3, 0] & wet[:, 2:-3, 1] & wet[:, 2:-3, 2])
    i, j = (int(x) for x in cand[len(cand) // 2])
    j += 2
    vol[i, j, 1] = 0.0
    volc = Cube(vol, **g.volcello.properties)
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=volc, lon=g.lon, lat=g.lat, lev=g.lev,
                                  lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    fill = g.umo.properties["_FillValue"]
    for count_in_ff in (True, False):
        asm, umo, vmo = _assembler(g, gm, count_in_ff=count_in_ff)
        for onepass in (True, False):
            with pytest.raises(OtmbError) as e:
                asm.step(umo, vmo, fill, onepass=onepass)
            assert e.value.name == "FLUX_INTO_LAND", (count_in_ff, onepass)
