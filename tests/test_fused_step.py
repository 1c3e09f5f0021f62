"""The fused device-resident step (include/otmb.h, otmb_step_dev): (umo, vmo) -> five matrices with only ϕtop ever stored; the fill pass
re-derives the other five fluxes from umo / vmo where it uses them.  An extension beside the two-call API (facefluxesfrommasstransport
returns the six arrays, src/velocities.jl:245-254) -- and bit-identical to it: every case against the oracle."""
import numpy as np
import pytest

from helpers import CASES, COUNTS_ON, MATS, assert_csc_equal, gridmetrics_of, make_case

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not COUNTS_ON, reason="OTMB_COUNT_IN_FF=0: otmb_step_dev needs the counts of its own facefluxes")]


def _assembler(g, gm, upwind=True, only_T=False):
    import torch

    from otmb_amd.device import DeviceAssembler

    asm = DeviceAssembler(0)
    asm.only_T = only_T
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=upwind)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    return asm, umo, vmo


def _reference(oracle, g, gm, upwind=True):
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    return oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind), rphi, fill


@pytest.mark.parametrize("upwind", [True, False])
@pytest.mark.parametrize("name", [n for n in CASES if CASES[n][0]["nx"] >= 3])
def test_fused_step_equals_the_oracle(oracle, name, upwind):
    g, gm = make_case(name)
    rtm, rphi, fill = _reference(oracle, g, gm, upwind)
    asm, umo, vmo = _assembler(g, gm, upwind=upwind)
    asm.ctx.timing_enable(True)
    asm.step_fused_async(umo, vmo, fill)
    asm.finish()
    k = asm.ctx.timing_collect()
    asm.ctx.timing_enable(False)
    assert "tm_count_kernel" not in k and "push_mask_kernel" not in k, sorted(k)
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], f"{name}/{m}")
    top = asm.phi_top.cpu().numpy().reshape(rphi["top"].shape, order="F")
    assert np.array_equal(top, rphi["top"])  # the one flux array that is stored is facefluxes' ϕtop


def test_nx2_is_refused(oracle):
    from otmb_amd.capi import OtmbError

    g, gm = make_case("nx2")
    asm, umo, vmo = _assembler(g, gm)
    with pytest.raises(OtmbError) as e:
        asm.step_fused_async(umo, vmo, g.umo.properties["_FillValue"])
    assert e.value.name == "INVALID_ARG"


@pytest.mark.parametrize("rows", [1, 4])
@pytest.mark.parametrize("topology", ["tripolar", "bipolar"])
def test_fused_step_on_both_wave_geometries(oracle, monkeypatch, rows, topology):
    from otmb_amd import synthetic

    monkeypatch.setenv("OTMB_FF_ROWS", str(rows))
    g = synthetic.make_grid(150, 13, 7, seed=32, rho="array", topology=topology, dtype_flux=np.float32)
    gm = gridmetrics_of(g)
    for upwind in (True, False):
        rtm, _, fill = _reference(oracle, g, gm, upwind)
        asm, umo, vmo = _assembler(g, gm, upwind=upwind)
        assert umo.dtype.is_floating_point and umo.element_size() == 4
        asm.step_fused_async(umo, vmo, fill)
        asm.finish()
        got = asm.result_to_host()
        for m in MATS:
            assert_csc_equal(got[m], rtm[m], f"{m}/rows={rows}/{topology}/upwind={upwind}")


def test_fused_and_two_call_steps_in_one_pipeline(oracle):
    """Twelve asynchronous steps with different fields, fused and two-call steps alternating on one context (the count buffers, the
    validity-flag ring and the state ring are shared): every step's matrices equal the oracle's for ITS field."""
    import torch

    g, gm = make_case("small_rho3d")
    fill = g.umo.properties["_FillValue"]
    asm, umo, vmo = _assembler(g, gm)
    ref = oracle.makeindices(gm.v3D)
    gen = np.random.default_rng(7)
    fields, sets = [], []
    for s in range(12):
        su = np.where(gen.random(g.umo.data.shape) < 0.5, -1.0, 1.0)
        u = np.where(g.umo.data == fill, g.umo.data, g.umo.data * su)
        v = np.where(g.vmo.data == fill, g.vmo.data, g.vmo.data * su[::-1])
        fields.append((u, v))
    dev = lambda a: torch.from_numpy(np.asfortranarray(a).ravel(order="F")).cuda()
    for s, (u, v) in enumerate(fields):
        o = asm.new_output_set()
        sets.append(o)
        if s % 2 == 0:
            asm.step_fused_async(dev(u), dev(v), fill, out=o)
        else:
            asm.transportmatrix_onepass(asm.facefluxes_async(dev(u), dev(v), fill), sync=False, out=o)
    asm.finish()
    for s, (u, v) in enumerate(fields):
        rphi = oracle.facefluxes(u, v, ref["wet3D"], fill, gm.gridtopology.kind)
        rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
        rc, nnz = asm.result_step(s)
        assert rc == 0
        for k, m in enumerate(MATS):
            o = sets[s][m]
            assert_csc_equal((o[0].cpu().numpy(), o[1][: nnz[k]].cpu().numpy(), o[2][: nnz[k]].cpu().numpy()), rtm[m], f"step {s}/{m}")


def test_fused_step_T_alone_and_errors(oracle):
    import torch

    from otmb_amd.capi import OtmbError

    g, gm = make_case("small_rho3d")
    rtm, _, fill = _reference(oracle, g, gm)
    asm, umo, vmo = _assembler(g, gm, only_T=True)
    asm.step_fused_async(umo, vmo, fill)
    asm.finish()
    got = asm.result_to_host()
    assert_csc_equal(got["T"], rtm["T"], "T")
    assert asm.nnz[1:] == [0, 0, 0, 0]
    # the reference's errors, from the fused step: ρ with a NaN on a wet cell (src/matrixbuilding.jl:233) ...
    asm, umo, vmo = _assembler(g, gm)
    L = int(asm.lwet[asm.N // 2].item()) - 1
    asm.rho[L] = float("nan")
    asm.step_fused_async(umo, vmo, fill)
    with pytest.raises(OtmbError) as e:
        asm.finish()
    assert e.value.name == "RHO_NAN"
    # ... and fields without a single valid value (src/velocities.jl:199-200; nofluxboundaries! zeroes land first, so: all wet + all NaN)
    asm, umo, vmo = _assembler(g, gm)
    wet = asm.wet3d.clone()
    asm.wet3d.fill_(1)
    asm.step_fused_async(torch.full_like(umo, float("nan")), vmo, fill)
    with pytest.raises(OtmbError) as e:
        asm.finish()
    assert e.value.name == "ALL_MISSING"
    asm.wet3d.copy_(wet)
    asm.step_fused_async(umo, vmo, fill)  # and the context is still usable
    asm.finish()
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)
