"""An asynchronous pipeline of steps never loses an error (run with -m gpu).

The reference throws where the failure is (src/matrixbuilding.jl:39,61,90,114,233; src/velocities.jl:199-200).  The
device pipeline (step_async x K, then finish) checks after the fact, so every step keeps its own error flags on the
device (a ring of state blocks) and finish() raises the FIRST failing step's error with the step's index."""
import numpy as np
import pytest

from helpers import MATS, assert_csc_equal, make_case

pytestmark = pytest.mark.gpu


def _setup(oracle, name="small_rho3d"):
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    return g, gm, ref, rtm, asm, umo, vmo


@pytest.mark.parametrize("bad_step", [0, 1, 4])
def test_rho_nan_in_one_step_of_five_is_reported_with_its_step(oracle, bad_step):
    from otmb_amd.capi import OtmbError

    g, gm, ref, rtm, asm, umo, vmo = _setup(oracle)
    L = int(ref["Lwet"][ref["N"] // 3] - 1)
    for sidx in range(5):
        if sidx == bad_step:  # stream-ordered in-place edits: only this step sees the NaN
            old = asm.rho[L].clone()
            asm.rho[L] = float("nan")
        asm.step_async(umo, vmo, 1e20)
        if sidx == bad_step:
            asm.rho[L] = old
    with pytest.raises(OtmbError, match="ρ contains NaNs") as e:
        asm.finish()
    assert e.value.step == bad_step and f"step {bad_step + 1} of 5" in str(e.value)
    # the pipeline is clean again
    for _ in range(3):
        asm.step_async(umo, vmo, 1e20)
    asm.finish()
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)


def test_first_of_two_different_failures_wins(oracle):
    """step 2: a metric NaN (TκH contains NaNs.), step 3: NaN in ρ -- the reference would have stopped at step 2."""
    from otmb_amd.capi import OtmbError

    g, gm, ref, rtm, asm, umo, vmo = _setup(oracle)
    wet = ref["wet3D"].astype(bool)[:, :, 0]
    ii, jj = np.argwhere(wet & np.roll(wet, 1, axis=0))[0]
    s2 = int(ii + wet.shape[0] * jj)
    L = int(ref["Lwet"][5] - 1)
    for sidx in range(4):
        if sidx == 1:
            old_e = asm.edge[0][s2].clone()  # OTMB_DIR_WEST
            asm.edge[0][s2] = float("nan")
        if sidx == 2:
            old_r = asm.rho[L].clone()
            asm.rho[L] = float("nan")
        asm.step_async(umo, vmo, 1e20)
        if sidx == 1:
            asm.edge[0][s2] = old_e
        if sidx == 2:
            asm.rho[L] = old_r
    with pytest.raises(OtmbError, match="TκH contains NaNs.") as e:
        asm.finish()
    assert e.value.step == 1 and e.value.name == "TKH_NAN"


def test_all_missing_field_in_step_three_of_five(oracle):
    """velocities.jl:199-200 fires only when NOTHING is valid after nofluxboundaries!: an all-wet mask and an all-NaN umo."""
    import torch

    from otmb_amd.capi import OtmbError

    g, gm, ref, rtm, asm, umo, vmo = _setup(oracle)
    asm.wet3d.fill_(1)
    bad = torch.full_like(umo, float("nan"))
    for sidx in range(5):
        asm.facefluxes_async(bad if sidx == 2 else umo, vmo, 1e20)  # transportmatrix would trip over the fake mask: fluxes only
    with pytest.raises(OtmbError, match="AssertionError") as e:
        asm.finish_facefluxes()
    assert e.value.step == 2 and "step 3 of 5" in str(e.value)
    for sidx in range(3):
        asm.facefluxes_async(umo, vmo, 1e20)
    asm.finish_facefluxes()


def test_pipeline_longer_than_the_ring_keeps_an_early_failure(oracle):
    """150 steps with a failure in step 7: the library folds its 64-slot rings as they fill up, the assembler drains its
    pipeline every 60 steps -- the error surfaces at the first drain at the latest, with the right index."""
    from otmb_amd.capi import OtmbError

    g, gm, ref, rtm, asm, umo, vmo = _setup(oracle, "tiny_rho3d")
    L = int(ref["Lwet"][3] - 1)
    with pytest.raises(OtmbError, match="ρ contains NaNs") as e:
        for sidx in range(150):
            if sidx == 7:
                old = asm.rho[L].clone()
                asm.rho[L] = float("nan")
            asm.step_async(umo, vmo, 1e20)
            if sidx == 7:
                asm.rho[L] = old
        asm.finish()
    assert e.value.step == 7
    # the C ABI on its own (no periodic drain by the host object): 100 raw calls, failure in call 5
    phi = asm.facefluxes(umo, vmo, 1e20)
    for sidx in range(100):
        if sidx == 5:
            old = asm.rho[L].clone()
            asm.rho[L] = float("nan")
        asm.transportmatrix_onepass(phi, sync=False)
        if sidx == 5:
            asm.rho[L] = old
    with pytest.raises(OtmbError, match="ρ contains NaNs") as e:
        asm.result()
    assert e.value.step == 5 and "step 6 of 100" in str(e.value)


def test_clean_pipeline_across_the_ring_returns_the_last_step(oracle):
    """130 raw asynchronous calls, no failure: the state blocks are zeroed by the preceding fill pass and fetched when the ring is
    full and at result() -- the totals and the matrices are the last step's, bit for bit."""
    g, gm, ref, rtm, asm, umo, vmo = _setup(oracle, "tiny_rho3d")
    phi = asm.facefluxes(umo, vmo, 1e20)
    for _ in range(130):
        asm.transportmatrix_onepass(phi, sync=False)
    out = asm.result()
    for k, m in enumerate(MATS):
        n = asm.nnz[k]
        assert n == len(rtm[m][1])
        cp, rv, nz = (t.cpu().numpy() for t in out[m])
        assert_csc_equal((cp, rv[:n], nz[:n]), rtm[m], m)
    # and a failure after a clean drain is reported relative to the new pipeline
    from otmb_amd.capi import OtmbError

    L = int(ref["Lwet"][3] - 1)
    for sidx in range(3):
        if sidx == 1:
            old = asm.rho[L].clone()
            asm.rho[L] = float("nan")
        asm.transportmatrix_onepass(phi, sync=False)
        if sidx == 1:
            asm.rho[L] = old
    with pytest.raises(OtmbError, match="ρ contains NaNs") as e:
        asm.result()
    assert e.value.step == 1 and "step 2 of 3" in str(e.value)


@pytest.mark.parametrize("fault", ["1:1:rho", "2:3:rho"])
def test_hip_slab_pipeline_reports_first_failing_step_on_every_rank(oracle, tmp_path, fault):
    from test_dist_cpu import check_against_whole_grid, check_fault_reports, run_ranks

    case = (24, 18, 11, 35, "array", "tripolar")
    z = run_ranks(3, "hip", case, tmp_path, fault=fault)
    check_fault_reports(tmp_path, 3, "OtmbError", int(fault.split(":")[1]), "ρ contains NaNs")
    check_against_whole_grid(oracle, z, case)


def test_every_asynchronous_step_keeps_its_own_matrices(oracle):
    """Two asynchronous calls into two output sets; the FIRST one with exact cancellations in T (κ = 0: every diffusive
    value is an explicit zero that sparse() keeps in the operators and `+` drops from T, src/matrixbuilding.jl:147), the
    second an ordinary one.  Both must come out as the reference's matrices, each with its own nnz: every transportmatrix
    call of the reference returns its own five matrices (src/matrixbuilding.jl:147-149)."""
    g, gm, ref, rtm, asm, umo, vmo = _setup(oracle, "tiny_tripolar")
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm0 = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, 0.0, 0.0, 0.0, True)
    phi = asm.facefluxes(umo, vmo, 1e20)
    outs = [asm.new_output_set() for _ in range(3)]
    kappa = asm.kappa
    asm.kappa = (0.0, 0.0, 0.0)
    asm.transportmatrix_onepass(phi, sync=False, out=outs[0])
    asm.kappa = kappa
    asm.transportmatrix_onepass(phi, sync=False, out=outs[1])
    asm.kappa = (0.0, 0.0, 0.0)
    asm.transportmatrix_onepass(phi, sync=False, out=outs[2])
    asm.kappa = kappa
    asm.result()
    for k, want in enumerate((rtm0, rtm, rtm0)):
        rc, nnz = asm.result_step(k)
        assert rc == 0
        assert nnz == [len(want[m][1]) for m in MATS], (k, nnz)
        for q, m in enumerate(MATS):
            cp, rv, nz = (t.cpu().numpy() for t in outs[k][m])
            assert_csc_equal((cp, rv[:nnz[q]], nz[:nnz[q]]), want[m], f"step {k} {m}")
    assert len(rtm0["T"][1]) < len(rtm["T"][1])  # the cancellation really happened
    # one output set reused by every call (what bench.py does): the last call's matrices are what it holds
    asm.kappa = (0.0, 0.0, 0.0)
    asm.transportmatrix_onepass(phi, sync=False)
    asm.kappa = kappa
    asm.transportmatrix_onepass(phi, sync=False)
    out = asm.result()
    for q, m in enumerate(MATS):
        cp, rv, nz = (t.cpu().numpy() for t in out[m])
        assert_csc_equal((cp, rv[:asm.nnz[q]], nz[:asm.nnz[q]]), rtm[m], m)


def test_a_failed_step_does_not_hide_the_results_of_the_others(oracle):
    from otmb_amd.capi import OtmbError

    g, gm, ref, rtm, asm, umo, vmo = _setup(oracle, "tiny_rho3d")
    phi = asm.facefluxes(umo, vmo, 1e20)
    outs = [asm.new_output_set() for _ in range(3)]
    L = int(ref["Lwet"][3] - 1)
    for k in range(3):
        if k == 1:
            old = asm.rho[L].clone()
            asm.rho[L] = float("nan")
        asm.transportmatrix_onepass(phi, sync=False, out=outs[k])
        if k == 1:
            asm.rho[L] = old
    with pytest.raises(OtmbError, match="ρ contains NaNs") as e:
        asm.result()
    assert e.value.step == 1
    for k in (0, 2):
        rc, nnz = asm.result_step(k)
        assert rc == 0 and nnz == [len(rtm[m][1]) for m in MATS]
        for q, m in enumerate(MATS):
            cp, rv, nz = (t.cpu().numpy() for t in outs[k][m])
            assert_csc_equal((cp, rv[:nnz[q]], nz[:nnz[q]]), rtm[m], f"step {k} {m}")
    assert asm.result_step(1)[0] == 1  # OTMB_ERR_RHO_NAN
    assert asm.result_step(3)[0] == 11  # no such step


def test_failures_of_unpaired_calls_are_ordered_by_issue(oracle):
    """A facefluxes call that is not part of a step (no transportmatrix after it) shifts the two kinds of index against each
    other: the failure that was ISSUED first must win (facefluxes call 2 comes after transportmatrix call 0 here)."""
    import torch

    from otmb_amd.capi import OtmbError

    g, gm, ref, rtm, asm, umo, vmo = _setup(oracle, "tiny_rho3d")
    L = int(ref["Lwet"][3] - 1)
    old = asm.rho[L].clone()
    asm.rho[L] = float("nan")
    asm.step_async(umo, vmo, 1e20)      # facefluxes call 0 (fine), transportmatrix call 0 (ρ NaN)
    asm.rho[L] = old
    asm.facefluxes_async(umo, vmo, 1e20)  # facefluxes call 1, unpaired
    wet = asm.wet3d.clone()
    asm.wet3d.fill_(1)
    asm.facefluxes_async(torch.full_like(umo, float("nan")), vmo, 1e20)  # facefluxes call 2: nothing valid
    asm.wet3d.copy_(wet)
    with pytest.raises(OtmbError, match="ρ contains NaNs") as e:
        asm.finish()
    assert e.value.step == 0
