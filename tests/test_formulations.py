"""The speed-only switches of transportmatrix must never change a result (run with -m gpu): every tile order of the fill pass
(otmb_ctx_set_tile_order), every work mapping and the placement search against the oracle, bit for bit, in both protocols."""
import os

import numpy as np
import pytest

from helpers import MATS, assert_csc_equal, gridmetrics_of, randomize_metrics

pytestmark = pytest.mark.gpu


def _setup(oracle, case, upwind=True):
    import torch

    from otmb_amd import synthetic
    from otmb_amd.device import DeviceAssembler

    nx, ny, nz, seed, rho, topo = case
    g = synthetic.make_grid(nx, ny, nz, seed=seed, rho=rho, topology=topo)
    gm = gridmetrics_of(g)
    if nx % 2 == 1 and topo == "tripolar":  # (the fold-centre cell is its own north neighbour: the real distance is 0)
        randomize_metrics(gm)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=upwind)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    return asm, umo, vmo, rtm


def _check(asm, rtm, what):
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], f"{what}/{m}")


def test_every_tile_order_gives_the_same_matrices(oracle):
    asm, umo, vmo, rtm = _setup(oracle, (120, 100, 23, 86, "array", "tripolar"))  # 600 tiles: above the threshold of the march order
    for rows in (0, 1, 3, 8, 64, 1000, -1):
        asm.ctx.set_tile_order(rows)
        asm.step(umo, vmo, 1e20)
        _check(asm, rtm, f"tile order {rows}")
        asm.step(umo, vmo, 1e20, onepass=False)
        _check(asm, rtm, f"tile order {rows}, two-phase")



@pytest.mark.parametrize("cols", ["0", "1", "17", "50", "64", "119", "120", "100000"])
def test_march_order_in_blocks_of_columns_gives_the_same_matrices(oracle, monkeypatch, cols):
    """Round 4: on grids with long rows the march order's buckets are (band, block of columns, level) (OTMB_MARCH_AUTO_COLS; a context
    reads OTMB_MARCH_COLS when it is created).  Any block width -- narrower than a tile, not a divisor of nx, wider than the row --
    is a permutation of the tiles and gives the oracle's matrices bit for bit, in both protocols."""
    monkeypatch.setenv("OTMB_MARCH_COLS", cols)
    asm, umo, vmo, rtm = _setup(oracle, (120, 100, 23, 89, "array", "tripolar"))
    for rows in (-1, 3):
        asm.ctx.set_tile_order(rows)
        asm.step(umo, vmo, 1e20)
        _check(asm, rtm, f"cols {cols}/order {rows}/async")
        asm.step(umo, vmo, 1e20, onepass=False)
        _check(asm, rtm, f"cols {cols}/order {rows}/two-phase")


def test_choose_placement_never_changes_a_result(oracle):
    """DeviceAssembler.choose_placement (set-up: which allocations the flux arrays and the matrices live in, chosen by timing) keeps the fastest of
    its candidates and leaves every result as it was, in both protocols; a candidate count that cannot fit is skipped, not an error."""
    asm, umo, vmo, rtm = _setup(oracle, (120, 100, 23, 90, "array", "tripolar"))
    assert "skipped" in asm.choose_placement(umo, vmo, 1e20, candidates=3)  # (a small grid: nothing to choose by default)
    rec = asm.choose_placement(umo, vmo, 1e20, candidates=3, reps=2, min_output_bytes=0)
    assert rec["chosen"] is not None and len(rec["fill_ms"]) == 3 and len(rec["facefluxes_ms"]) == 3, rec
    assert rec["chosen"] == [int(np.argmin(rec["facefluxes_ms"])), int(np.argmin(rec["fill_ms"]))]
    asm.step(umo, vmo, 1e20)
    _check(asm, rtm, "after choose_placement/async")
    asm.step(umo, vmo, 1e20, onepass=False)
    _check(asm, rtm, "after choose_placement/two-phase")
    for _ in range(3):
        asm.step_async(umo, vmo, 1e20)
    asm.finish()
    _check(asm, rtm, "after choose_placement/pipeline")
    assert asm.choose_placement(umo, vmo, 1e20, candidates=1, min_output_bytes=0)["chosen"] is None
    assert "skipped" in asm.choose_placement(umo, vmo, 1e20, candidates=10 ** 9, min_output_bytes=0)

@pytest.mark.skipif(bool(os.environ.get("OTMB_MARCH_ROWS")),
                    reason="counts the default tile order's kernels (the suite is also run under the library's experiment switches)")
def test_tile_order_is_computed_once_per_grid(oracle):
    """The march order is a function of the grid: its three kernels run on the first step and again only when the band height changes
    (otmb_ctx_set_tile_order); the default (-1) is the march order, i.e. they do run."""
    asm, umo, vmo, rtm = _setup(oracle, (120, 100, 23, 87, "array", "tripolar"))
    asm.ctx.timing_enable(True)
    for _ in range(3):
        asm.step(umo, vmo, 1e20)
    t = asm.ctx.timing_collect()
    assert t["tm_order_kernels"][1] == 1, t
    assert t["tm_kernel<fill>"][1] == 3, t
    asm.ctx.set_tile_order(5)
    for _ in range(2):
        asm.step(umo, vmo, 1e20)
    t = asm.ctx.timing_collect()
    assert t["tm_order_kernels"][1] == 1, t
    _check(asm, rtm, "after the order was rebuilt")
    asm.ctx.set_tile_order(0)
    asm.step(umo, vmo, 1e20)
    assert "tm_order_kernels" not in asm.ctx.timing_collect()


@pytest.mark.parametrize("env", [
    dict(OTMB_FF_XCD="0", OTMB_COUNT_ORDER="0", OTMB_DEAL_HEAVY="0", OTMB_FF_ROWS="1"),   # rounds 1-3: blockIdx order everywhere
    dict(OTMB_FF_XCD="1", OTMB_COUNT_ORDER="1", OTMB_DEAL_HEAVY="1", OTMB_FF_ROWS="4"),
    dict(OTMB_FF_XCD="1", OTMB_COUNT_ORDER="2", OTMB_DEAL_HEAVY="0", OTMB_FF_ROWS="4"),
    dict(OTMB_FF_XCD="0", OTMB_COUNT_ORDER="2", OTMB_DEAL_HEAVY="1", OTMB_FF_ROWS="1"),
], ids=lambda e: "-".join(f"{k[5:].lower()}{v}" for k, v in e.items()))
def test_work_mappings_never_change_a_result(oracle, monkeypatch, env):
    """Round 4's speed-only mappings (a context reads them from the environment when it is created): facefluxes and the counting
    pass in XCD-contiguous eighths, the counting pass in the fill pass's tile order, the heavy (seam-row) tiles of the fill pass dealt
    over the XCDs, four-row facefluxes workgroups -- every combination gives the oracle's matrices and face fluxes bit for bit, in
    both protocols, on a tripolar grid with 600 tiles (above the threshold of the tile order) whose rows are not a multiple of 64 cells."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case = (120, 100, 23, 88, "array", "tripolar")
    asm, umo, vmo, rtm = _setup(oracle, case)
    for rows in (-1, 3):
        asm.ctx.set_tile_order(rows)
        asm.step(umo, vmo, 1e20)
        _check(asm, rtm, f"{env}/order {rows}/async")
        asm.step(umo, vmo, 1e20, onepass=False)
        _check(asm, rtm, f"{env}/order {rows}/two-phase")
    # the face fluxes themselves (four-row workgroups, XCD chunks): against the oracle on the same grid
    from otmb_amd import synthetic

    nx, ny, nz, seed, rho, topo = case
    g = synthetic.make_grid(nx, ny, nz, seed=seed, rho=rho, topology=topo)
    gm = gridmetrics_of(g)
    ref = oracle.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    for k, name in enumerate(("east", "west", "north", "south", "top", "bottom")):
        got = asm.phi[k].cpu().numpy().reshape((nx, ny, nz), order="F")
        same = (got == rphi[name]) & (np.signbit(got) == np.signbit(rphi[name]))
        assert same.all(), (env, name)
