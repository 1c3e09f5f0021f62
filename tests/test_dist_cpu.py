"""Multi-process depth-slab path on CPU (gloo, world_size 2 and 3): the distributed orchestration of
otmb_amd.dist (partition, static halo exchange, global wet ranks, the ϕtop chain across slabs, nnz
all_gather, colptr bases, concatenation) with the oracle as the per-slab compute backend, checked bit
for bit against the oracle run on the whole grid in one process."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import otmb_amd
from helpers import MATS, assert_csc_equal
from otmb_amd import dist as od, synthetic

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def whole_grid_reference(oracle, nx, ny, nz, seed, rho, topo, upwind=True):
    g = synthetic.make_slab(nx, ny, nz, 0, nz, seed=seed, rho=rho, topology=topo)
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                  lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    idx = oracle.makeindices(gm.v3D)
    phi = oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], 1e20, gm.gridtopology.kind)
    return idx, oracle.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)


def run_ranks(world, kind, case, outdir, timeout=300, async_mode=False, fault=None, rccl=False, pieces=None, extra_env=None):
    port = free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OTMB_TEST_ASYNC="1" if async_mode else "0", OTMB_TEST_RCCL="1" if rccl else "0")
    env.update(extra_env or {})
    if pieces is not None:
        env["OTMB_CHAIN_PIECES"] = str(pieces)  # SlabRunner: the facefluxes chain in this many row bands
    if fault:
        env["OTMB_TEST_FAULT"] = fault
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), str(r), str(world), str(port), kind,
                               *[str(x) for x in case], str(outdir)], env=env) for r in range(world)]
    rcs = [p.wait(timeout=timeout) for p in procs]
    assert rcs == [0] * world, rcs
    return np.load(os.path.join(outdir, "global.npz"))


def check_against_whole_grid(oracle, z, case, upwind=True):
    idx, ref = whole_grid_reference(oracle, *case, upwind=upwind)
    assert int(z["n"]) == idx["N"]
    for q, m in enumerate(MATS):
        got = (z[f"0_{q}"], z[f"1_{q}"], z[f"2_{q}"])
        assert_csc_equal(got, ref[m], m)


def test_balanced_partition_properties():
    rng = np.random.default_rng(0)
    for _ in range(50):
        nz = int(rng.integers(1, 40))
        world = int(rng.integers(1, min(nz, 8) + 1))
        counts = rng.integers(0, 1000, nz)
        parts = od.balanced_partition(counts, world)
        assert parts[0][0] == 0 and parts[-1][1] == nz and len(parts) == world
        assert all(a < b for a, b in parts) and all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))


@pytest.mark.parametrize("world,rho", [(2, "array"), (3, "scalar"), (4, "array")])
def test_slab_orchestration_gloo(oracle, tmp_path, world, rho):
    case = (12, 10, 9, 21, rho, "tripolar")
    z = run_ranks(world, "oracle", case, tmp_path)
    check_against_whole_grid(oracle, z, case)


def test_slab_async_pipeline_gloo(oracle, tmp_path):
    """step_async x3 + finish: no collective per field, local colptrs shifted to global ones at the end."""
    case = (12, 10, 9, 22, "array", "tripolar")
    z = run_ranks(3, "oracle", case, tmp_path, async_mode=True)
    check_against_whole_grid(oracle, z, case)


@pytest.mark.parametrize("pieces,world,async_mode", [(2, 2, False), (5, 3, False), (10, 3, True), (3, 4, True)])
def test_facefluxes_chain_in_row_bands_gloo(oracle, tmp_path, pieces, world, async_mode):
    """slab s piece c waits for slab s + 1 piece c only (whole rows per piece; 10 = one row per piece here).  The checker
    backend computes a piece with whatever the plane buffer holds, so a piece used before it arrived would give a wrong matrix."""
    case = (12, 10, 9, 23, "array", "tripolar")
    z = run_ranks(world, "oracle", case, tmp_path, async_mode=async_mode, pieces=pieces)
    check_against_whole_grid(oracle, z, case)


def check_fault_reports(outdir, world, exc, step, text):
    for r in range(world):
        name, st, msg = open(os.path.join(outdir, f"fault_{r}.txt")).read().split("|", 2)
        assert name == exc and text in msg and f"step {step + 1} of 5" in msg, (r, name, st, msg)
        if exc == "OtmbError":
            assert int(st) == step


@pytest.mark.parametrize("fault,exc,text", [("1:1:rho", "OtmbError", "ρ contains NaNs"), ("2:4:rho", "OtmbError", "ρ contains NaNs"),
                                            ("0:0:rho", "OtmbError", "ρ contains NaNs")])
def test_slab_pipeline_reports_first_failing_step_on_every_rank(oracle, tmp_path, fault, exc, text):
    """A failure in step k of 5 asynchronous steps (a NaN in ρ in ONE slab) is raised by finish() on EVERY rank with the
    step's index -- nobody hangs in a collective -- and the result afterwards is right.  (The facefluxes assertion,
    velocities.jl:199-200, cannot fire on a grid with land: nofluxboundaries! zeroes those faces first; its per-step
    bookkeeping is tested on an all-wet grid in tests/test_pipeline_errors.py.)"""
    case = (12, 10, 9, 23, "array", "tripolar")
    z = run_ranks(3, "oracle", case, tmp_path, fault=fault)
    check_fault_reports(tmp_path, 3, exc, int(fault.split(":")[1]), text)
    check_against_whole_grid(oracle, z, case)


def test_slab_orchestration_gloo_world8_on_75_levels(oracle, tmp_path):
    """BASELINE.json configs[3]'s shape in small: 75 levels cut into 8 depth slabs."""
    case = (10, 8, 75, 24, "array", "tripolar")
    z = run_ranks(8, "oracle", case, tmp_path, async_mode=True)
    check_against_whole_grid(oracle, z, case)
