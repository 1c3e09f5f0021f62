"""tests/golden/known_answer: a 4 x 2 x 2 tripolar grid made of small fractions, every triplet listed with the line of
src/matrixbuilding.jl that pushes it and the arithmetic behind it (KNOWN_ANSWER.md) so that a reader can pin the oracle
by eye.  The C oracle, the Python transliteration and the HIP library must all reproduce known_answer.json bit for bit."""
import json
import os

import numpy as np
import pytest

from helpers import MATS, assert_csc_equal

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    d = json.load(open(os.path.join(HERE, "golden", "known_answer", "known_answer.json"), encoding="utf-8"))
    nx, ny, nz = d["shape"]

    def a3(x):
        return np.asfortranarray(np.array([np.nan if v is None else v for v in x], dtype=np.float64).reshape((nx, ny, nz), order="F"))

    def a2(x):
        return np.asfortranarray(np.array([np.nan if v is None else v for v in x], dtype=np.float64).reshape((nx, ny), order="F"))

    gm = dict(v3D=a3(d["v3D"]), thkcello=a3(d["thkcello"]), area2D=a2(d["area2D"]), zt=np.array(d["zt"]), gridtopology=dict(kind=1),
              edge_length_2D={k: a2(v) for k, v in d["edge_length_2D"].items()},
              distance_to_neighbour_2D={k: a2(v) for k, v in d["distance_to_neighbour_2D"].items()})
    mats = {m: (np.array(d["matrices"][m]["colptr"], np.int64), np.array(d["matrices"][m]["rowval"], np.int64),
                np.array(d["matrices"][m]["nzval"], np.float64)) for m in MATS}
    phi = {k: a3(v) for k, v in d["phi"].items()}
    return d, gm, a3(d["umo"]), a3(d["vmo"]), a2(d["mlotst"]), phi, mats


def test_fixture_is_what_its_generator_writes(tmp_path):
    """The committed JSON is reproducible: the generator (which imports neither oracle/ nor the product) rewrites it identically."""
    import shutil
    import subprocess
    import sys

    p = os.path.join(HERE, "golden", "known_answer")
    src = open(os.path.join(p, "make_known_answer.py"), encoding="utf-8").read()
    assert "import oracle" not in src and "from oracle" not in src and "otmb_amd" not in src
    shutil.copy(os.path.join(p, "make_known_answer.py"), tmp_path / "make_known_answer.py")
    subprocess.check_call([sys.executable, str(tmp_path / "make_known_answer.py")], stdout=subprocess.DEVNULL)
    for f in ("known_answer.json", "KNOWN_ANSWER.md"):
        assert open(tmp_path / f, encoding="utf-8").read() == open(os.path.join(p, f), encoding="utf-8").read(), f


def test_oracle_and_transliteration_reproduce_the_known_answer(oracle):
    from oracle import pyref

    d, gm, umo, vmo, ml, phi, mats = load()
    idx = oracle.makeindices(gm["v3D"])
    assert idx["N"] == d["N"]
    for cell, w in d["wet_rank"].items():
        i, j, k = (int(x) for x in cell.split(","))
        assert idx["Lwet3D"][i - 1, j - 1, k - 1] == w
    got_phi = oracle.facefluxes(umo, vmo, idx["wet3D"], d["fill"], 1)
    for k in phi:
        assert np.array_equal(got_phi[k], phi[k]), k
    kH, kML, kD = d["kappa"]
    tm = oracle.transportmatrix(phi, gm, idx, d["rho"], ml, kH, kML, kD, True)
    for m in MATS:
        assert_csc_equal(tm[m], mats[m], m)
    # the triplets themselves, in emission order
    topo = pyref.Topo(1, *d["shape"])
    pidx = pyref.makeindices(gm["v3D"])
    I, J, V = pyref.advection_entries(phi, gm["v3D"], d["rho"], pidx, topo, True)
    want = d["triplets"]["Tadv"]
    assert [int(x) for x in I] == [t["row"] for t in want] and [int(x) for x in J] == [t["col"] for t in want]
    assert [float(x) for x in V] == [t["val"] for t in want]
    ptm = pyref.transportmatrix(phi, gm, pidx, topo, d["rho"], ml, kH, kML, kD, True)
    for m in MATS:
        assert_csc_equal(ptm[m], mats[m], "pyref " + m)


@pytest.mark.gpu
def test_hip_reproduces_the_known_answer():
    import otmb_amd.api as api

    d, gm, umo, vmo, ml, phi, mats = load()
    idx = api.makeindices(gm["v3D"])
    assert idx.N == d["N"]
    got_phi = api.facefluxes(umo, vmo, gm, idx, FillValue=d["fill"])
    for k in phi:
        assert np.array_equal(got_phi[k], phi[k]), k
    kH, kML, kD = d["kappa"]
    tm = api.transportmatrix(ϕ=got_phi, mlotst=ml, gridmetrics=gm, indices=idx, ρ=d["rho"], κH=kH, κVML=kML, κVdeep=kD)
    for m in MATS:
        assert_csc_equal(tuple(tm[m]), mats[m], m)
