"""examples/otmb_c_example.c: the C ABI called from plain C (gcc, no Python, no shim) -- include/otmb.h must be valid C, the library must link
with nothing but itself, and the program's matrices must be the ones the Python host layer builds from the same inputs."""
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "lib")


def _build(tmp_path, extra=()):
    exe = str(tmp_path / "otmb_c_example")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", *extra, "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "otmb_c_example.c"), "-L", LIBDIR, "-lotmb_hip", f"-Wl,-rpath,{LIBDIR}", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_the_example_links(tmp_path):
    """No GPU needed: gcc -std=c99 -pedantic -Werror accepts include/otmb.h, and the program links against libotmb_hip.so alone."""
    if not os.path.exists(os.path.join(LIBDIR, "libotmb_hip.so")):
        pytest.skip("library not built")
    _build(tmp_path)


@pytest.mark.gpu
def test_c_program_builds_the_same_matrices_as_the_python_layer(tmp_path):
    import otmb_amd.api as api

    nx, ny, nz = 12, 8, 5
    exe = _build(tmp_path)
    r = subprocess.run([exe, str(nx), str(ny), str(nz)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    out = dict(kv.split("=") for kv in r.stdout.split())
    # the second time slice: the two grid-constant operators passed back -- derived (state 1), not built (nnz 0), T the same bit for bit
    assert out["given_state"] == "1,1" and out["T_same"] == "1"
    first, second = [int(x) for x in out["nnz"].split(",")], [int(x) for x in out["nnz2"].split(",")]
    assert second == [first[0], first[1], 0, first[3], 0]
    # the same inputs in numpy (the C program's formulas)
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    dx = 111e3 * np.cos((np.arange(ny) - ny / 2.0) * 0.01)
    dx2 = np.asfortranarray(np.broadcast_to(dx[None, :], (nx, ny)).copy())
    dy2 = np.asfortranarray(np.full((nx, ny), 111e3))
    area = np.asfortranarray(dx2 * 111e3)
    land = (i == 2) | ((k == nz - 1) & ((i + j) % 3 == 0))
    v3d = np.asfortranarray(np.where(land, np.nan, (area * 10.0)[:, :, None]))
    thk = np.asfortranarray(np.full((nx, ny, nz), 10.0))
    rho = np.asfortranarray(1025.0 + 0.01 * k + 0.001 * i)
    umo = np.asfortranarray(1e6 * np.sin(0.3 * i + 0.2 * j + 0.1 * k))
    vmo = np.asfortranarray(1e6 * np.cos(0.2 * i - 0.3 * j + 0.2 * k))
    ds = dy2.copy(order="F")
    ds[:, 0] = np.nan
    ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    ml = np.asfortranarray(12.0 + 3.0 * ((ii + jj) % 4))
    gm = dict(v3D=v3d, thkcello=thk, area2D=area, zt=5.0 + 10.0 * np.arange(nz), gridtopology=dict(kind=1),
              edge_length_2D=dict(west=dy2, east=dy2, south=dx2, north=dx2),
              distance_to_neighbour_2D=dict(west=dx2, east=dx2, south=ds, north=dy2))
    idx = api.makeindices(v3d)
    phi = api.facefluxes(umo, vmo, gm, idx, FillValue=1e20)
    tm = api.transportmatrix(ϕ=phi, mlotst=ml, gridmetrics=gm, indices=idx, ρ=rho)
    assert int(out["N"]) == idx.N
    assert [int(x) for x in out["nnz"].split(",")] == [tm[m].nnz for m in ("T", "Tadv", "TκH", "TκVML", "TκVdeep")]
    assert int(out["colptrT_last"]) == tm["T"].nnz + 1
    # libm's sin / cos in the C program and numpy's may differ in the last ulp of umo / vmo: the sums agree to 1e-9 relative
    # (the PATTERN above is exact: signs of the fluxes, wet mask and mixed-layer mask)
    seq = 0.0
    for x in tm["T"].nzval.tolist():
        seq += x
    sabs = math.fsum(abs(x) for x in tm["T"].nzval.tolist())
    assert abs(float(out["sumabsT"]) - sabs) <= 1e-9 * sabs
    assert abs(float(out["sumT"]) - seq) <= 1e-9 * sabs
