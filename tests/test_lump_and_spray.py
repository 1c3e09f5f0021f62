"""lump_and_spray (src/extratools.jl:38-119) on the GPU against the oracle: the coarse row of every wet cell, SPRAY's
structure and the coarse volumes bit for bit, LUMP's values bit for bit."""
import numpy as np
import pytest

from helpers import CASES, MATS, make_case
from test_oracle import LUMP_SETTINGS, lump_inputs, lump_mask

pytestmark = pytest.mark.gpu


def _same(got, want, what):
    (L, S, vc), (rL, rS, rvc) = got, want
    for a, b, nm in ((L.colptr, rL[0], "LUMP.colptr"), (L.rowval, rL[1], "LUMP.rowval"), (S.colptr, rS[0], "SPRAY.colptr"),
                     (S.rowval, rS[1], "SPRAY.rowval")):
        assert np.array_equal(a, b), (what, nm, np.flatnonzero(np.asarray(a) != np.asarray(b))[:5])
    for a, b, nm in ((L.nzval, rL[2], "LUMP.nzval"), (S.nzval, rS[2], "SPRAY.nzval"), (vc, rvc, "vol_c")):
        assert np.array_equal(a, b), (what, nm, np.flatnonzero(np.asarray(a) != np.asarray(b))[:5])
    assert L.shape == (len(rvc), len(rL[1])) and S.shape == (len(rL[1]), len(rvc))


@pytest.mark.parametrize("name", ["tiny_tripolar", "tiny_bipolar", "odd_nx_fold", "small_rho3d", "nx2"])
def test_lump_and_spray_matches_oracle(oracle, name):
    import otmb_amd.api as api

    wet, vol, tm, N = lump_inputs(oracle, name)
    T = api.SparseMatrixCSC(N, N, *tm["T"])
    for q, (di, dj, dk, usemask) in enumerate(LUMP_SETTINGS + [(1, 7, 2, True), (5, 1, 4, False), (3, 3, 3, True)]):
        mask = lump_mask(wet, q) if usemask else None
        want = oracle.lump_and_spray(wet, vol, tm["T"], mask, di, dj, dk)
        got = api.lump_and_spray(wet, vol, T, mask, di=di, dj=dj, dk=dk)
        _same(got, want, (name, di, dj, dk, usemask))


def test_lump_and_spray_structured_masks(oracle):
    """Masks like the reference's tests use (column-wise regions, test/online.jl:126-129) and adversarial ones: single
    cells, stripes that shift the block alignment from row to row, everything outside."""
    import otmb_amd.api as api

    wet, vol, tm, N = lump_inputs(oracle, "small_rho3d")
    T = api.SparseMatrixCSC(N, N, *tm["T"])
    nx, ny, nz = wet.shape
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    masks = {
        "columns": np.repeat(((jj[:, :, :1] > ny // 4) & ~((ii[:, :, :1] > nx // 2) & (jj[:, :, :1] > 2 * ny // 3))), nz, axis=2),
        "diagonal_stripes": ((ii + 2 * jj + kk) % 5) != 0,
        "checkerboard": ((ii + jj + kk) % 2) == 0,
        "none": np.zeros(wet.shape, dtype=bool),
        "one_cell_holes": ~((ii % 7 == 3) & (jj % 5 == 1)),
    }
    for nm, mask in masks.items():
        for (di, dj, dk) in ((2, 2, 1), (3, 4, 2), (10, 10, 1)):
            want = oracle.lump_and_spray(wet, vol, tm["T"], mask, di, dj, dk)
            got = api.lump_and_spray(wet, vol, T, mask, di=di, dj=dj, dk=dk)
            _same(got, want, (nm, di, dj, dk))


def test_lump_and_spray_errors_and_operator_identities(oracle):
    import otmb_amd.api as api
    from otmb_amd.capi import OtmbError

    wet, vol, tm, N = lump_inputs(oracle, "tiny_tripolar")
    Tadv = api.SparseMatrixCSC(N, N, *tm["Tadv"])
    with pytest.raises(OtmbError, match="symmetric") as e:
        api.lump_and_spray(wet, vol, Tadv)
    assert e.value.name == "ASYMMETRIC_PATTERN"
    T = api.SparseMatrixCSC(N, N, *tm["T"])
    with pytest.raises(OtmbError):
        api.lump_and_spray(wet, vol, T, di=100, dj=100, dk=1)  # block larger than the 4096-cell limit
    LUMP, SPRAY, vol_c = api.lump_and_spray(wet, vol, T)  # defaults di = dj = 2, dk = 1
    Lm, Sm = LUMP.to_scipy(), SPRAY.to_scipy()
    Nc = len(vol_c)
    assert abs(Lm @ Sm - np.eye(Nc)).max() < 1e-14          # lumping a sprayed field gives it back
    assert np.allclose(Lm @ np.ones(N), 1.0)                 # a coarse value is a volume-weighted MEAN
    assert np.allclose(Lm.T @ vol_c, vol)                    # with weights vol / vol_c
    assert np.allclose(Sm @ np.ones(Nc), 1.0)                # every fine cell belongs to exactly one coarse cell
    Tc = Lm @ T.to_scipy() @ Sm                               # the coarse operator of the reference's docstring
    assert Tc.shape == (Nc, Nc)


def test_device_lump_and_spray_on_resident_result(oracle):
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm = make_case("small_rho3d")
    wet, vol, tm, N = lump_inputs(oracle, "small_rho3d")
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    asm.step(umo, vmo, g.umo.properties["_FillValue"])
    mask = lump_mask(wet, 5)
    for m, (di, dj, dk) in ((None, (2, 2, 1)), (mask, (4, 3, 2))):
        want = oracle.lump_and_spray(wet, vol, tm["T"], m, di, dj, dk)
        dm = None if m is None else torch.from_numpy(np.asfortranarray(m).ravel(order="F").astype(np.uint8)).cuda()
        L, S, vc = asm.lump_and_spray(dm, di, dj, dk)
        assert np.array_equal(L[1].cpu().numpy(), want[0][1]) and np.array_equal(L[2].cpu().numpy(), want[0][2])
        assert np.array_equal(L[0].cpu().numpy(), want[0][0])
        assert np.array_equal(S[0].cpu().numpy(), want[1][0]) and np.array_equal(S[1].cpu().numpy(), want[1][1])
        assert np.array_equal(S[2].cpu().numpy(), want[1][2]) and np.array_equal(vc.cpu().numpy(), want[2])


@pytest.mark.parametrize("seed", range(30))
def test_lump_and_spray_random_sweep(oracle, seed):
    """Random small grids, masks, block sizes (including blocks larger than the grid and dk > 1) and synthetic symmetric
    connectivity patterns: the device result must equal the oracle's, or both must refuse an asymmetric pattern."""
    import otmb_amd.api as api
    from otmb_amd.capi import OtmbError

    rng = np.random.default_rng(7000 + seed)
    nx, ny, nz = int(rng.integers(1, 14)), int(rng.integers(1, 9)), int(rng.integers(1, 6))
    wet = rng.random((nx, ny, nz)) < rng.uniform(0.3, 1.0)
    if not wet.any():
        wet[0, 0, 0] = True
    N = int(wet.sum())
    rank = np.zeros(wet.shape, dtype=np.int64)
    rank.reshape(-1, order="F")[wet.reshape(-1, order="F")] = np.arange(1, N + 1)
    # pattern: diagonal + a random subset of the 6-neighbour links (periodic in i), symmetric unless `broken`
    broken = seed % 7 == 3
    cols = [set([c]) for c in range(N + 1)]
    for (a, b, c) in np.argwhere(wet):
        for (da, db, dc) in ((1, 0, 0), (0, 1, 0), (0, 0, 1)):
            a2, b2, c2 = (a + da) % nx, b + db, c + dc
            if b2 < ny and c2 < nz and wet[a2, b2, c2] and rng.random() < 0.7:
                r1, r2 = int(rank[a, b, c]), int(rank[a2, b2, c2])
                cols[r1].add(r2)
                if not (broken and rng.random() < 0.3):
                    cols[r2].add(r1)
    colptr = np.ones(N + 1, dtype=np.int64)
    rowval = []
    for c in range(1, N + 1):
        rows = sorted(cols[c])
        rowval.extend(rows)
        colptr[c] = colptr[c - 1] + len(rows)
    rowval = np.array(rowval, dtype=np.int64)
    T = (colptr, rowval, np.ones(len(rowval)))
    vol = rng.uniform(1.0, 1e6, N)
    di, dj, dk = int(rng.integers(1, 6)), int(rng.integers(1, 5)), int(rng.integers(1, 4))
    mask = None if rng.random() < 0.3 else (rng.random(wet.shape) < rng.uniform(0.2, 1.0))
    Tm = api.SparseMatrixCSC(N, N, *T)
    try:
        want = oracle.lump_and_spray(wet, vol, T, mask, di, dj, dk)
    except oracle.OracleError:
        with pytest.raises(OtmbError) as e:
            api.lump_and_spray(wet, vol, Tm, mask, di=di, dj=dj, dk=dk)
        assert e.value.name == "ASYMMETRIC_PATTERN"
        return
    got = api.lump_and_spray(wet, vol, Tm, mask, di=di, dj=dj, dk=dk)
    _same(got, want, (seed, nx, ny, nz, di, dj, dk))


def test_lump_and_spray_edge_grids(oracle):
    import otmb_amd.api as api

    # one wet cell; a single column; blocks larger than the grid; everything dry except one level
    cases = []
    w = np.zeros((4, 3, 2), dtype=bool); w[1, 1, 0] = True
    cases.append((w, (2, 2, 1)))
    w = np.zeros((1, 1, 5), dtype=bool); w[0, 0, :3] = True
    cases.append((w, (1, 1, 2)))
    w = np.ones((3, 2, 2), dtype=bool)
    cases.append((w, (8, 8, 4)))
    w = np.zeros((6, 5, 3), dtype=bool); w[:, :, 1] = True
    cases.append((w, (2, 3, 3)))
    for wet, (di, dj, dk) in cases:
        N = int(wet.sum())
        rank = np.zeros(wet.shape, dtype=np.int64)
        rank.reshape(-1, order="F")[wet.reshape(-1, order="F")] = np.arange(1, N + 1)
        cols = [set([c]) for c in range(N + 1)]
        for (a, b, c) in np.argwhere(wet):
            for (da, db, dc) in ((1, 0, 0), (0, 1, 0), (0, 0, 1)):
                a2, b2, c2 = a + da, b + db, c + dc
                if a2 < wet.shape[0] and b2 < wet.shape[1] and c2 < wet.shape[2] and wet[a2, b2, c2]:
                    cols[int(rank[a, b, c])].add(int(rank[a2, b2, c2])); cols[int(rank[a2, b2, c2])].add(int(rank[a, b, c]))
        colptr = np.ones(N + 1, dtype=np.int64); rowval = []
        for c in range(1, N + 1):
            rowval.extend(sorted(cols[c])); colptr[c] = colptr[c - 1] + len(cols[c])
        T = (colptr, np.array(rowval, dtype=np.int64), np.ones(len(rowval)))
        vol = np.linspace(1.0, 2.0, N)
        for mask in (None, np.zeros(wet.shape, dtype=bool)):
            want = oracle.lump_and_spray(wet, vol, T, mask, di, dj, dk)
            got = api.lump_and_spray(wet, vol, api.SparseMatrixCSC(N, N, *T), mask, di=di, dj=dj, dk=dk)
            _same(got, want, (wet.shape, di, dj, dk, mask is None))
            if mask is not None:
                assert len(got[2]) == N  # nothing is lumped outside the mask


def _lump_fixture_paths():
    from test_oracle import lump_fixtures

    return lump_fixtures()


@pytest.mark.parametrize("path", _lump_fixture_paths(), ids=lambda p: p.split("/")[-1][:-4])
def test_hip_reproduces_lump_fixtures(path):
    import otmb_amd.api as api

    z = np.load(path)
    di, dj, dk = (int(x) for x in z["block"])
    N = len(z["vol"])
    T = api.SparseMatrixCSC(N, N, z["T_colptr"], z["T_rowval"], np.ones(len(z["T_rowval"])))
    LUMP, SPRAY, vol_c = api.lump_and_spray(z["wet3D"], z["vol"], T, z["mask"], di=di, dj=dj, dk=dk)
    assert np.array_equal(LUMP.rowval, z["lump_rowval"]) and np.array_equal(LUMP.nzval, z["lump_nzval"])
    assert np.array_equal(SPRAY.colptr, z["spray_colptr"]) and np.array_equal(SPRAY.rowval, z["spray_rowval"])
    assert np.array_equal(vol_c, z["vol_c"])
