"""The reference's own two-step formulation on the device: COO generators in emission order, then
sparse(I,J,V,N,N) -- bit-exact against the oracle's generators / sparse(), and equal to the fused kernel."""
import numpy as np
import pytest

from helpers import MATS, assert_csc_equal, make_case

pytestmark = pytest.mark.gpu


def _setup(oracle, name):
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    asm.facefluxes(umo, vmo, fill)
    return g, gm, ref, rphi, asm


@pytest.mark.parametrize("name", ["tiny_tripolar", "tiny_rho3d", "odd_nx_fold", "nx2", "tiny_bipolar", "small_rho3d"])
def test_coo_generators_and_sparse_match_oracle(oracle, name):
    g, gm, ref, rphi, asm = _setup(oracle, name)
    kind, N = gm.gridtopology.kind, ref["N"]
    Om = oracle.ml_mask(gm.zt, g.mlotst, ref["Lwet"], gm.v3D.shape)
    want = {
        "Tadv": oracle.advection_entries(rphi, gm.v3D, g.rho, ref["Lwet"], ref["Lwet3D"], kind, True),
        "TκH": oracle.hdiff_entries(gm.v3D, gm.thkcello, gm.edge_length_2D, gm.distance_to_neighbour_2D, ref["Lwet"], ref["Lwet3D"], kind, g.kappaH),
        "TκVML": oracle.vdiff_entries(gm.v3D, gm.area2D, gm.zt, ref["Lwet"], ref["Lwet3D"], kind, g.kappaVML, Om),
        "TκVdeep": oracle.vdiff_entries(gm.v3D, gm.area2D, gm.zt, ref["Lwet"], ref["Lwet3D"], kind, g.kappaVdeep, None),
    }
    asm.step(None, None, None) if False else None
    fused = None
    for m in ("Tadv", "TκH", "TκVML", "TκVdeep"):
        I, J, V = asm.sparse_entries(m)
        wi, wj, wv = want[m]
        assert np.array_equal(I.cpu().numpy(), wi) and np.array_equal(J.cpu().numpy(), wj), m  # emission order
        assert np.array_equal(V.cpu().numpy(), wv), m
        cp, rv, nz = asm.sparse(I, J, V, N, N)
        got = (cp.cpu().numpy(), rv.cpu().numpy(), nz.cpu().numpy())
        assert_csc_equal(got, oracle.sparse(wi, wj, wv, N, N), f"{name}/{m} sparse()")
        if fused is None:
            asm.transportmatrix(asm.phi)
            fused = asm.result_to_host()
        assert_csc_equal(got, fused[m], f"{name}/{m} general path == fused kernel")


def test_device_sparse_on_arbitrary_triplets(oracle):
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm, ref, rphi, asm = _setup(oracle, "tiny_tripolar")
    rng = np.random.default_rng(5)
    m, n, ln = 300, 257, 20000
    I = rng.integers(1, m + 1, ln); J = rng.integers(1, n + 1, ln); V = rng.standard_normal(ln)
    V[::11] = 0.0
    J[J == 17] = 18  # an empty column in the middle; the last columns may be empty as well
    J[J > 250] = 250
    cp, rv, nz = asm.sparse(torch.from_numpy(I).cuda(), torch.from_numpy(J).cuda(), torch.from_numpy(V).cuda(), m, n)
    assert_csc_equal((cp.cpu().numpy(), rv.cpu().numpy(), nz.cpu().numpy()), oracle.sparse(I, J, V, m, n), "random COO")
    e = torch.empty(0, dtype=torch.int64, device="cuda")
    cp, rv, nz = asm.sparse(e, e, torch.empty(0, dtype=torch.float64, device="cuda"), 5, 5)
    assert cp.cpu().tolist() == [1] * 6 and rv.numel() == 0
    # indices outside 1..m / 1..n: SparseArrays throws an ArgumentError; nothing may be written out of bounds
    from otmb_amd.capi import OtmbError

    for bi, bj in [(0, 3), (m + 1, 3), (5, 0), (5, n + 1), (-7, 3), (5, 1 << 40)]:
        I2, J2 = I.copy(), J.copy()
        I2[123], J2[123] = bi, bj
        with pytest.raises(OtmbError, match="ArgumentError") as ei:
            asm.sparse(torch.from_numpy(I2).cuda(), torch.from_numpy(J2).cuda(), torch.from_numpy(V).cuda(), m, n)
        assert ei.value.name == "INVALID_ARG"
    cp, rv, nz = asm.sparse(torch.from_numpy(I).cuda(), torch.from_numpy(J).cuda(), torch.from_numpy(V).cuda(), m, n)
    assert_csc_equal((cp.cpu().numpy(), rv.cpu().numpy(), nz.cpu().numpy()), oracle.sparse(I, J, V, m, n), "after the failures")


def test_device_sparse_with_long_runs_of_empty_columns(oracle):
    """sparse(I, J, V, m, n) whose entries sit in a few columns of a wide matrix (TκVML: the mixed layer only): colptr over a run of empty columns
    is filled by the entry after it -- short runs by its thread, long ones (> 16 columns, in pieces of 65 536) through a list and whole workgroups
    (one thread used to walk them all: 26 ms of a 27 ms call at 1 degree).  Leading, inner and trailing runs of every kind, against scipy's CSC."""
    import scipy.sparse as sp
    import torch

    g, gm, ref, rphi, asm = _setup(oracle, "tiny_tripolar")
    rng = np.random.default_rng(9)
    m, n = 1000, 400000
    cols = np.concatenate([[40, 41, 42, 58, 59, 75, 76], np.arange(100, 140), [70000, 70001, 135537, 135538, 201074, 399000]])  # runs of 39, 15, 16, 17, ~70 k, 65 536, 65 535, ...
    J = np.repeat(cols, 3).astype(np.int64)
    I = rng.integers(1, m + 1, J.size).astype(np.int64)
    V = rng.standard_normal(J.size)
    perm = rng.permutation(J.size)
    I, J, V = I[perm], J[perm], V[perm]
    cp, rv, nz = asm.sparse(torch.from_numpy(I).cuda(), torch.from_numpy(J).cuda(), torch.from_numpy(V).cuda(), m, n)
    want = sp.coo_matrix((V, (I - 1, J - 1)), shape=(m, n)).tocsc()
    want.sum_duplicates()
    want.sort_indices()
    assert np.array_equal(cp.cpu().numpy(), want.indptr.astype(np.int64) + 1)
    assert np.array_equal(rv.cpu().numpy(), want.indices.astype(np.int64) + 1)
    assert np.allclose(nz.cpu().numpy(), want.data, rtol=1e-15, atol=0)
    assert_csc_equal((cp.cpu().numpy(), rv.cpu().numpy(), nz.cpu().numpy()), oracle.sparse(I, J, V, m, n), "wide matrix, few columns")
    # one entry in the last column; one entry in the first column; nothing at all
    for jj in (n, 1):
        cp, rv, nz = asm.sparse(torch.tensor([7], device="cuda"), torch.tensor([jj], device="cuda"), torch.tensor([2.5], dtype=torch.float64, device="cuda"), m, n)
        c = cp.cpu().numpy()
        assert (c[:jj] == 1).all() and (c[jj:] == 2).all() and rv.cpu().tolist() == [7]
    e = torch.empty(0, dtype=torch.int64, device="cuda")
    cp, rv, nz = asm.sparse(e, e, torch.empty(0, dtype=torch.float64, device="cuda"), m, n)
    assert (cp.cpu().numpy() == 1).all() and cp.numel() == n + 1


@pytest.mark.parametrize("name", ["tiny_tripolar", "small_rho3d", "odd_nx_fold"])
def test_build_single_operators_like_the_reference(oracle, name):
    """buildTadv / buildTκH / buildTκVML / buildTκVdeep (src/matrixbuilding.jl:31-120) one at a time equal the operators
    transportmatrix returns."""
    import otmb_amd
    from helpers import MATS, assert_csc_equal, make_case

    g, gm = make_case(name)
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    idx = otmb_amd.makeindices(gm.v3D)
    got = {
        "Tadv": otmb_amd.buildTadv(ϕ=rphi, gridmetrics=gm, indices=idx, ρ=g.rho),
        "TκH": otmb_amd.buildTκH(gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH),
        "TκVML": otmb_amd.buildTκVML(mlotst=g.mlotst, gridmetrics=gm, indices=idx, κVML=g.kappaVML),
        "TκVdeep": otmb_amd.buildTκVdeep(mlotst=g.mlotst, gridmetrics=gm, indices=idx, κVdeep=g.kappaVdeep),
    }
    for m in MATS[1:]:
        assert_csc_equal(tuple(got[m]), rtm[m], f"{name}/{m}")
