"""transportmatrix(...; Tadv = ..., TκH = ..., TκVML = ..., TκVdeep = ...) -- src/matrixbuilding.jl:133-147: an operator that is passed in is NOT
built, is returned as the very object passed, and T = ((Tadv + TκH) + TκVML) + TκVdeep is formed with it (run with -m gpu).

The library's three ways (otmb_tm_args.given; otmb_ctx_given_state):
  derived (1) -- a given TκH / TκVdeep that is bit for bit what the fill pass computes for this grid and κ (one comparing pass, verdict cached):
             neither counted, stored nor copied home;
  derived rows, other values (3) -- built with another κ, or edited: treated alike, and the fill pass READS its values where they lie;
  foreign (2) -- any other matrix (another pattern, Tadv, TκVML): the built operators are written and T is the device sparse add of the
             four operands (two-phase protocol).
Every combination is compared with the oracle: the built matrices against orc_transportmatrix, T against the left fold of orc_spadd over the
operands actually used -- the GIVEN values, not re-derived ones."""
import itertools

import numpy as np
import pytest

from helpers import COUNTS_ON, CASES, MATS, assert_csc_equal, gridmetrics_of, make_case

pytestmark = pytest.mark.gpu
OPS = MATS[1:]


def _fold(oracle, ops, N):
    """T = ((Tadv + TκH) + TκVML) + TκVdeep with SparseArrays' `+` (orc_spadd: union pattern, exact zeros dropped), :147."""
    return oracle.spadd(oracle.spadd(oracle.spadd(ops["Tadv"], ops["TκH"], N), ops["TκVML"], N), ops["TκVdeep"], N)


def _setup(oracle, name_or_grid, upwind=True, kappa=None):
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm = make_case(name_or_grid) if isinstance(name_or_grid, str) else name_or_grid
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    kap = kappa or (g.kappaH, g.kappaVML, g.kappaVdeep)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, *kap, upwind)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, *kap, upwind=upwind)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    return g, gm, ref, rphi, rtm, asm, umo, vmo, fill


def _dev(asm, csc):
    import torch

    return tuple(torch.from_numpy(np.ascontiguousarray(x)).to(asm.device) for x in csc)


def _checks(asm):
    return int(asm.lib.otmb_ctx_given_checks(asm.ctx.handle))


@pytest.mark.parametrize("name", ["tiny_tripolar", "tiny_rho3d", "odd_nx_fold", "nx2", "even_fold_open", "small_rho3d"])
@pytest.mark.parametrize("upwind", [True, False])
def test_every_subset_of_given_operators_two_phase(oracle, name, upwind):
    """1, 2, 3 and 4 operators given (all 15 subsets), the given matrices being the oracle's own: TκH / TκVdeep are then DERIVED (state 1),
    Tadv / TκVML FOREIGN (state 2).  The built matrices are the oracle's, T is the oracle's fold over the operands, nothing of a given
    operator is written, and its nnz comes back 0."""
    import torch

    g, gm, ref, rphi, rtm, asm, umo, vmo, fill = _setup(oracle, name, upwind)
    N = ref["N"]
    want_T = _fold(oracle, rtm, N)
    assert_csc_equal(want_T, rtm["T"], "the oracle's own fold")
    for r in range(1, 5):
        for sub in itertools.combinations(OPS, r):
            asm.set_given(**{m: (_dev(asm, rtm[m]) if m in sub else None) for m in OPS})
            phi = asm.facefluxes(umo, vmo, fill)
            asm.out = None
            asm.transportmatrix(phi)
            states = [asm.ctx.given_state(k) for k in range(5)]
            assert states == [0] + [(1 if m in ("TκH", "TκVdeep") else 2) if m in sub else 0 for m in OPS], (sub, states)
            got = asm.result_to_host()
            assert_csc_equal(got["T"], rtm["T"], f"{sub}: T")
            for k, m in enumerate(MATS):
                if m in sub:
                    assert asm.nnz[k] == 0, (sub, m)
                elif m != "T":
                    assert_csc_equal(got[m], rtm[m], f"{sub}: {m}")
    asm.set_given(**{m: None for m in OPS})
    asm.step(umo, vmo, fill)
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], f"after the given operators were taken back: {m}")
    del torch


@pytest.mark.parametrize("name", ["tiny_tripolar", "odd_nx_fold", "small_rho3d", "float32_flux"])
def test_derived_operators_in_the_asynchronous_protocols(oracle, name):
    """TκH / TκVdeep / both given and derived: otmb_transportmatrix_dev (count -> scan -> fill with the given fields masked out of the counts;
    counts from facefluxes and from the counting pass), a pipeline of steps, and the fused step.  Tadv / TκVML given: GIVEN_FOREIGN there."""
    from otmb_amd.capi import OtmbError

    g, gm, ref, rphi, rtm, asm, umo, vmo, fill = _setup(oracle, name)
    for sub in (("TκH",), ("TκVdeep",), ("TκH", "TκVdeep")):
        asm.set_given(**{m: (_dev(asm, rtm[m]) if m in sub else None) for m in OPS})
        for count_in_ff in (True, False):
            asm.count_in_ff = count_in_ff
            asm.step(umo, vmo, fill)
            got = asm.result_to_host()
            for k, m in enumerate(MATS):
                if m in sub:
                    assert asm.nnz[k] == 0
                else:
                    assert_csc_equal(got[m], rtm[m], f"{sub} async count_in_ff={count_in_ff}: {m}")
        asm.count_in_ff = True
        for _ in range(3):
            asm.step_async(umo, vmo, fill)
        asm.finish()
        got = asm.result_to_host()
        for m in MATS:
            if m not in sub:
                assert_csc_equal(got[m], rtm[m], f"{sub} pipeline: {m}")
        if asm.nx >= 3 and COUNTS_ON:  # (otmb_step_dev needs the counts of its own facefluxes)
            asm.step_fused_async(umo, vmo, fill)
            asm.finish()
            got = asm.result_to_host()
            for m in MATS:
                if m not in sub:
                    assert_csc_equal(got[m], rtm[m], f"{sub} fused step: {m}")
    for sub in (("Tadv",), ("TκVML", "TκH")):
        asm.set_given(**{m: (_dev(asm, rtm[m]) if m in sub else None) for m in OPS})
        with pytest.raises(OtmbError) as e:
            asm.step(umo, vmo, fill)
        assert e.value.name == "GIVEN_FOREIGN"
    asm.set_given(**{m: None for m in OPS})
    asm.step(umo, vmo, fill)
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], f"afterwards: {m}")


@pytest.mark.parametrize("name", ["tiny_tripolar", "tiny_rho3d", "odd_nx_fold", "nx2", "even_fold_open", "small_rho3d", "float32_flux"])
def test_operators_built_with_another_kappa_enter_T_with_their_own_values(oracle, name):
    """The reference adds the OBJECTS passed in (:147).  A TκH / TκVdeep built with another κ has the derived rows and other values (state 3):
    T must carry ITS values -- equal to building everything with that κ -- never the re-derived ones; the fill pass reads them where they lie,
    in every protocol (two-phase, fused step with and without the one-pass count)."""
    g, gm, ref, rphi, rtm, asm, umo, vmo, fill = _setup(oracle, name)
    N = ref["N"]
    other = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, 123.0, g.kappaVML, 7.5e-5, True)
    for sub in (("TκH",), ("TκVdeep",), ("TκH", "TκVdeep")):
        ops = {m: (other[m] if m in sub else rtm[m]) for m in OPS}
        want = _fold(oracle, ops, N)
        asm.set_given(**{m: (_dev(asm, other[m]) if m in sub else None) for m in OPS})
        for protocol in ("two-phase", "step", "step onepass") + (("fused",) if asm.nx >= 3 and COUNTS_ON else ()):
            asm.out = None
            if protocol == "two-phase":
                asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
            elif protocol == "fused":  # (otmb_step_dev: tm_kernel<1 | 2, GIVEN> -- Float64 / Float32 transports)
                asm.step_fused_async(umo, vmo, fill)
                asm.finish()
            else:
                asm.step(umo, vmo, fill, onepass=protocol.endswith("onepass"))
            assert [asm.ctx.given_state(MATS.index(m)) for m in sub] == [3] * len(sub)
            got = asm.result_to_host()
            assert_csc_equal(got["T"], want, f"{sub} with another κ, {protocol}: T")
            if sub == ("TκH", "TκVdeep"):
                assert_csc_equal(got["T"], other["T"], "the same as building with those κ")
            assert not np.array_equal(got["T"][2], rtm["T"][2])
            for m in MATS[1:]:
                if m not in sub:
                    assert_csc_equal(got[m], rtm[m], f"{sub} with another κ, {protocol}: {m}")
    # one given operator derived, the other with other values
    asm.set_given(TκH=_dev(asm, rtm["TκH"]), TκVdeep=_dev(asm, other["TκVdeep"]), Tadv=None, TκVML=None)
    phi = asm.facefluxes(umo, vmo, fill)
    asm.out = None
    asm.transportmatrix(phi)
    assert (asm.ctx.given_state(2), asm.ctx.given_state(4)) == (1, 3)
    got = asm.result_to_host()
    assert_csc_equal(got["T"], _fold(oracle, {**rtm, "TκVdeep": other["TκVdeep"]}, N), "derived TκH + TκVdeep of another κ")
    asm.set_given(TκH=_dev(asm, other["TκH"]), TκVdeep=_dev(asm, rtm["TκVdeep"]), Tadv=None, TκVML=None)
    asm.out = None
    asm.step(umo, vmo, fill)
    assert (asm.ctx.given_state(2), asm.ctx.given_state(4)) == (3, 1)
    assert_csc_equal(asm.result_to_host()["T"], _fold(oracle, {**rtm, "TκH": other["TκH"]}, N), "TκH of another κ + derived TκVdeep")


@pytest.mark.parametrize("seed", range(40))
def test_random_grids_random_given_operators(oracle, seed):
    """Differential sweep: random small grids (nx 2..11, tripolar / bipolar, with and without land, scalar / 3-D ρ, randomised metrics, both
    weightings), a random subset of given operators, each either the oracle's own, one built with another κ, or the derived rows with
    arbitrary values (NaN and signed zeros among them); two-phase call and -- when nothing is foreign -- the asynchronous and fused steps.
    T is the oracle's left fold over the operands, bit for bit; what is built is the oracle's."""
    import torch

    from helpers import gridmetrics_of, randomize_metrics
    from otmb_amd import synthetic
    from otmb_amd.device import DeviceAssembler

    rng = np.random.default_rng(4200 + seed)
    nx, ny, nz = int(rng.integers(2, 12)), int(rng.integers(3, 9)), int(rng.integers(1, 7))
    topo = "tripolar" if rng.random() < 0.65 else "bipolar"
    upwind = bool(rng.random() < 0.7)
    g = synthetic.make_grid(nx, ny, nz, seed=900 + seed, land_fraction=float(rng.choice([0.0, 0.15, 0.4])), topology=topo,
                            rho=str(rng.choice(["scalar", "array"])))
    gm = gridmetrics_of(g)
    if rng.random() < 0.5 or (topo == "tripolar" and nx % 2 == 1):  # (odd nx on a tripolar grid: the fold-centre cell's real distance to itself is 0 -> NaN, :61)
        randomize_metrics(gm, seed=seed)
    ref = oracle.makeindices(gm.v3D)
    N = ref["N"]
    if N == 0:
        pytest.skip("no wet cell")
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    kap = (g.kappaH, g.kappaVML, g.kappaVdeep)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, *kap, upwind)
    other = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, 61.0, 0.37, 2.5e-4, upwind)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, *kap, upwind=upwind)
    dev = asm.device
    umo, vmo = (torch.from_numpy(np.ascontiguousarray(x.data.ravel(order="F"))).to(dev) for x in (g.umo, g.vmo))
    for trial in range(3):
        sub = [m for m in OPS if rng.random() < 0.5] or [OPS[int(rng.integers(0, 4))]]
        ops, states = dict(rtm), {}
        for m in sub:
            kind = int(rng.integers(0, 3))
            if kind == 1:
                ops[m] = other[m]
            elif kind == 2 and len(rtm[m][2]):
                v = rng.standard_normal(len(rtm[m][2]))
                v[rng.integers(0, len(v), size=max(1, len(v) // 7))] = rng.choice([0.0, -0.0, np.nan, 1e300, -1e-300], size=max(1, len(v) // 7))
                ops[m] = (rtm[m][0], rtm[m][1], v)
            same = ops[m] is rtm[m] or np.array_equal(ops[m][2].view(np.int64), rtm[m][2].view(np.int64))
            states[m] = (1 if same else 3) if m in ("TκH", "TκVdeep") else 2
        want = {m: ops[m] for m in OPS}
        want["T"] = _fold(oracle, ops, N)
        asm.set_given(**{m: (_dev(asm, ops[m]) if m in sub else None) for m in OPS})
        fast = all(states[m] != 2 for m in sub) and COUNTS_ON
        protocols = ["two-phase"] + (["async"] if fast else []) + (["fused"] if fast and nx >= 3 else [])  # (otmb_step_dev needs nx >= 3)
        for protocol in protocols:
            asm.out = None
            if protocol == "two-phase":
                asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
            elif protocol == "async":
                asm.step_async(umo, vmo, fill)
                asm.finish()
            else:
                asm.step_fused_async(umo, vmo, fill)
                asm.finish()
            what = f"seed {seed} ({nx}x{ny}x{nz} {topo}, upwind={upwind}) trial {trial} given {sub} {protocol}"
            assert {m: asm.ctx.given_state(MATS.index(m)) for m in sub} == states, what
            got = asm.result_to_host()
            for m in MATS:
                if m in sub:
                    continue
                a, b = got[m], want[m]
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), f"{what}: pattern of {m}"
                assert np.array_equal(np.asarray(a[2]).view(np.int64), np.asarray(b[2]).view(np.int64)) or _same_but_nan_payload(a[2], b[2]), f"{what}: values of {m}"


def _same_but_nan_payload(a, b):
    """bit-equal except where both are NaN (an addition may return either operand's payload)"""
    a, b = np.asarray(a), np.asarray(b)
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.array_equal(a[~nan].view(np.int64), b[~nan].view(np.int64)))


def test_foreign_patterns_and_one_changed_bit(oracle):
    """What a given TκH is taken for: one value bit changed (-0.0 for +0.0 included) leaves the derived rows -- state 3, the values are read;
    one row index, one column offset, a shorter matrix, an empty one, a diagonal matrix are foreign -- state 2, the sparse add.  Either way
    the matrix is ADDED as it is."""
    import scipy.sparse as sp

    g, gm, ref, rphi, rtm, asm, umo, vmo, fill = _setup(oracle, "small_rho3d")
    N = ref["N"]
    cp, rv, nz = rtm["TκH"]

    def run(H):
        asm.set_given(TκH=_dev(asm, H), Tadv=None, TκVML=None, TκVdeep=None)
        phi = asm.facefluxes(umo, vmo, fill)
        asm.out = None
        asm.transportmatrix(phi)
        return asm.ctx.given_state(2), asm.result_to_host()

    st, got = run((cp, rv, nz))
    assert st == 1
    variants = {}
    v = nz.copy(); v[len(v) // 2] = np.nextafter(v[len(v) // 2], np.inf); variants["one ulp"] = (cp, rv, v)
    v = nz.copy(); v[7] = -v[7]; variants["a sign"] = (cp, rv, v)
    r = rv.copy(); c = int(np.searchsorted(cp, len(rv) // 3)); q = cp[c] - 1
    if cp[c + 1] - cp[c] >= 2:
        r[q], r[q + 1] = r[q + 1], r[q]  # rows of a column swapped (not a valid CSC order, but foreign all the same)
        x = nz.copy(); x[q], x[q + 1] = x[q + 1], x[q]
        variants["two rows swapped"] = None  # (unsorted rows are not a SparseMatrixCSC: only the verdict is checked below)
    diag = sp.identity(N, format="csc") * 2.5
    variants["diagonal"] = ((diag.indptr + 1).astype(np.int64), (diag.indices + 1).astype(np.int64), diag.data.astype(np.float64))
    variants["empty"] = (np.ones(N + 1, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float64))
    drop = sp.csc_matrix((nz, rv - 1, cp - 1), shape=(N, N)).tolil()
    rr, cc = drop.nonzero()
    drop[rr[5], cc[5]] = 0.0
    drop = drop.tocsc(); drop.eliminate_zeros(); drop.sort_indices()
    variants["one entry less"] = ((drop.indptr + 1).astype(np.int64), (drop.indices + 1).astype(np.int64), drop.data.astype(np.float64))
    for what, H in variants.items():
        if H is None:
            continue
        st, got = run(H)
        assert st == (3 if what in ("one ulp", "a sign") else 2), what
        assert_csc_equal(got["T"], _fold(oracle, {**rtm, "TκH": H}, N), f"{what}: T")
        for m in ("Tadv", "TκVML", "TκVdeep"):
            assert_csc_equal(got[m], rtm[m], f"{what}: {m}")
    # a stored +0.0 where the derived matrix stores -0.0 (or the other way round) is another matrix: bits, not values, are compared
    g0, gm0, ref0, rphi0, rtm0, asm0, umo0, vmo0, fill0 = _setup(oracle, "tiny_tripolar", kappa=(0.0, 0.0, 0.0))
    H = rtm0["TκH"]
    assert len(H[2]) > 0 and np.all(H[2] == 0.0)
    asm0.set_given(TκH=_dev(asm0, H))
    asm0.transportmatrix(asm0.facefluxes(umo0, vmo0, fill0))
    assert asm0.ctx.given_state(2) == 1
    flipped = (H[0], H[1], -H[2])
    asm0.set_given(TκH=_dev(asm0, flipped))
    asm0.out = None
    asm0.transportmatrix(asm0.facefluxes(umo0, vmo0, fill0))
    assert asm0.ctx.given_state(2) == 3
    got = asm0.result_to_host()
    assert_csc_equal(got["T"], _fold(oracle, {**rtm0, "TκH": flipped}, ref0["N"]), "signed zeros: T")


def test_exact_cancellation_with_given_operators(oracle):
    """`+` drops entries whose sum is exactly zero (:147).  (a) derived: κ = 0 leaves explicit zeros in the diffusive operators -- given or built,
    T holds only what does not cancel (the kernel's own compaction); (b) foreign: a given Tadv that is exactly -TκH cancels in the FIRST add of
    the fold, and those entries must come back through the later adds where TκVML / TκVdeep have them."""
    g, gm, ref, rphi, rtm, asm, umo, vmo, fill = _setup(oracle, "tiny_rho3d", kappa=(0.0, 0.0, 0.0))
    N = ref["N"]
    assert np.all(rtm["TκH"][2] == 0.0) and len(rtm["T"][1]) == len(rtm["Tadv"][1])
    asm.set_given(TκH=_dev(asm, rtm["TκH"]), TκVdeep=_dev(asm, rtm["TκVdeep"]))
    for onepass in (True, False):
        asm.out = None
        asm.step(umo, vmo, fill, onepass=onepass)
        assert (asm.ctx.given_state(2), asm.ctx.given_state(4)) == (1, 1)
        got = asm.result_to_host()
        assert_csc_equal(got["T"], rtm["T"], f"κ = 0, derived, onepass={onepass}: T")
        assert not np.any(got["T"][2] == 0.0)
    g, gm, ref, rphi, rtm, asm, umo, vmo, fill = _setup(oracle, "small_rho3d")
    N = ref["N"]
    neg = (rtm["TκH"][0], rtm["TκH"][1], -rtm["TκH"][2])
    asm.set_given(Tadv=_dev(asm, neg))
    asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
    got = asm.result_to_host()
    want = _fold(oracle, {**rtm, "Tadv": neg}, N)
    first = oracle.spadd(neg, rtm["TκH"], N)
    assert len(first[1]) == 0 and len(want[1]) > 0
    assert_csc_equal(got["T"], want, "Tadv = -TκH: T")


def test_the_verdict_is_cached_and_forgotten_when_an_array_changes(oracle):
    """One comparing pass per (given arrays, grid arrays, κ): later calls with the same tensors run none; an in-place edit of the given values,
    of a grid array, another κ, or forget_given() runs it again -- and finds the new truth."""
    import torch

    g, gm, ref, rphi, rtm, asm, umo, vmo, fill = _setup(oracle, "small_rho3d")
    H, D = _dev(asm, rtm["TκH"]), _dev(asm, rtm["TκVdeep"])
    asm.set_given(TκH=H, TκVdeep=D)
    c0 = _checks(asm)
    for _ in range(4):
        asm.step(umo, vmo, fill)
    assert _checks(asm) == c0 + 1 and (asm.ctx.given_state(2), asm.ctx.given_state(4)) == (1, 1)
    asm.transportmatrix(asm.facefluxes(umo, vmo, fill))  # the two-phase protocol shares the verdicts
    assert _checks(asm) == c0 + 1
    H[2][11] *= 2.0  # in place: torch bumps the tensor's version, the assembler tells the library to look again
    asm.out = None
    asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
    assert _checks(asm) == c0 + 2 and (asm.ctx.given_state(2), asm.ctx.given_state(4)) == (3, 1)
    H2 = (rtm["TκH"][0], rtm["TκH"][1], H[2].cpu().numpy())
    assert_csc_equal(asm.result_to_host()["T"], _fold(oracle, {**rtm, "TκH": H2}, ref["N"]), "edited TκH: T")
    H[2][11] /= 2.0
    asm.step(umo, vmo, fill)
    assert _checks(asm) == c0 + 3 and asm.ctx.given_state(2) == 1
    asm.kappa = (asm.kappa[0], asm.kappa[1], 3.0e-5)  # another κVdeep: the given TκVdeep is no longer what would be derived
    asm.out = None
    asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
    assert (asm.ctx.given_state(2), asm.ctx.given_state(4)) == (1, 3)
    assert_csc_equal(asm.result_to_host()["T"], rtm["T"], "the given TκVdeep's values, not those of the call's κVdeep")
    asm.kappa = (g.kappaH, g.kappaVML, g.kappaVdeep)
    asm.area.mul_(1.0)  # an in-place op on a grid array (values unchanged): looked at again, still derived
    n = _checks(asm)
    asm.step(umo, vmo, fill)
    assert _checks(asm) == n + 1 and (asm.ctx.given_state(2), asm.ctx.given_state(4)) == (1, 1)
    got = asm.result_to_host()
    for m in ("T", "Tadv", "TκVML"):
        assert_csc_equal(got[m], rtm[m], m)
    del torch


def test_errors_of_a_given_operator_are_not_raised_and_the_others_are(oracle):
    """matrixbuilding.jl:140-143: a given operator is never built, so nothing it alone would raise is raised -- a NaN in ρ or a flux into land
    with Tadv given, a NaN metric with TκH given -- while the operators that ARE built keep their checks."""
    import torch

    from otmb_amd.capi import OtmbError

    g, gm, ref, rphi, rtm, asm, umo, vmo, fill = _setup(oracle, "tiny_rho3d")
    L = int(asm.lwet[5].item()) - 1
    old = asm.rho[L].clone()
    asm.rho[L] = float("nan")
    asm.set_given(TκH=_dev(asm, rtm["TκH"]))
    with pytest.raises(OtmbError, match="ρ contains NaNs"):
        asm.step(umo, vmo, fill)
    asm.set_given(Tadv=_dev(asm, rtm["Tadv"]), TκH=None)
    asm.out = None
    asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
    assert_csc_equal(asm.result_to_host()["T"], rtm["T"], "ρ NaN with Tadv given: T")
    asm.rho[L] = old
    asm.set_given(Tadv=None)
    e0 = asm.edge[0].clone()
    wet2 = (asm.wet3d[: asm.nx * asm.ny] != 0)
    asm.edge[0][torch.nonzero(wet2)[3]] = float("nan")
    with pytest.raises(OtmbError, match="TκH contains NaNs."):
        asm.step(umo, vmo, fill)
    asm.set_given(TκH=_dev(asm, rtm["TκH"]))  # (other values now: the grid it is compared with has a NaN edge)
    asm.out = None
    asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
    assert asm.ctx.given_state(2) == 3
    assert_csc_equal(asm.result_to_host()["T"], rtm["T"], "NaN metric with TκH given: T")
    asm.edge[0].copy_(e0)
    asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
    assert asm.ctx.given_state(2) == 1
    # a given operator that holds NaNs is added as it is (the reference's `+` does not look)
    bad = (rtm["TκVdeep"][0], rtm["TκVdeep"][1], rtm["TκVdeep"][2].copy())
    bad[2][3] = np.nan
    asm.set_given(TκH=None, TκVdeep=_dev(asm, bad))
    asm.out = None
    asm.transportmatrix(asm.facefluxes(umo, vmo, fill))
    assert int(np.isnan(asm.result_to_host()["T"][2]).sum()) == 1
    with pytest.raises(ValueError):
        asm.set_given(T=_dev(asm, rtm["T"]))
    with pytest.raises(ValueError):
        asm.set_given(TκH=_dev(asm, (rtm["TκH"][0][:-1], rtm["TκH"][1], rtm["TκH"][2])))


# ---- through the host API (what a Julia caller does) -----------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def api():
    import otmb_amd.api as api

    return api


def _host_case(api, oracle, grid):
    from otmb_amd import synthetic

    g = synthetic.make_grid(*grid[:3], seed=grid[3], rho="array")
    gm = gridmetrics_of(g)
    ref = oracle.makeindices(gm.v3D)
    idx = api.makeindices(gm.v3D)
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], 1e20, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    kw = dict(ϕ=rphi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH, κVML=g.kappaVML, κVdeep=g.kappaVdeep)
    return g, gm, ref, idx, rphi, rtm, kw


@pytest.mark.parametrize("slabs", [0, 1, 3, None])
def test_host_api_returns_the_objects_passed_and_moves_fewer_bytes(api, oracle, slabs):
    """buildTκH / buildTκVdeep once, then transportmatrix(...; TκH, TκVdeep) per time slice (the TMIP loop): the two objects come back as they
    are, T / Tadv / TκVML are the oracle's, through the two-phase call (slabs = 0), the pipelined call on 1 and 3 slabs and the default."""
    g, gm, ref, idx, rphi, rtm, kw = _host_case(api, oracle, (40, 30, 12, 61))
    H = api.buildTκH(gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH)
    D = api.buildTκVdeep(mlotst=g.mlotst, gridmetrics=gm, indices=idx, κVdeep=g.kappaVdeep)
    assert_csc_equal(tuple(H), rtm["TκH"], "buildTκH")
    assert_csc_equal(tuple(D), rtm["TκVdeep"], "buildTκVdeep")
    for call in range(3):
        tm = api.transportmatrix(TκH=H, TκVdeep=D, slabs=slabs, reuse_grid=call > 0, **kw)
        assert tm.TκH is H and tm.TκVdeep is D
        for m in ("T", "Tadv", "TκVML"):
            assert_csc_equal(tuple(tm[m]), rtm[m], f"slabs={slabs} call {call}: {m}")
    # all four given
    A = api.buildTadv(ϕ=rphi, gridmetrics=gm, indices=idx, ρ=g.rho)
    M = api.buildTκVML(mlotst=g.mlotst, gridmetrics=gm, indices=idx, κVML=g.kappaVML)
    tm = api.transportmatrix(ϕ=None, ρ=None, mlotst=None, gridmetrics=gm, indices=idx, Tadv=A, TκH=H, TκVML=M, TκVdeep=D, slabs=slabs)
    assert tm.Tadv is A and tm.TκVML is M
    assert_csc_equal(tuple(tm.T), rtm["T"], f"slabs={slabs}: T from four given operators")


def test_host_api_another_kappa_is_pipelined_and_a_foreign_matrix_falls_back(api, oracle):
    """Another κ through the default (pipelined, multi-slab) call: the derived rows with other values -- every slab reads its slice, no fall-back,
    T carries the given values.  A matrix with other rows: the slabs refuse (GIVEN_FOREIGN), the host layer takes the two-phase call and
    remembers it for the next time slice."""
    g, gm, ref, idx, rphi, rtm, kw = _host_case(api, oracle, (40, 30, 12, 62))
    N = ref["N"]
    H2 = api.buildTκH(gridmetrics=gm, indices=idx, ρ=g.rho, κH=77.0)
    D2 = api.buildTκVdeep(mlotst=g.mlotst, gridmetrics=gm, indices=idx, κVdeep=4.0e-5)
    other = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, 77.0, g.kappaVML, 4.0e-5, True)
    assert_csc_equal(tuple(H2), other["TκH"], "buildTκH(κH = 77)")
    assert_csc_equal(tuple(D2), other["TκVdeep"], "buildTκVdeep(κVdeep = 4e-5)")
    seen = set(api._foreign_seen)
    for call, slabs in enumerate((3, None, 0, 1)):
        tm = api.transportmatrix(TκH=H2, TκVdeep=D2, slabs=slabs, **kw)  # (the call's own κH / κVdeep are the grid's defaults: not used, :141-143)
        assert tm.TκH is H2 and tm.TκVdeep is D2
        assert_csc_equal(tuple(tm.T), other["T"], f"call {call} (slabs={slabs}): T with TκH(77), TκVdeep(4e-5) given")
        for m in ("Tadv", "TκVML"):
            assert_csc_equal(tuple(tm[m]), rtm[m], m)
    assert set(api._foreign_seen) == seen  # no fall-back was needed
    Hd = api.SparseMatrixCSC(N, N, np.arange(1, N + 2, dtype=np.int64), np.arange(1, N + 1, dtype=np.int64), np.full(N, 2.5))
    want = _fold(oracle, {**rtm, "TκH": (Hd.colptr, Hd.rowval, Hd.nzval)}, N)
    for call in range(2):
        tm = api.transportmatrix(TκH=Hd, slabs=3 if call == 0 else None, **kw)
        assert tm.TκH is Hd
        assert_csc_equal(tuple(tm.T), want, f"call {call}: T with a diagonal TκH given")
        for m in ("Tadv", "TκVML", "TκVdeep"):
            assert_csc_equal(tuple(tm[m]), rtm[m], m)
    assert api._foreign_key(dict(TκH=Hd)) in api._foreign_seen
    with pytest.raises(ValueError):
        api.transportmatrix(TκH=api.SparseMatrixCSC(3, 3, np.ones(4, np.int64), np.zeros(0, np.int64), np.zeros(0)), **kw)


def test_host_api_upload_and_download_bytes_with_given_operators(api, oracle):
    """What the switch is for: with TκH and TκVdeep given and the reuse_grid promise, a time slice uploads neither the grid nor the two
    operators, and the library hands back three matrices instead of five."""
    g, gm, ref, idx, rphi, rtm, kw = _host_case(api, oracle, (40, 30, 12, 63))
    H = api.buildTκH(gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH)
    D = api.buildTκVdeep(mlotst=g.mlotst, gridmetrics=gm, indices=idx, κVdeep=g.kappaVdeep)
    ctx = api.context(0)
    lib = __import__("otmb_amd.capi", fromlist=["lib"]).lib()
    up = lambda: int(lib.otmb_ctx_uploaded_bytes(ctx.handle))
    G, P = gm.v3D.size, gm.v3D.shape[0] * gm.v3D.shape[1]
    api.transportmatrix(TκH=H, TκVdeep=D, slabs=0, reuse_grid=True, **kw)
    api.transportmatrix(TκH=H, TκVdeep=D, slabs=0, reuse_grid=True, **kw)
    b0, c0 = up(), int(lib.otmb_ctx_given_checks(ctx.handle))
    tm = api.transportmatrix(TκH=H, TκVdeep=D, slabs=0, reuse_grid=True, **kw)
    per_slice = 8 * (6 * G + G + P)  # ϕ, ρ, mlotst: everything else is resident -- the given operators included
    assert up() - b0 == per_slice, (up() - b0, per_slice)
    assert int(lib.otmb_ctx_given_checks(ctx.handle)) == c0  # ... and their verdict is kept
    assert_csc_equal(tuple(tm.T), rtm["T"], "T")
    b0 = up()
    api.transportmatrix(TκH=H, TκVdeep=D, slabs=0, reuse_grid=False, **kw)  # no promise: everything goes up again and is compared again
    assert up() - b0 > per_slice + 16 * (len(H.rowval) + len(D.rowval))
    assert int(lib.otmb_ctx_given_checks(ctx.handle)) == c0 + 1
