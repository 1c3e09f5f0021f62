"""Size-independent properties of the hot path's device-resident result (TEST INFRASTRUCTURE; torch on the GPU).

What the reference's own tests assert about its matrices -- CSC Float64/Int64 well-formedness
(test/online.jl:93-95), ‖1‖/‖M·1‖ and ‖v‖/‖Mᵀv‖ above 10⁶ Myr for the diffusive operators (:110-115),
diag(T) > 0 and off-diag(T) < 0 for upwind (:119-123) -- plus the identities that follow from the
algorithm: T is the sum of the four operators, T stores no zero, the six face fluxes satisfy the shifts
and the continuity recurrence of src/velocities.jl:203-243 exactly.  Everything is evaluated level by level /
column chunk by column chunk so that it also runs on the 0.1 degree grid (nnz(T) = 2.7e9).
"""
import torch

from helpers import MATS

MYR = 365.25 * 86400 * 1e6


def check_facefluxes(phi, nx, ny, nz):
    """phi: the six flat device tensors in OTMB_EAST..OTMB_BOTTOM order."""
    e, w_, n_, s_, top, bot = [p.view(nz, ny, nx) for p in phi]  # torch (k, j, i) == Julia (i, j, k) column-major
    ok = dict(phi_finite=True, bottom_is_top_below=True, west_is_east_shifted=True, south_is_north_shifted=True,
              continuity_exact=True)
    for k in range(nz):
        ok["phi_finite"] &= bool(all(torch.isfinite(p[k]).all() for p in (e, w_, n_, s_, top, bot)))
        ok["bottom_is_top_below"] &= bool(torch.equal(bot[k], top[k + 1]) if k + 1 < nz else (bot[k] == 0).all())
        ok["west_is_east_shifted"] &= bool(torch.equal(w_[k], torch.roll(e[k], 1, dims=1)))
        ok["south_is_north_shifted"] &= bool(torch.equal(s_[k, 1:], n_[k, :-1]) and (s_[k, 0] == 0).all())
        ok["continuity_exact"] &= bool((((((bot[k] + w_[k]) + s_[k]) - e[k]) - n_[k]) - top[k] == 0).all())
    return ok


def check_matrices(out, nnz, N, vol_wet, upwind=True, nchunks=64, seed=0):
    """out: {name: (colptr, rowval, nzval)} device tensors, nnz: the five counts, vol_wet: v3D on the wet cells."""
    dev = vol_wet.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    x = torch.randn(N, dtype=torch.float64, device=dev, generator=gen)
    ones_norm = float(N) ** 0.5
    bounds = [N * c // nchunks for c in range(nchunks + 1)]
    checks = {}
    acc_Tx = torch.zeros(N, dtype=torch.float64, device=dev)
    Tx = None
    for kk, m in enumerate(MATS):
        cp, rv, nzv = out[m]
        nn = int(nnz[kk])
        wellformed = bool(cp[0] == 1 and cp[N] == nn + 1 and (cp[1:N + 1] >= cp[:N]).all())
        rowsum = torch.zeros(N, dtype=torch.float64, device=dev)
        MTv = torch.zeros(N, dtype=torch.float64, device=dev)
        Mx = torch.zeros(N, dtype=torch.float64, device=dev)
        diag_ok, offd_ok, nozero, ndiag = True, True, True, 0
        for c in range(nchunks):
            c0, c1 = bounds[c], bounds[c + 1]
            if c1 <= c0:
                continue
            a, b = int(cp[c0]) - 1, int(cp[c1]) - 1
            if b <= a:
                continue
            r, z = rv[a:b], nzv[a:b]
            cnt = cp[c0 + 1:c1 + 1] - cp[c0:c1]
            col = torch.repeat_interleave(torch.arange(c0, c1, device=dev), cnt)
            first = torch.zeros(b - a, dtype=torch.bool, device=dev)
            first[(cp[c0:c1] - 1 - a)[cnt > 0]] = True
            # rows strictly ascending inside a column: a descent is allowed only where a new column starts
            wellformed &= bool(((r[1:] > r[:-1]) | first[1:]).all()) and bool((r >= 1).all() and (r <= N).all())
            rowsum.index_add_(0, r - 1, z)
            MTv.index_add_(0, col, z * vol_wet[r - 1])
            Mx.index_add_(0, r - 1, z * x[col])
            if m == "T":
                isd = (r - 1) == col
                ndiag += int(isd.sum())
                diag_ok &= bool((z[isd] > 0).all())
                offd_ok &= bool((z[~isd] < 0).all())
                nozero &= bool((z != 0).all())
            del col, first, r, z, cnt
        checks[f"{m}_csc_wellformed"] = wellformed
        if m == "T":
            Tx = Mx
            if upwind:
                checks["T_diag_positive"] = diag_ok and ndiag == N
                checks["T_offdiag_negative"] = offd_ok
            checks["T_no_stored_zero"] = nozero
        else:
            acc_Tx += Mx
        if m not in ("T", "Tadv"):  # with a 3-D rho the advective operator conserves mass, not volume
            checks[f"{m}_divergence_Myr"] = float(ones_norm / rowsum.norm().clamp_min(1e-300) / MYR)
            checks[f"{m}_volume_Myr"] = float(vol_wet.norm() / MTv.norm().clamp_min(1e-300) / MYR)
        del rowsum, MTv
    checks["T_is_sum_of_operators_relerr"] = float((Tx - acc_Tx).norm() / Tx.norm())
    return checks


def failed(checks):
    return [k for k, val in checks.items()
            if (val is False) or (k.endswith("_Myr") and val < 1e6) or (k.endswith("relerr") and val > 1e-12)]
