// otmb_redigm.hip -- bolus_GM_velocity (src/RediGM.jl:46-79): vertical-face triad slopes of ρ in i and j
// (src/triads.jl:84-146), clamp, tanh taper, then the vertical dyad derivative of κGM·S (src/dyads.jl:38-78).
// EXPERIMENTAL in the reference ("Note: This is experimental at this stage", RediGM.jl:44); its result never
// enters T and no reference test asserts anything about it: PARITY UNPINNED, checked against the oracle only.
// Two streaming kernels over all cells (NaN outside indices.Lwet, as fill(NaN, size(χ)) does).  The horizontal
// distance C->E (horizontaldistance, gridcellgeometry.jl:182-188) is distance_to_neighbour_2D[:east|:north].
// Julia's `false * NaN == 0.0` in the NaN-aware means is a select here, never a multiply.
#include "otmb_common.h"
#include "otmb_topology.h"

__device__ __forceinline__ double gm_gnan(const double *x, i64 L) { return (L < 0) ? __builtin_nan("") : x[L]; }

__device__ __forceinline__ double gm_triad(const double *__restrict__ chi, const double *__restrict__ Z, double dCE, i64 I, i64 N,
                                           i64 S, i64 E, i64 NE, i64 SE) {
    const double vC = chi[I], vN = gm_gnan(chi, N), vS = gm_gnan(chi, S), vE = gm_gnan(chi, E), vNE = gm_gnan(chi, NE),
                 vSE = gm_gnan(chi, SE);
    const double zI = Z[I], zE = gm_gnan(Z, E);
    const double dCN = fabs(gm_gnan(Z, N) - zI), dCS = fabs(gm_gnan(Z, S) - zI);
    const double dENE = fabs(gm_gnan(Z, NE) - zE), dESE = fabs(gm_gnan(Z, SE) - zE);
    const double CN = (vN - vC) / dCN, CS = (vC - vS) / dCS, CE = (vE - vC) / dCE, ENE = (vNE - vE) / dENE, ESE = (vE - vSE) / dESE;
    const double r0 = CE / CN, r1 = CE / CS, r2 = CE / ENE, r3 = CE / ESE;  // triads.jl:123-128
    const bool w0 = !isnan(r0), w1 = !isnan(r1), w2 = !isnan(r2), w3 = !isnan(r3);
    const double s = (((w0 ? r0 : 0.0) + (w1 ? r1 : 0.0)) + (w2 ? r2 : 0.0)) + (w3 ? r3 : 0.0);
    return s / (double)((int)w0 + (int)w1 + (int)w2 + (int)w3);  // :130-132
}

__global__ __launch_bounds__(256) void gm_slopes_kernel(const double *__restrict__ rho, const double *__restrict__ Z,
                                                        const uint8_t *__restrict__ wet, const double *__restrict__ dist_e,
                                                        const double *__restrict__ dist_n, int nx, int ny, int nz, i64 P, i64 G,
                                                        double kappaGM, double maxslope, double *__restrict__ Ki,
                                                        double *__restrict__ Kj) {
    const i64 L = (i64)blockIdx.x * 256 + threadIdx.x;
    if (L >= G) return;
    double si = __builtin_nan(""), sj = __builtin_nan("");
    if (wet[L]) {
        const Cell c = cell_of(L, nx, ny, P);
        const i64 N = nb_km1(c, P), S = nb_kp1(c, nz, P);
        const i64 Ei = nb_ip1(c, nx), Ej = nb_jp1(c, nx, ny, OTMB_TRIPOLAR);  // host rejects j₊₁ == nothing
        const i64 s2 = (i64)c.j * nx + c.i;
        si = gm_triad(rho, Z, dist_e[s2], L, N, S, Ei, (c.k > 0) ? Ei - P : -1, (c.k + 1 < nz) ? Ei + P : -1);
        sj = gm_triad(rho, Z, dist_n[s2], L, N, S, Ej, (c.k > 0) ? Ej - P : -1, (c.k + 1 < nz) ? Ej + P : -1);
    }
    si = (si > maxslope) ? maxslope : ((si < -maxslope) ? -maxslope : si);  // clamp, RediGM.jl:56-57 (NaN stays NaN)
    sj = (sj > maxslope) ? maxslope : ((sj < -maxslope) ? -maxslope : sj);
    const double taper = 0.5 * (1 + tanh((0.004 - sqrt(si * si + sj * sj)) / 0.001));  // :59-62
    Ki[L] = kappaGM * (taper * si);  // :63, :76
    Kj[L] = kappaGM * (taper * sj);  // :64, :77
}

__device__ __forceinline__ double gm_dyad(const double *__restrict__ chi, const double *__restrict__ Z, i64 I, i64 N, i64 S) {
    const double zI = Z[I], c = chi[I];
    const double a = (gm_gnan(chi, N) - c) / fabs(gm_gnan(Z, N) - zI), b = (c - gm_gnan(chi, S)) / fabs(gm_gnan(Z, S) - zI);
    const bool wa = !isnan(a), wb = !isnan(b);
    return ((wa ? a : 0.0) + (wb ? b : 0.0)) / (double)((int)wa + (int)wb);  // dyads.jl:57-65
}

__global__ __launch_bounds__(256) void gm_dyad_kernel(const double *__restrict__ Ki, const double *__restrict__ Kj,
                                                      const double *__restrict__ Z, const uint8_t *__restrict__ wet, int nz, i64 P,
                                                      i64 G, double *__restrict__ u, double *__restrict__ v) {
    const i64 L = (i64)blockIdx.x * 256 + threadIdx.x;
    if (L >= G) return;
    double a = __builtin_nan(""), b = __builtin_nan("");
    if (wet[L]) {
        const i64 k = L / P;
        const i64 N = (k > 0) ? L - P : -1, S = (k + 1 < nz) ? L + P : -1;
        a = gm_dyad(Ki, Z, L, N, S);
        b = gm_dyad(Kj, Z, L, N, S);
    }
    u[L] = a;
    v[L] = b;
}

// Both steps in one kernel: a workgroup takes 64 consecutive columns of the plane, its eight waves an eighth of the levels each.  Every
// thread computes κGM·S of its cells exactly as gm_slopes_kernel does and posts it in LDS ([2][nz][64] doubles: 51 KB at nz = 50); after one
// barrier it takes the vertical dyad derivative of its own cells from there (the levels k - 1 / k + 1 may belong to the wave next to it).
// The two (nx,ny,nz) arrays between the kernels -- written, then read three times over -- never exist: 178 MB instead of 350 MB of traffic at
// 1 degree; what remains is the slopes' arithmetic (~25 Float64 divisions and a tanh per cell, in long dependent chains): 0.178-0.183 ->
// 0.164-0.166 ms (4 / 8 / 16 waves per workgroup: 0.169 / 0.165 / 0.176 ms).  Same operations in the same order: bit-identical.
#define GM_COLS 64
#define GM_GROUPS 8
__global__ __launch_bounds__(GM_COLS *GM_GROUPS) void gm_fused_kernel(const double *__restrict__ rho, const double *__restrict__ Z,
                                                                      const uint8_t *__restrict__ wet, const double *__restrict__ dist_e,
                                                                      const double *__restrict__ dist_n, int nx, int ny, int nz, i64 P,
                                                                      double kappaGM, double maxslope, double *__restrict__ u,
                                                                      double *__restrict__ v) {
    extern __shared__ double gm_lds[];  // Ki: [nz][64], then Kj: [nz][64]
    double *sKi = gm_lds, *sKj = gm_lds + (size_t)nz * GM_COLS;
    const int lane = threadIdx.x & (GM_COLS - 1), grp = threadIdx.x / GM_COLS;
    const i64 s = (i64)blockIdx.x * GM_COLS + lane;
    const bool inside = s < P;
    const int kper = (nz + GM_GROUPS - 1) / GM_GROUPS, k0 = grp * kper, k1 = (k0 + kper < nz) ? k0 + kper : nz;
    const int j = inside ? (int)(s / nx) : 0, i = inside ? (int)(s - (i64)j * nx) : 0;
    const double de = inside ? dist_e[s] : 0.0, dn = inside ? dist_n[s] : 0.0;
    for (int k = k0; k < k1; ++k) {
        double ki = __builtin_nan(""), kj = __builtin_nan("");
        if (inside) {
            const i64 L = (i64)k * P + s;
            double si = __builtin_nan(""), sj = __builtin_nan("");
            if (wet[L]) {
                Cell c;
                c.i = i; c.j = j; c.k = k; c.L = L; c.row0 = L - i;
                const i64 N = nb_km1(c, P), S = nb_kp1(c, nz, P);
                const i64 Ei = nb_ip1(c, nx), Ej = nb_jp1(c, nx, ny, OTMB_TRIPOLAR);
                si = gm_triad(rho, Z, de, L, N, S, Ei, (k > 0) ? Ei - P : -1, (k + 1 < nz) ? Ei + P : -1);
                sj = gm_triad(rho, Z, dn, L, N, S, Ej, (k > 0) ? Ej - P : -1, (k + 1 < nz) ? Ej + P : -1);
            }
            si = (si > maxslope) ? maxslope : ((si < -maxslope) ? -maxslope : si);
            sj = (sj > maxslope) ? maxslope : ((sj < -maxslope) ? -maxslope : sj);
            const double taper = 0.5 * (1 + tanh((0.004 - sqrt(si * si + sj * sj)) / 0.001));
            ki = kappaGM * (taper * si);
            kj = kappaGM * (taper * sj);
        }
        sKi[(size_t)k * GM_COLS + lane] = ki;
        sKj[(size_t)k * GM_COLS + lane] = kj;
    }
    __syncthreads();
    if (!inside) return;
    for (int k = k0; k < k1; ++k) {
        const i64 L = (i64)k * P + s;
        double a = __builtin_nan(""), b = __builtin_nan("");
        if (wet[L]) {
            const double nanv = __builtin_nan("");
            const double zI = Z[L], zN = (k > 0) ? Z[L - P] : nanv, zS = (k + 1 < nz) ? Z[L + P] : nanv;
            const double dN = fabs(zN - zI), dS = fabs(zS - zI);
            {   // gm_dyad on κGM·S_i
                const double c = sKi[(size_t)k * GM_COLS + lane];
                const double cn = (k > 0) ? sKi[(size_t)(k - 1) * GM_COLS + lane] : nanv, cs = (k + 1 < nz) ? sKi[(size_t)(k + 1) * GM_COLS + lane] : nanv;
                const double x = (cn - c) / dN, y = (c - cs) / dS;
                const bool wx = !isnan(x), wy = !isnan(y);
                a = ((wx ? x : 0.0) + (wy ? y : 0.0)) / (double)((int)wx + (int)wy);
            }
            {
                const double c = sKj[(size_t)k * GM_COLS + lane];
                const double cn = (k > 0) ? sKj[(size_t)(k - 1) * GM_COLS + lane] : nanv, cs = (k + 1 < nz) ? sKj[(size_t)(k + 1) * GM_COLS + lane] : nanv;
                const double x = (cn - c) / dN, y = (c - cs) / dS;
                const bool wx = !isnan(x), wy = !isnan(y);
                b = ((wx ? x : 0.0) + (wy ? y : 0.0)) / (double)((int)wx + (int)wy);
            }
        }
        u[L] = a;
        v[L] = b;
    }
}

extern "C" int32_t otmb_bolus_gm_velocity_dev(otmb_ctx *ctx, const double *rho, const double *z3d, const uint8_t *wet3d,
                                              const double *dist_east, const double *dist_north, int64_t nx, int64_t ny,
                                              int64_t nz, int32_t topology, double kappa_gm, double maxslope, double *u,
                                              double *v) {
    if (!ctx || !rho || !z3d || !wet3d || !dist_east || !dist_north || !u || !v) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1 || nx * ny * nz >= (1ll << 32)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    if (topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);
    // bipolar: E = j₊₁ is `nothing` on the top row and k₋₁(nothing) throws in the reference (triads.jl:87-88)
    if (topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_FLUX_INTO_LAND, "bolus_GM_velocity indexes j₊₁ == nothing on a bipolar grid");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 P = nx * ny, G = P * nz;
    int32_t rc;
    const size_t lds = (size_t)2 * nz * GM_COLS * sizeof(double);
    // the fused kernel keeps κGM·S of a block of columns in LDS: as many levels as the DEVICE's opt-in limit holds (queried once per context,
    // like the switch OTMB_GM_FUSED=0 -- the two streaming kernels, A/B and tests); if it cannot be launched the two-kernel path below runs
    if (ctx->gm_lds_limit < 0) {
        int lim = 0;
        ctx->gm_lds_limit = (hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) == hipSuccess && lim > 0) ? lim : 64 * 1024;
        const char *sw = getenv("OTMB_GM_FUSED");
        if (sw && sw[0] == '0') ctx->gm_lds_limit = 0;
    }
    if (lds <= (size_t)ctx->gm_lds_limit) {
        bool ok = true;
        if (lds > (size_t)48 * 1024 && lds > ctx->gm_lds_set) {
            ok = hipFuncSetAttribute((const void *)gm_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
            if (ok) ctx->gm_lds_set = lds;
        }
        if (ok) {
            KernelTimer kt(ctx, K_GM);
            hipLaunchKernelGGL(gm_fused_kernel, dim3((unsigned)((P + GM_COLS - 1) / GM_COLS)), dim3(GM_COLS * GM_GROUPS), lds, ctx->stream, rho, z3d,
                               wet3d, dist_east, dist_north, (int)nx, (int)ny, (int)nz, P, kappa_gm, maxslope, u, v);
            ok = hipGetLastError() == hipSuccess;
        }
        if (ok) return OTMB_OK;
        (void)hipGetLastError();
        ctx->gm_lds_limit = 0;  // (not on this device: the streaming pair from now on)
    }
    if ((rc = otmb_reserve(ctx, ctx->tfix[1], (size_t)G * 8))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->tfix[2], (size_t)G * 8))) return rc;
    double *Ki = (double *)ctx->tfix[1].p, *Kj = (double *)ctx->tfix[2].p;
    const unsigned nb = (unsigned)((G + 255) / 256);
    {
        KernelTimer kt(ctx, K_GM);
        hipLaunchKernelGGL(gm_slopes_kernel, dim3(nb), dim3(256), 0, ctx->stream, rho, z3d, wet3d, dist_east, dist_north, (int)nx,
                           (int)ny, (int)nz, P, G, kappa_gm, maxslope, Ki, Kj);
        hipLaunchKernelGGL(gm_dyad_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const double *)Ki, (const double *)Kj, z3d, wet3d,
                           (int)nz, P, G, u, v);
    }
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}
