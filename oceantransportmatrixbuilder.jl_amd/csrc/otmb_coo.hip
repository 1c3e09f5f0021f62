// otmb_coo.hip -- the reference's own two-step formulation, kept as a general (non-fused) path:
//   1. the three COO generators in the reference's emission order
//        advection_operator_sparse_entries            src/matrixbuilding.jl:221-299 (+ :193-204)
//        horizontal_diffusion_operator_sparse_entries src/matrixbuilding.jl:337-418 (+ :426-435)
//        vertical_diffusion_operator_sparse_entries   src/matrixbuilding.jl:438-479   (Ω = ML mask :85 or all :109)
//      one thread per wet cell 𝑖 (loop order of the reference = wet rank), count -> tile scan -> write;
//   2. sparse(I, J, V, m, n) (SparseArrays; called at :41,63,92,116): stable radix sort of the triplets by
//      (column, row), duplicates summed LEFT TO RIGHT in input order by the thread that owns the segment head
//      (first touch copies, as sparse! does), stored zeros kept, rows ascending inside a column.
// The fused kernel (otmb_transportmatrix.hip) never forms COO -- that is what saves the ~528 B per wet cell of
// intermediate traffic; this path exists because sparse() is a general contract (arbitrary triplets) and as an
// independent cross-check of the fused path on the device.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "otmb_tm_column.h"

#define COO_THREADS 256
enum { COO_ADV = 0, COO_H = 1, COO_VML = 2, COO_VDEEP = 3 };

template <int WHICH, bool WRITE>
__global__ __launch_bounds__(COO_THREADS) void coo_kernel(const TmParams p, uint32_t *__restrict__ tilesums, const i64 *__restrict__ tileoffs,
                                                           i64 *__restrict__ Iout, i64 *__restrict__ Jout, double *__restrict__ Vout) {
    __shared__ unsigned wave_tot[COO_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const i64 w = (i64)blockIdx.x * COO_THREADS + tid;
    i64 rows[12], cols[12];
    double vals[12];
    int n = 0;
    bool bad = false, nanv = false;
    if (w < p.n_own) {
        const i64 L = p.lwet[w] - 1;
        const Cell c = cell_of(L, p.nx, p.ny, p.P);
        const i64 wi = p.wet_base + w + 1;
        const i64 nb[6] = {nb_im1(c, p.nx), nb_ip1(c, p.nx), nb_jm1(c, p.nx), nb_jp1(c, p.nx, p.ny, p.topo), nb_kp1(c, p.nz, p.P), nb_km1(c, p.P)};
        const double vi = p.v[L];
#define PUSH(i_, j_, v_) { rows[n] = (i_); cols[n] = (j_); vals[n] = (v_); nanv |= isnan(vals[n]); ++n; }
        if (WHICH == COO_ADV) {
            const double ri = p.rho ? p.rho[L] : p.rho_s;
            if (isnan(ri)) raise_flag(p.flags, FLAG_RHO_NAN);  // :233
            // W, E, S, N, B, T with the selected flux and the sign passed to pushTadvectionvalues! (:244-296)
            const double f[6] = {sel_pos(p.phi[OTMB_WEST][L], p.upwind), sel_neg(p.phi[OTMB_EAST][L], p.upwind),
                                 sel_pos(p.phi[OTMB_SOUTH][L], p.upwind), sel_neg(p.phi[OTMB_NORTH][L], p.upwind),
                                 sel_pos(p.phi[OTMB_BOTTOM][L], p.upwind), (c.k > 0) ? sel_neg(p.phi[OTMB_TOP][L], p.upwind) : 0.0};
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                if (nonzero(f[d])) {
                    const i64 Cj = nb[d];
                    const i64 wj = (Cj >= 0) ? p.lw[Cj] : 0;
                    if (wj == 0) { bad = true; continue; }  // Lwet3D[nothing] / push!(…, missing)
                    const double phi = (d & 1) ? -f[d] : f[d];
                    const double rj = p.rho ? p.rho[Cj] : p.rho_s;
                    const double r = (ri + rj) / 2, mi = r * vi, mj = r * p.v[Cj];  // :194-196
                    PUSH(wi, wj, -phi / mi)  // :197-199
                    PUSH(wj, wj, phi / mj)   // :200-202
                }
            }
        } else if (WHICH == COO_H) {
            const i64 s = (i64)c.j * p.nx + c.i;
            const double thc = p.thk[L];
#pragma unroll
            for (int d = 0; d < 4; ++d) {  // W, E, S, N
                const i64 Cj = nb[d];
                if (Cj < 0) continue;
                const i64 wj = p.lw[Cj];
                if (wj == 0) continue;
                const int opp = (d == 0) ? 1 : (d == 1) ? 0 : (d == 2) ? 3 : ((c.j == p.ny - 1) ? 3 : 2);  // :407
                const i64 sj = Cj % p.P;
                const double *e_d = (d == 0) ? p.edge[0] : (d == 1) ? p.edge[1] : (d == 2) ? p.edge[2] : p.edge[3];
                const double *e_o = (opp == 0) ? p.edge[0] : (opp == 1) ? p.edge[1] : (opp == 2) ? p.edge[2] : p.edge[3];
                const double *d_d = (d == 0) ? p.dist[0] : (d == 1) ? p.dist[1] : (d == 2) ? p.dist[2] : p.dist[3];
                const double a = jl_min(thc * e_d[s], p.thk[Cj] * e_o[sj]);
                const double T = (p.kH * a) / (d_d[s] * vi);  // :427
                PUSH(wi, wi, T)
                PUSH(wi, wj, -T)
            }
        } else {
            const i64 s = (i64)c.j * p.nx + c.i;
            const double ar = p.area[s], ztk = p.zt[c.k], mld = p.ml[s];
            const double kap = (WHICH == COO_VML) ? p.kML : p.kDeep;
            const bool om = (WHICH == COO_VDEEP) || (ztk < mld);  // Ω[𝑖] (:85 / :109)
            if (om) {
#pragma unroll
                for (int d = 4; d < 6; ++d) {  // bottom, then top (:458, :468)
                    const i64 Cj = nb[d];
                    if (Cj < 0) continue;
                    const i64 wj = p.lw[Cj];
                    if (wj == 0) continue;
                    const double ztj = p.zt[(d == 4) ? c.k + 1 : c.k - 1];
                    if (WHICH == COO_VML && !(ztj < mld)) continue;  // Ω[𝑗]
                    const double T = (kap * ar) / (fabs(ztk - ztj) * vi);
                    PUSH(wi, wi, T)
                    PUSH(wi, wj, -T)
                }
            }
        }
#undef PUSH
        if (bad) raise_flag(p.flags, FLAG_FLUX_INTO_LAND);
        if (nanv) raise_flag(p.flags, WHICH == COO_ADV ? FLAG_TADV_NAN : WHICH == COO_H ? FLAG_TKH_NAN : WHICH == COO_VML ? FLAG_TKVML_NAN : FLAG_TKVDEEP_NAN);
    }
    unsigned incl = (unsigned)n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < COO_THREADS / 64; ++q) {
        const unsigned v = wave_tot[q];
        if (q < wid) before += v;
        all += v;
    }
    if (!WRITE) {
        if (tid == 0) tilesums[blockIdx.x] = all;
        return;
    }
    i64 q0 = tileoffs[blockIdx.x] + before + incl - (unsigned)n;
#pragma unroll
    for (int e = 0; e < 12; ++e)
        if (e < n) { Iout[q0 + e] = rows[e]; Jout[q0 + e] = cols[e]; Vout[q0 + e] = vals[e]; }
}

// ---- sparse(): after the stable sort by (col,row) --------------------------------------------------------
// sparse(I, J, V, m, n) demands 1 <= I <= m and 1 <= J <= n (SparseArrays throws an ArgumentError otherwise): checked here,
// before anything is packed into 32+32-bit keys or used as a colptr position; *bad counts the offending triplets.
// Keys are (column << rowbits) | row with rowbits = the bits m needs: the radix sort then runs over rowbits + bits(n) bits instead of 64 (44 for a
// 1 degree transport operator: six 8-bit passes instead of eight).
__global__ __launch_bounds__(256) void sp_keys_kernel(const i64 *__restrict__ I, const i64 *__restrict__ J, i64 len, i64 m, i64 n, int rowbits, u64 *keys,
                                                      u64 *idx, i64 *bad) {
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e < len) {
        const i64 i = I[e], j = J[e];
        const bool ok = i >= 1 && i <= m && j >= 1 && j <= n;
        keys[e] = ok ? (((u64)j << rowbits) | (u64)i) : ~0ull;
        idx[e] = (u64)e;
        if (!ok && *bad == 0) atomicAdd((unsigned long long *)bad, 1ull);
    }
}
// colptr over runs of EMPTY columns: the head that follows a run owns it.  Short runs it writes itself; a long one (a matrix whose entries sit in a
// few columns -- TκVML lives in the mixed layer only: one thread used to walk two million empty columns, 26 ms of a 27 ms call) goes to a list of
// (first column, last column, value) pieces of at most SP_GAP_PIECE columns that sp_fill_gaps_kernel fills with whole workgroups.
#define SP_GAP_SELF 16
#define SP_GAP_PIECE 65536
struct SpGap { i64 lo, hi, val; };
__device__ __forceinline__ void sp_colptr_run(i64 lo, i64 hi, i64 val, i64 *__restrict__ colptr, unsigned long long *__restrict__ ngaps, SpGap *__restrict__ gaps) {
    if (hi - lo < SP_GAP_SELF) {
        for (i64 c = lo; c <= hi; ++c) colptr[c - 1] = val;
        return;
    }
    for (i64 a = lo; a <= hi; a += SP_GAP_PIECE) {
        const i64 b = (a + SP_GAP_PIECE - 1 < hi) ? a + SP_GAP_PIECE - 1 : hi;
        const unsigned long long slot = atomicAdd(ngaps, 1ull);
        gaps[slot].lo = a; gaps[slot].hi = b; gaps[slot].val = val;
    }
}
__global__ __launch_bounds__(256) void sp_fill_gaps_kernel(const unsigned long long *__restrict__ ngaps, const SpGap *__restrict__ gaps, i64 *__restrict__ colptr) {
    const unsigned long long n = *ngaps;
    for (unsigned long long g = blockIdx.x; g < n; g += gridDim.x) {
        const i64 lo = gaps[g].lo, hi = gaps[g].hi, val = gaps[g].val;
        for (i64 c = lo + threadIdx.x; c <= hi; c += 256) colptr[c - 1] = val;
    }
}
template <bool WRITE>
__global__ __launch_bounds__(256) void sp_heads_kernel(const u64 *__restrict__ keys, const u64 *__restrict__ idx, const double *__restrict__ V,
                                                       i64 len, uint32_t *__restrict__ tilesums, const i64 *__restrict__ tileoffs,
                                                       i64 n, i64 *__restrict__ colptr, i64 *__restrict__ rowval, double *__restrict__ nzval,
                                                       unsigned long long *__restrict__ ngaps, SpGap *__restrict__ gaps, int rowbits) {
    __shared__ unsigned wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const i64 e = (i64)blockIdx.x * 256 + tid;
    const bool head = e < len && (e == 0 || keys[e] != keys[e - 1]);
    const u64 b = __ballot(head);
    const unsigned inwave = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wid] = __popcll(b);
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const unsigned v = wave_tot[q]; if (q < wid) before += v; all += v; }
    if (!WRITE) {
        if (tid == 0) tilesums[blockIdx.x] = all;
        return;
    }
    if (head) {
        const i64 q = tileoffs[blockIdx.x] + before + inwave;  // entries before this one
        const u64 key = keys[e];
        double acc = V[idx[e]];  // first touch copies
        for (i64 f = e + 1; f < len && keys[f] == key; ++f) acc = acc + V[idx[f]];  // then combine in input order
        rowval[q] = (i64)(key & ((1ull << rowbits) - 1ull));
        nzval[q] = acc;
        // colptr: every column from the previous head's column + 1 up to this one starts at q + 1
        const i64 col = (i64)(key >> rowbits);
        const i64 pcol = (e == 0) ? 0 : (i64)(keys[e - 1] >> rowbits);
        if (col > pcol) sp_colptr_run(pcol + 1, col, q + 1, colptr, ngaps, gaps);
    }
    if (e == len - 1 || (len == 0 && e == 0)) {
        const i64 lastcol = (len == 0) ? 0 : (i64)(keys[len - 1] >> rowbits);
        const i64 nnz = (len == 0) ? 0 : tileoffs[blockIdx.x] + before + inwave + (head ? 1 : 0);
        sp_colptr_run(lastcol + 1, n + 1, nnz + 1, colptr, ngaps, gaps);
    }
}


static void coo_params(TmParams &p, const otmb_tm_args &a, otmb_ctx *ctx) {
    memset(&p, 0, sizeof p);
    for (int f = 0; f < 6; ++f) p.phi[f] = a.phi[f];
    p.v = a.v3d; p.thk = a.thkcello; p.rho = a.rho; p.rho_s = a.rho_scalar;
    p.lw = (const i64 *)a.lwet3d; p.lwet = (const i64 *)a.lwet;
    for (int d = 0; d < 4; ++d) { p.edge[d] = a.edge_length[d]; p.dist[d] = a.dist_nbr[d]; }
    p.area = a.area2d; p.zt = a.zt; p.ml = a.mlotst;
    p.kH = a.kappa_h; p.kML = a.kappa_vml; p.kDeep = a.kappa_vdeep;
    p.nx = (int)a.nx; p.ny = (int)a.ny; p.nz = (int)a.nz; p.topo = a.topology; p.upwind = a.upwind;
    p.P = a.nx * a.ny; p.G = p.P * a.nz; p.n_own = a.n_wet;
    p.flags = (int *)ctx->flags.p;
}

template <bool WRITE>
static void coo_launch(int which, unsigned nt, hipStream_t s, const TmParams &p, uint32_t *sums, const i64 *offs, i64 *I, i64 *J, double *V) {
    switch (which) {
        case COO_ADV: hipLaunchKernelGGL((coo_kernel<COO_ADV, WRITE>), dim3(nt), dim3(COO_THREADS), 0, s, p, sums, offs, I, J, V); break;
        case COO_H: hipLaunchKernelGGL((coo_kernel<COO_H, WRITE>), dim3(nt), dim3(COO_THREADS), 0, s, p, sums, offs, I, J, V); break;
        case COO_VML: hipLaunchKernelGGL((coo_kernel<COO_VML, WRITE>), dim3(nt), dim3(COO_THREADS), 0, s, p, sums, offs, I, J, V); break;
        default: hipLaunchKernelGGL((coo_kernel<COO_VDEEP, WRITE>), dim3(nt), dim3(COO_THREADS), 0, s, p, sums, offs, I, J, V); break;
    }
}

static int32_t flags_to_status(otmb_ctx *ctx) {
    const int *f = ctx->h_flags;
    if (f[FLAG_RHO_NAN]) return otmb_fail(ctx, OTMB_ERR_RHO_NAN);
    if (f[FLAG_FLUX_INTO_LAND]) return otmb_fail(ctx, OTMB_ERR_FLUX_INTO_LAND);
    if (f[FLAG_TADV_NAN]) return otmb_fail(ctx, OTMB_ERR_TADV_NAN);
    if (f[FLAG_TKH_NAN]) return otmb_fail(ctx, OTMB_ERR_TKH_NAN);
    if (f[FLAG_TKVML_NAN]) return otmb_fail(ctx, OTMB_ERR_TKVML_NAN);
    if (f[FLAG_TKVDEEP_NAN]) return otmb_fail(ctx, OTMB_ERR_TKVDEEP_NAN);
    return OTMB_OK;
}

extern "C" {

// which: 0 advection, 1 horizontal diffusion, 2 vertical diffusion in the mixed layer, 3 background vertical diffusion.
// plan -> number of triplets; fill -> I, J, V (device, length len) in the reference's push order.
int32_t otmb_sparse_entries_plan_dev(otmb_ctx *ctx, int32_t which, const otmb_tm_args *a, int64_t *len) {
    if (!ctx || !a || !len || which < 0 || which > 3) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "argument");
    if (a->topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);
    if (a->n_wet > 0 && !a->lwet) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "lwet");
    if (!a->rho && a->rho_scalar != a->rho_scalar && which == COO_ADV && a->n_wet > 0) return otmb_fail(ctx, OTMB_ERR_RHO_NAN);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 nt = (a->n_wet + COO_THREADS - 1) / COO_THREADS;
    int32_t rc;
    if ((rc = otmb_reserve(ctx, ctx->blocksums, (size_t)(nt + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->blockoffs, (size_t)(nt + 1) * sizeof(i64) + otmb_scan_scratch(nt, 1)))) return rc;
    TmParams p;
    coo_params(p, *a, ctx);
    int *dflags = (int *)ctx->flags.p;
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS) + 10;
    HIP_TRY(ctx, hipMemsetAsync(dflags, 0, OTMB_NFLAGS_TM * sizeof(int), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dtot, 0, sizeof(i64), ctx->stream));
    if (nt > 0) {
        coo_launch<false>(which, (unsigned)nt, ctx->stream, p, (uint32_t *)ctx->blocksums.p, nullptr, nullptr, nullptr, nullptr);
        otmb_launch_tilescan(ctx->stream, (const uint32_t *)ctx->blocksums.p, (i64 *)ctx->blockoffs.p, dtot, nt, 1, (i64 *)ctx->blockoffs.p + nt + 1);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_NFLAGS_TM * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot + 10, dtot, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = flags_to_status(ctx))) return rc;
    ctx->coo.which = which; ctx->coo.args = *a; ctx->coo.ntiles = nt; ctx->coo.len = ctx->h_tot[10];
    *len = ctx->coo.len;
    return OTMB_OK;
}

int32_t otmb_sparse_entries_fill_dev(otmb_ctx *ctx, int64_t *I, int64_t *J, double *V) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    if (ctx->coo.which < 0) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    if (ctx->coo.len > 0 && (!I || !J || !V)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    TmParams p;
    coo_params(p, ctx->coo.args, ctx);
    if (ctx->coo.ntiles > 0)
        coo_launch<true>(ctx->coo.which, (unsigned)ctx->coo.ntiles, ctx->stream, p, nullptr, (const i64 *)ctx->blockoffs.p, (i64 *)I, (i64 *)J, V);
    HIP_TRY(ctx, hipGetLastError());
    ctx->coo.which = -1;
    return OTMB_OK;
}

// sparse(I, J, V, m, n): plan -> nnz (sorts; keeps the sorted keys in the context), fill -> colptr (n+1), rowval, nzval.
int32_t otmb_sparse_plan_dev(otmb_ctx *ctx, const int64_t *I, const int64_t *J, const double *V, int64_t len, int64_t m, int64_t n,
                             int64_t *nnz) {
    if (!ctx || !nnz || len < 0 || m < 0 || n < 0 || (len > 0 && (!I || !J || !V))) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "argument");
    if (m >= (1ll << 32) || n >= (1ll << 32)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "matrix too large for 32+32-bit sort keys");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    const size_t L = (size_t)(len > 0 ? len : 1);
    if ((rc = otmb_reserve(ctx, ctx->sort[0], L * 8))) return rc;  // keys in
    if ((rc = otmb_reserve(ctx, ctx->sort[1], L * 8))) return rc;  // keys out
    if ((rc = otmb_reserve(ctx, ctx->sort[2], L * 8))) return rc;  // idx in
    if ((rc = otmb_reserve(ctx, ctx->sort[3], L * 8))) return rc;  // idx out
    u64 *k0 = (u64 *)ctx->sort[0].p, *k1 = (u64 *)ctx->sort[1].p, *v0 = (u64 *)ctx->sort[2].p, *v1 = (u64 *)ctx->sort[3].p;
    const i64 nt = (len + 255) / 256;
    if ((rc = otmb_reserve(ctx, ctx->blocksums, (size_t)(nt + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->blockoffs, (size_t)(nt + 1) * sizeof(i64) + otmb_scan_scratch(nt, 1)))) return rc;
    i64 *dtot = (i64 *)((int *)ctx->flags.p + OTMB_NFLAGS) + 11;  // [11] nnz, [12] triplets with an index out of range
    ctx->sp.len = -1;                                               // no valid plan until this one succeeds
    HIP_TRY(ctx, hipMemsetAsync(dtot, 0, 2 * sizeof(i64), ctx->stream));
    int rowbits = 1, colbits = 1;
    while (rowbits < 32 && (m >> rowbits) != 0) ++rowbits;  // 1 <= row <= m < 2^rowbits
    while (colbits < 32 && (n >> colbits) != 0) ++colbits;
    if (len > 0) {
        hipLaunchKernelGGL(sp_keys_kernel, dim3((unsigned)nt), dim3(256), 0, ctx->stream, (const i64 *)I, (const i64 *)J, (i64)len, (i64)m, (i64)n, rowbits,
                           k0, v0, dtot + 1);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot + 12, dtot + 1, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_tot[12] != 0)
            return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "ArgumentError: row indices I[k] must satisfy 1 <= I[k] <= m and column indices J[k] 1 <= J[k] <= n");
        size_t tmp = 0;
        if (rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, (size_t)len, 0, (unsigned)(rowbits + colbits), ctx->stream) != hipSuccess) return otmb_fail(ctx, OTMB_ERR_HIP, "radix_sort_pairs (size)");
        if ((rc = otmb_reserve(ctx, ctx->sort[4], tmp + 16))) return rc;
        if (rocprim::radix_sort_pairs(ctx->sort[4].p, tmp, k0, k1, v0, v1, (size_t)len, 0, (unsigned)(rowbits + colbits), ctx->stream) != hipSuccess) return otmb_fail(ctx, OTMB_ERR_HIP, "radix_sort_pairs");
        hipLaunchKernelGGL(sp_heads_kernel<false>, dim3((unsigned)nt), dim3(256), 0, ctx->stream, (const u64 *)k1, (const u64 *)v1, V, (i64)len,
                           (uint32_t *)ctx->blocksums.p, (const i64 *)nullptr, (i64)n, (i64 *)nullptr, (i64 *)nullptr, (double *)nullptr,
                           (unsigned long long *)nullptr, (SpGap *)nullptr, rowbits);
        otmb_launch_tilescan(ctx->stream, (const uint32_t *)ctx->blocksums.p, (i64 *)ctx->blockoffs.p, dtot, nt, 1, (i64 *)ctx->blockoffs.p + nt + 1);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot + 11, dtot, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sp.I = I; ctx->sp.J = J; ctx->sp.V = V; ctx->sp.len = len; ctx->sp.m = m; ctx->sp.n = n; ctx->sp.nnz = ctx->h_tot[11]; ctx->sp.rowbits = rowbits;
    *nnz = ctx->sp.nnz;
    return OTMB_OK;
}

int32_t otmb_sparse_fill_dev(otmb_ctx *ctx, int64_t *colptr, int64_t *rowval, double *nzval) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    if (ctx->sp.len < 0) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    if (!colptr || (ctx->sp.nnz > 0 && (!rowval || !nzval))) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 len = ctx->sp.len, nt = (len + 255) / 256;
    // the list of long runs of empty columns: at most one per head (+ the tail) and at most (n + 1) / SP_GAP_SELF, each cut into pieces
    const i64 ncol = ctx->sp.n + 1;
    const i64 runs = ((len + 1 < ncol / SP_GAP_SELF + 1) ? len + 1 : ncol / SP_GAP_SELF + 1);
    const size_t gap_bytes = 16 + (size_t)(runs + ncol / SP_GAP_PIECE + 2) * sizeof(SpGap);
    int32_t rc;
    if ((rc = otmb_reserve(ctx, ctx->sort[0], gap_bytes))) return rc;  // (the unsorted keys are no longer needed)
    unsigned long long *ngaps = (unsigned long long *)ctx->sort[0].p;
    SpGap *gaps = (SpGap *)((char *)ctx->sort[0].p + 16);
    HIP_TRY(ctx, hipMemsetAsync(ngaps, 0, 16, ctx->stream));
    hipLaunchKernelGGL(sp_heads_kernel<true>, dim3((unsigned)(nt > 0 ? nt : 1)), dim3(256), 0, ctx->stream, (const u64 *)ctx->sort[1].p,
                       (const u64 *)ctx->sort[3].p, ctx->sp.V, len, (uint32_t *)nullptr, (const i64 *)ctx->blockoffs.p, ctx->sp.n, (i64 *)colptr,
                       (i64 *)rowval, nzval, ngaps, gaps, ctx->sp.rowbits);
    hipLaunchKernelGGL(sp_fill_gaps_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const unsigned long long *)ngaps, (const SpGap *)gaps, (i64 *)colptr);
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}

}  // extern "C"
