// otmb_transportmatrix.hip -- fused assembly of (T, Tadv, TκH, TκVML, TκVdeep) in CSC.
//
// Replaces, on the device, the whole of `transportmatrix` (src/matrixbuilding.jl:128-150): the three
// COO generators, the four sparse() calls and the three sparse adds.  No COO is materialised.
//
// Work decomposition: one thread per WET cell (driven by Lwet, so every lane works), one workgroup
// (tile) per 256 consecutive wet cells = 256 consecutive columns of all five matrices.  Per tile:
//   1. every thread builds its column in registers (otmb_tm_column.h);
//   2. a packed 64-bit block scan gives each column's offset inside the tile for the five matrices;
//   3. the tile's global offsets come from
//        COUNT + tile scan + FILL   (two-phase C ABI: the caller allocates after plan), or
//        ONEPASS                    (device-resident callers with known capacity): decoupled look-back
//                                   over per-tile status words, so inputs are read once and outputs
//                                   written once -- the algorithmic HBM traffic;
//   4. entries are staged through LDS and streamed out with fully coalesced 8-byte-per-lane stores
//      (a column's entries are contiguous, a tile's columns are contiguous).
#include "otmb_tm_column.h"

#define TM_THREADS 256
#define TM_NF 5
#define TM_MAXROWS 7  // rows per column: A, S, W, SELF, E, N|fold, B
#define TM_STAGE (TM_THREADS * TM_MAXROWS)

enum { MODE_COUNT = 0, MODE_FILL = 1, MODE_ONEPASS = 2 };

struct TmPlan {
    otmb_tm_args args;  // device pointers
    i64 ntiles = 0;
    i64 nnz[5] = {0, 0, 0, 0, 0};
    bool valid = false;
    bool onepass_pending = false;
    i64 wet_base = 0;
    i64 nnz_base[5] = {0, 0, 0, 0, 0};
};

// look-back status word: [63:62] flag (0 empty, 1 tile aggregate, 2 inclusive prefix), [61:0] value.
// One naturally aligned 8-byte word written by ONE agent-scope store and polled with agent-scope loads:
// data and tag travel together, so no fence is needed and nothing depends on workgroup placement.
#define ST_AGG (1ull << 62)
#define ST_PFX (2ull << 62)
#define ST_VAL(x) ((x) & ((1ull << 62) - 1))
#define LOOKBACK_SPIN_LIMIT (1 << 22)

__device__ __forceinline__ void st_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 st_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(TM_THREADS) void tm_kernel(const TmParams p) {
    __shared__ u64 wave_tot[TM_THREADS / 64];
    __shared__ i64 s_prefix[TM_NF];
    __shared__ int s_tile;
    __shared__ i64 s_row[TM_STAGE];
    __shared__ double s_val[TM_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    i64 tile = blockIdx.x;
    if (MODE == MODE_ONEPASS) {
        // dynamic tile id: tiles start in ticket order, so every predecessor a tile waits for is already
        // running or finished whatever order the hardware dispatches workgroups in
        if (tid == 0) s_tile = atomicAdd(p.ticket, 1);
        __syncthreads();
        tile = s_tile;
    }
    const i64 w0 = tile * TM_THREADS;
    const i64 w = w0 + tid;
    const bool valid = w < p.n_own;

    // tile-uniform base pointers: all neighbours of all cells of the tile sit at non-negative 32-bit
    // byte offsets from them
    const i64 Lmin = p.lwet[w0] - 1;
    const i64 wlast = (w0 + TM_THREADS - 1 < p.n_own) ? w0 + TM_THREADS - 1 : p.n_own - 1;
    const i64 Lmax = p.lwet[wlast] - 1;
    const i64 base_elem = (Lmin > p.P) ? Lmin - p.P : 0;
    const bool span_ok = (Lmax + p.P - base_elem) < (1ll << 28) && Lmin >= 0 && Lmax < p.G && Lmin <= Lmax;
    if (!span_ok && tid == 0) raise_flag(p.flags, FLAG_NONCANONICAL);
    TileBase tb;
    tb.lw = (const char *)(p.lw + base_elem);
    tb.v = (const char *)(p.v + base_elem);
    tb.thk = (const char *)(p.thk + base_elem);
    tb.rho = p.rho ? (const char *)(p.rho + base_elem) : nullptr;
    tb.pe = (const char *)(p.phi[OTMB_EAST] + base_elem);
    tb.pw = (const char *)(p.phi[OTMB_WEST] + base_elem);
    tb.pn = (const char *)(p.phi[OTMB_NORTH] + base_elem);
    tb.ps = (const char *)(p.phi[OTMB_SOUTH] + base_elem);
    tb.pt = (const char *)(p.phi[OTMB_TOP] + base_elem);
    tb.pb = (const char *)(p.phi[OTMB_BOTTOM] + base_elem);

    // ---- 1. the column ----
    Column col;
    unsigned pT = 0, nT = 0, nA = 0, nH = 0, nM = 0, nD = 0;
    bool live = false;
    if (valid && span_ok) {
        const i64 L = p.lwet[w] - 1;
        const i64 Lnext = (w + 1 < p.n_own) ? p.lwet[w + 1] - 1 : p.G;
        const i64 c = p.wet_base + w + 1;  // this column's (global) wet rank
        // Lwet ascending inside [Lmin, Lmax] and Lwet3D[Lwet[w]] == w + 1: together they make the wet
        // rank monotone in the linear index, which is what orders the rows of a column
        if (L < Lmin || L > Lmax || Lnext <= L) {
            raise_flag(p.flags, FLAG_NONCANONICAL);
        } else {
            const Cell cell = cell_of(L, p.nx, p.ny, p.P);
            const unsigned oC = (unsigned)(L - base_elem) * 8u;
            if (ldi(tb.lw, oC) != c) {
                raise_flag(p.flags, FLAG_NONCANONICAL);
            } else {
                const bool regular = (p.nx >= 3) && !(p.topo == OTMB_TRIPOLAR && cell.j == p.ny - 1);
                if (regular) fast_column(p, tb, oC, cell.i, cell.j, cell.k, c, col);
                else build_column(p, cell, c, col);
                live = true;
                const unsigned uni = col.padv | col.phh | col.pml | col.pdp;
#pragma unroll
                for (int s = 0; s < NSLOT; ++s)
                    if (((uni >> s) & 1u) && t_value(col, s) != 0.0) pT |= 1u << s;
                nT = __popc(pT); nA = __popc(col.padv); nH = __popc(col.phh); nM = __popc(col.pml); nD = __popc(col.pdp);
            }
        }
    }

    // ---- 2. packed block scan: T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10 bits ----
    const u64 mine = (u64)nT | ((u64)nA << 11) | ((u64)nH << 22) | ((u64)nM << 33) | ((u64)nD << 43);
    u64 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    u64 before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < TM_THREADS / 64; ++q) {
        const u64 v = wave_tot[q];
        if (q < wid) before += v;
        all += v;
    }
    const u64 excl = before + incl - mine;
    const unsigned ex[5] = {(unsigned)(excl & 0x7ff), (unsigned)((excl >> 11) & 0x7ff), (unsigned)((excl >> 22) & 0x7ff),
                            (unsigned)((excl >> 33) & 0x3ff), (unsigned)((excl >> 43) & 0x3ff)};
    const unsigned agg[5] = {(unsigned)(all & 0x7ff), (unsigned)((all >> 11) & 0x7ff), (unsigned)((all >> 22) & 0x7ff),
                             (unsigned)((all >> 33) & 0x3ff), (unsigned)((all >> 43) & 0x3ff)};

    if (MODE == MODE_COUNT) {
        if (tid < TM_NF) {
            unsigned a = 0;
#pragma unroll
            for (int m = 0; m < TM_NF; ++m)
                if (m == tid) a = agg[m];
            p.tilesums[tile * TM_NF + tid] = a;
        }
        return;
    }

    // ---- 3. the tile's global offsets ----
    if (MODE == MODE_FILL) {
        if (tid < TM_NF) s_prefix[tid] = p.tileoffs[tile * TM_NF + tid];
    } else {
        if (wid == 0) {
#pragma unroll
            for (int m = 0; m < TM_NF; ++m) {
                u64 *st = p.status + m;  // status[t * TM_NF + m]
                if (lane == 0) st_store(st + tile * TM_NF, (tile == 0 ? ST_PFX : ST_AGG) | (u64)agg[m]);
                i64 excl_prefix = 0;
                i64 look = tile - 1;  // nearest predecessor not yet accounted for
                int spins = 0;
                while (look >= 0) {
                    const i64 t = look - lane;  // lane 0 inspects the nearest predecessor
                    u64 sw = ST_PFX;             // positions before tile 0 read as "prefix 0"
                    if (t >= 0) {
                        sw = st_load(st + t * TM_NF);
                        while ((sw >> 62) == 0 && spins < LOOKBACK_SPIN_LIMIT) {
                            __builtin_amdgcn_s_sleep(1);
                            sw = st_load(st + t * TM_NF);
                            ++spins;
                        }
                    }
                    if (__any((sw >> 62) == 0)) {  // bounded spin expired: report, never hang
                        if (lane == 0) raise_flag(p.flags, FLAG_LOOKBACK_TIMEOUT);
                        break;
                    }
                    const u64 is_pfx = __ballot((sw >> 62) == 2);
                    // lanes up to and including the first one holding an inclusive prefix contribute
                    const int first = is_pfx ? __builtin_ctzll(is_pfx) : 64;
                    i64 contrib = (lane <= first) ? (i64)ST_VAL(sw) : 0;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) contrib += __shfl_xor(contrib, d);
                    excl_prefix += contrib;
                    if (is_pfx) break;
                    look -= 64;
                }
                if (lane == 0) {
                    if (tile != 0) st_store(st + tile * TM_NF, ST_PFX | (u64)(excl_prefix + agg[m]));
                    s_prefix[m] = excl_prefix;
                }
            }
        }
    }
    __syncthreads();
    i64 g0[5];
#pragma unroll
    for (int m = 0; m < TM_NF; ++m) g0[m] = s_prefix[m];  // entries of matrix m before this tile (this launch)

    if (MODE == MODE_ONEPASS && w0 + TM_THREADS >= p.n_own) {  // last tile: totals and the closing colptr entry
        if (tid < TM_NF) {
            i64 tot = 0;
#pragma unroll
            for (int m = 0; m < TM_NF; ++m)
                if (m == tid) tot = g0[m] + agg[m];
            p.totals[tid] = tot;
            i64 *cp = (tid == 0) ? p.colptr[0] : (tid == 1) ? p.colptr[1] : (tid == 2) ? p.colptr[2] : (tid == 3) ? p.colptr[3] : p.colptr[4];
            const i64 nb = (tid == 0) ? p.nnz_base[0] : (tid == 1) ? p.nnz_base[1] : (tid == 2) ? p.nnz_base[2] : (tid == 3) ? p.nnz_base[3] : p.nnz_base[4];
            cp[p.n_own] = nb + tot + 1;
        }
    }

    // ---- 4. write: colptr, then LDS-staged entries ----
    if (live) {
#pragma unroll
        for (int m = 0; m < TM_NF; ++m) p.colptr[m][w] = p.nnz_base[m] + g0[m] + ex[m] + 1;
    }
    const unsigned pm[5] = {pT, col.padv, col.phh, col.pml, col.pdp};
#pragma unroll
    for (int m = 0; m < TM_NF; ++m) {
        if (live) {
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                if ((pm[m] >> s) & 1u) {
                    const unsigned q = ex[m] + __popc(pm[m] & col.bef[s]);
                    s_row[q] = col.idx[s];
                    s_val[q] = (m == 0) ? t_value(col, s) : (m == 1) ? col.adv[s] : (m == 2) ? col.hh[s] : (m == 3) ? col.ml[s] : col.dp[s];
                }
            }
        }
        __syncthreads();
        const unsigned cnt = agg[m];
        bool room = true;
        if (MODE == MODE_ONEPASS) {
            room = g0[m] + cnt <= p.cap[m];
            if (!room && tid == 0) raise_flag(p.flags, FLAG_CAPACITY);
        }
        if (room) {
            i64 *rv = p.rowval[m] + g0[m];
            double *nz = p.nzval[m] + g0[m];
            for (unsigned e = tid; e < cnt; e += TM_THREADS) {
                rv[e] = s_row[e];
                nz[e] = s_val[e];
            }
        }
        __syncthreads();
    }
}

__global__ void tm_finish_colptr(i64 *c0, i64 *c1, i64 *c2, i64 *c3, i64 *c4, const i64 *tot, i64 N, i64 b0,
                                 i64 b1, i64 b2, i64 b3, i64 b4) {
    if (threadIdx.x == 0) {
        c0[N] = b0 + tot[0] + 1; c1[N] = b1 + tot[1] + 1; c2[N] = b2 + tot[2] + 1; c3[N] = b3 + tot[3] + 1;
        c4[N] = b4 + tot[4] + 1;
    }
}

// ---- host side ------------------------------------------------------------------------------
static void fill_params(TmParams &p, const otmb_tm_args &a, otmb_ctx *ctx, const TmPlan *pl) {
    memset(&p, 0, sizeof p);
    for (int f = 0; f < 6; ++f) p.phi[f] = a.phi[f];
    p.v = a.v3d; p.thk = a.thkcello; p.rho = a.rho; p.rho_s = a.rho_scalar;
    p.lw = (const i64 *)a.lwet3d; p.lwet = (const i64 *)a.lwet;
    for (int d = 0; d < 4; ++d) { p.edge[d] = a.edge_length[d]; p.dist[d] = a.dist_nbr[d]; }
    p.area = a.area2d; p.zt = a.zt; p.ml = a.mlotst;
    p.kH = a.kappa_h; p.kML = a.kappa_vml; p.kDeep = a.kappa_vdeep;
    p.nx = (int)a.nx; p.ny = (int)a.ny; p.nz = (int)a.nz; p.topo = a.topology; p.upwind = a.upwind;
    p.P = a.nx * a.ny; p.G = p.P * a.nz;
    p.n_own = a.n_wet;
    if (pl) {
        p.wet_base = pl->wet_base;
        for (int m = 0; m < 5; ++m) p.nnz_base[m] = pl->nnz_base[m];
    }
    p.tilesums = (uint32_t *)ctx->blocksums.p;
    p.tileoffs = (const i64 *)ctx->blockoffs.p;
    p.flags = (int *)ctx->flags.p;
}

static int32_t check_flags(otmb_ctx *ctx) {
    const int *f = ctx->h_flags;
    if (f[FLAG_NONCANONICAL]) return otmb_fail(ctx, OTMB_ERR_NONCANONICAL_INDICES);
    if (f[FLAG_LOOKBACK_TIMEOUT]) return otmb_fail(ctx, OTMB_ERR_HIP, "look-back spin limit reached");
    if (f[FLAG_RHO_NAN]) return otmb_fail(ctx, OTMB_ERR_RHO_NAN);  // reference order: :233, loop, :39, :61, :90, :114
    if (f[FLAG_FLUX_INTO_LAND]) return otmb_fail(ctx, OTMB_ERR_FLUX_INTO_LAND);
    if (f[FLAG_TADV_NAN]) return otmb_fail(ctx, OTMB_ERR_TADV_NAN);
    if (f[FLAG_TKH_NAN]) return otmb_fail(ctx, OTMB_ERR_TKH_NAN);
    if (f[FLAG_TKVML_NAN]) return otmb_fail(ctx, OTMB_ERR_TKVML_NAN);
    if (f[FLAG_TKVDEEP_NAN]) return otmb_fail(ctx, OTMB_ERR_TKVDEEP_NAN);
    if (f[FLAG_CAPACITY]) return otmb_fail(ctx, OTMB_ERR_CAPACITY);
    return OTMB_OK;
}

static int32_t validate_args(otmb_ctx *ctx, const otmb_tm_args *a) {
    if (a->nx < 1 || a->ny < 1 || a->nz < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    const i64 G = a->nx * a->ny * a->nz;
    if (a->nx * a->ny >= (1ll << 27) || G >= (1ll << 32)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid too large");
    if (a->topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);
    if (a->topology != OTMB_BIPOLAR && a->topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "topology");
    for (int f = 0; f < 6; ++f)
        if (!a->phi[f]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "phi");
    for (int d = 0; d < 4; ++d)
        if (!a->edge_length[d] || !a->dist_nbr[d]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "metrics");
    if (!a->v3d || !a->thkcello || !a->lwet3d || !a->area2d || !a->zt || !a->mlotst)
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null input array");
    if (a->n_wet < 0 || a->n_wet > G || (a->n_wet > 0 && !a->lwet)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "lwet / n_wet");
    if (!a->rho && a->rho_scalar != a->rho_scalar && a->n_wet > 0) return otmb_fail(ctx, OTMB_ERR_RHO_NAN);  // :233
    return OTMB_OK;
}

void otmb_tm_plan_free(otmb_ctx *ctx) {
    delete ctx->plan;
    ctx->plan = nullptr;
}

int32_t otmb_tm_plan_query(otmb_ctx *ctx, int64_t *nnz, int64_t *N) {
    if (!ctx->plan || !ctx->plan->valid) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    for (int m = 0; m < 5; ++m) nnz[m] = ctx->plan->nnz[m];
    *N = ctx->plan->args.n_wet;
    return OTMB_OK;
}

extern "C" {

int32_t otmb_transportmatrix_plan_dev(otmb_ctx *ctx, const otmb_tm_args *a, int64_t nnz[5]) {
    if (!ctx || !a || !nnz) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (ctx->plan) ctx->plan->valid = false;
    int32_t rc;
    if ((rc = validate_args(ctx, a))) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 ntiles = (a->n_wet + TM_THREADS - 1) / TM_THREADS;
    if ((rc = otmb_reserve(ctx, ctx->blocksums, (size_t)(ntiles + 1) * TM_NF * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->blockoffs, (size_t)(ntiles + 1) * TM_NF * sizeof(i64)))) return rc;
    if (!ctx->plan) ctx->plan = new TmPlan();
    TmPlan &pl = *ctx->plan;
    pl.args = *a;
    pl.ntiles = ntiles;
    TmParams p;
    fill_params(p, *a, ctx, &pl);
    int *dflags = (int *)ctx->flags.p;
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS);
    HIP_TRY(ctx, hipMemsetAsync(dflags, 0, OTMB_NFLAGS * sizeof(int) + 16 * sizeof(i64), ctx->stream));
    if (ntiles > 0) {
        {
            KernelTimer kt(ctx, K_TM_COUNT);
            hipLaunchKernelGGL(tm_kernel<MODE_COUNT>, dim3((unsigned)ntiles), dim3(TM_THREADS), 0, ctx->stream, p);
        }
        {
            KernelTimer kt(ctx, K_TILESCAN);
            otmb_launch_tilescan(ctx->stream, p.tilesums, (i64 *)ctx->blockoffs.p, dtot, ntiles, TM_NF);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_NFLAGS * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot, dtot, TM_NF * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = check_flags(ctx))) return rc;
    for (int m = 0; m < 5; ++m) nnz[m] = pl.nnz[m] = ctx->h_tot[m];
    pl.valid = true;
    return OTMB_OK;
}

int32_t otmb_transportmatrix_set_slab(otmb_ctx *ctx, int64_t wet_base) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    if (!ctx->plan) ctx->plan = new TmPlan();
    ctx->plan->valid = false;
    ctx->plan->wet_base = wet_base;
    for (int m = 0; m < 5; ++m) ctx->plan->nnz_base[m] = 0;
    return OTMB_OK;
}

int32_t otmb_transportmatrix_set_nnz_base(otmb_ctx *ctx, const int64_t nnz_base[5]) {
    if (!ctx || !nnz_base) return OTMB_ERR_INVALID_ARG;
    if (!ctx->plan) ctx->plan = new TmPlan();
    for (int m = 0; m < 5; ++m) ctx->plan->nnz_base[m] = nnz_base[m];
    return OTMB_OK;
}

int32_t otmb_transportmatrix_fill_dev(otmb_ctx *ctx, int64_t *const colptr[5], int64_t *const rowval[5],
                                      double *const nzval[5]) {
    if (!ctx || !colptr || !rowval || !nzval) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (!ctx->plan || !ctx->plan->valid) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    TmPlan &pl = *ctx->plan;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    TmParams p;
    fill_params(p, pl.args, ctx, &pl);
    for (int m = 0; m < 5; ++m) {
        if (!colptr[m] || (pl.nnz[m] > 0 && (!rowval[m] || !nzval[m]))) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
        p.colptr[m] = (i64 *)colptr[m]; p.rowval[m] = (i64 *)rowval[m]; p.nzval[m] = nzval[m];
    }
    int *dflags = (int *)ctx->flags.p;
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS);
    if (pl.ntiles > 0) {
        KernelTimer kt(ctx, K_TM_FILL);
        hipLaunchKernelGGL(tm_kernel<MODE_FILL>, dim3((unsigned)pl.ntiles), dim3(TM_THREADS), 0, ctx->stream, p);
    }
    {
        KernelTimer kt(ctx, K_TM_FINISH);
        hipLaunchKernelGGL(tm_finish_colptr, dim3(1), dim3(64), 0, ctx->stream, p.colptr[0], p.colptr[1], p.colptr[2],
                           p.colptr[3], p.colptr[4], dtot, (i64)pl.args.n_wet, p.nnz_base[0], p.nnz_base[1],
                           p.nnz_base[2], p.nnz_base[3], p.nnz_base[4]);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_NFLAGS * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return OTMB_OK;
}

// One pass, asynchronous: the caller provides output buffers of known capacity (nnz per column is at
// most 7, 7, 5, 3, 3 for T, Tadv, TκH, TκVML, TκVdeep).  Errors and the nnz are collected afterwards by
// otmb_transportmatrix_result (which synchronises).
int32_t otmb_transportmatrix_dev(otmb_ctx *ctx, const otmb_tm_args *a, int64_t *const colptr[5],
                                 int64_t *const rowval[5], double *const nzval[5], const int64_t capacity[5]) {
    if (!ctx || !a || !colptr || !rowval || !nzval || !capacity) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    int32_t rc;
    if ((rc = validate_args(ctx, a))) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 ntiles = (a->n_wet + TM_THREADS - 1) / TM_THREADS;
    // look-back words [ntiles][5] + the ticket, zeroed on the stream before every launch
    const size_t stbytes = (size_t)(ntiles + 1) * TM_NF * sizeof(u64);
    if ((rc = otmb_reserve(ctx, ctx->lookback, stbytes + 64))) return rc;
    if (!ctx->plan) ctx->plan = new TmPlan();
    TmPlan &pl = *ctx->plan;
    pl.valid = false;
    pl.args = *a;
    pl.ntiles = ntiles;
    TmParams p;
    fill_params(p, *a, ctx, &pl);
    for (int m = 0; m < 5; ++m) {
        if (!colptr[m] || !rowval[m] || !nzval[m]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
        p.colptr[m] = (i64 *)colptr[m]; p.rowval[m] = (i64 *)rowval[m]; p.nzval[m] = nzval[m];
        p.cap[m] = capacity[m];
    }
    p.status = (u64 *)ctx->lookback.p;
    p.ticket = (int *)((char *)ctx->lookback.p + stbytes);
    int *dflags = (int *)ctx->flags.p;
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS);
    p.totals = dtot;
    HIP_TRY(ctx, hipMemsetAsync(dflags, 0, OTMB_NFLAGS * sizeof(int) + 16 * sizeof(i64), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->lookback.p, 0, stbytes + 64, ctx->stream));
    if (ntiles > 0) {
        KernelTimer kt(ctx, K_TM_ONEPASS);
        hipLaunchKernelGGL(tm_kernel<MODE_ONEPASS>, dim3((unsigned)ntiles), dim3(TM_THREADS), 0, ctx->stream, p);
    } else {
        KernelTimer kt(ctx, K_TM_FINISH);
        hipLaunchKernelGGL(tm_finish_colptr, dim3(1), dim3(64), 0, ctx->stream, p.colptr[0], p.colptr[1], p.colptr[2],
                           p.colptr[3], p.colptr[4], dtot, (i64)0, p.nnz_base[0], p.nnz_base[1], p.nnz_base[2],
                           p.nnz_base[3], p.nnz_base[4]);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_NFLAGS * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot, dtot, TM_NF * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    pl.onepass_pending = true;
    return OTMB_OK;
}

int32_t otmb_transportmatrix_result(otmb_ctx *ctx, int64_t nnz[5]) {
    if (!ctx || !nnz) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (!ctx->plan || !ctx->plan->onepass_pending) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->plan->onepass_pending = false;
    int32_t rc;
    if ((rc = check_flags(ctx))) return rc;
    for (int m = 0; m < 5; ++m) nnz[m] = ctx->plan->nnz[m] = ctx->h_tot[m];
    return OTMB_OK;
}

}  // extern "C"
