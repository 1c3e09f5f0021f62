// otmb_lump.hip -- lump_and_spray(wet3D, vol, T, mask; di, dj, dk) on the device (src/extratools.jl:38-119): the
// coarsening operators LUMP (Nc x N, volume weighted), SPRAY (N x Nc, ones) and the coarse volumes.
//
// The reference sweeps all cells in linear order; an in-mask cell that is still unassigned grabs the di x dj x dk block
// it anchors, splits the block's wet cells into connected components of T's pattern and gives every component the
// next number; an out-of-mask cell gets a number of its own (:57-82).  Later assignments overwrite earlier ones.  What
// is sequential in that is only the question "which cells are anchors"; everything else follows from the anchors:
//   1. lump_sweep_kernel   anchors + the LAST anchor whose block covers a cell (its final owner).  Rows are taken in
//                          order (a row's anchors depend on the blocks of the rows and levels before it); inside a row
//                          the greedy "first free cell, skip di" chain is followed with pointer doubling.  With dk == 1
//                          blocks never cross levels and every level is an independent workgroup.
//   2. lump_components_kernel  one thread per anchor: min-label propagation over the block with T's columns as
//                          adjacency (Graphs.connected_components numbers components by ascending smallest vertex, and
//                          the label a component converges to IS its smallest vertex); symmetry of the local pattern
//                          is verified (Graphs.SimpleGraph throws otherwise).
//   3. exclusive scan of "numbers consumed at this cell" in linear order = the value of the reference's counter c
//   4. per wet cell: its number; flags + scan = rows that survive LUMP[wet_c, wet] (:88-91)
//   5. fill: LUMP is one entry per column; SPRAY = its transpose = stable sort of the wet cells by coarse row;
//      vol_c sums member volumes in ascending order like mul! does (:96); values ((1/vol_c)*1)*vol (:97).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "otmb_common.h"

#define LS_THREADS 256
#define LS_INF 0xFFFFu
#define LS_DRY 0xFFFFu
#define LS_MAX_BLOCK 4096  // di*dj*dk (labels are 16 bit, and one thread walks a block)

// ---- 1. anchors and owners -----------------------------------------------------------------------------------------
// LDS per workgroup: covj[nx] int (last row of this level covered at column i), four u16 arrays, reach[nx] u8
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(LS_THREADS) void lump_sweep_kernel(const uint8_t *__restrict__ mask, int nx, int ny, int nz, int di, int dj,
                                                                 int dk, int levels_per_group, uint8_t *__restrict__ isanchor,
                                                                 uint32_t *__restrict__ owner, int *covk) {
    extern __shared__ unsigned char lds[];
    int *covj = (int *)lds;
    unsigned short *bufA = (unsigned short *)(covj + nx), *bufB = bufA + nx, *bufC = bufB + nx, *bufD = bufC + nx;
    unsigned char *reach = (unsigned char *)(bufD + nx);
    const int tid = threadIdx.x;
    const i64 P = (i64)nx * ny;
    const int kbeg = blockIdx.x * levels_per_group, kend = min(nz, kbeg + levels_per_group);
    int rounds = 1;  // pointer doubling: a row holds at most ceil(nx/di) anchors
    while ((1 << rounds) < (nx + di - 1) / di + 1) ++rounds;
    for (int k = kbeg; k < kend; ++k) {
        for (int i = tid; i < nx; i += LS_THREADS) covj[i] = -1;
        __syncthreads();
        for (int j = 0; j < ny; ++j) {
            const i64 row = (i64)k * P + (i64)j * nx;
            // free cells of the row: in the mask and not inside a block anchored in an earlier row or level (:61)
            for (int i = tid; i < nx; i += LS_THREADS) {
                const bool inm = mask ? (mask[row + i] != 0) : true;
                const bool cov = (covj[i] >= j) || (dk > 1 && ld_agent(covk + (i64)j * nx + i) >= k);
                bufA[i] = (inm && !cov) ? (unsigned short)i : (unsigned short)LS_INF;
                reach[i] = 0;
            }
            __syncthreads();
            // suffix minimum: first[i] = first free cell at or after i
            unsigned short *src = bufA, *dst = bufB;
            for (int d = 1; d < nx; d <<= 1) {
                for (int i = tid; i < nx; i += LS_THREADS) {
                    const unsigned short a = src[i], b = (i + d < nx) ? src[i + d] : (unsigned short)LS_INF;
                    dst[i] = a < b ? a : b;
                }
                __syncthreads();
                unsigned short *t = src; src = dst; dst = t;
            }
            const unsigned short *first = src;
            // the greedy chain: an anchor at i makes the next one the first free cell at or after i + di
            unsigned short *Jc = bufC, *Jn = bufD;
            for (int i = tid; i < nx; i += LS_THREADS) Jc[i] = (i + di < nx) ? first[i + di] : (unsigned short)LS_INF;
            if (tid == 0 && first[0] != LS_INF) reach[first[0]] = 1;
            __syncthreads();
            for (int r = 0; r < rounds; ++r) {
                for (int i = tid; i < nx; i += LS_THREADS) {
                    const unsigned short t = Jc[i];
                    if (reach[i] && t != LS_INF) reach[t] = 1;  // monotone marks: a racing reader only sees them earlier
                    Jn[i] = (t != LS_INF) ? Jc[t] : (unsigned short)LS_INF;
                }
                __syncthreads();
                unsigned short *t2 = Jc; Jc = Jn; Jn = t2;
            }
            // anchors claim their blocks; a row's blocks are disjoint, later rows overwrite earlier owners (:75)
            for (int i = tid; i < nx; i += LS_THREADS) {
                if (!reach[i]) continue;
                const i64 L = row + i;
                isanchor[L] = 1;
                const int ie = min(nx, i + di), je = min(ny, j + dj), ke = min(nz, k + dk);
                for (int kk = k; kk < ke; ++kk)
                    for (int jj = j; jj < je; ++jj)
                        for (int ii = i; ii < ie; ++ii) owner[(i64)kk * P + (i64)jj * nx + ii] = (uint32_t)L;
                for (int ii = i; ii < ie; ++ii) covj[ii] = j + dj - 1;
                if (dk > 1)
                    for (int jj = j; jj < je; ++jj)
                        for (int ii = i; ii < ie; ++ii) st_agent(covk + (i64)jj * nx + ii, k + dk - 1);
            }
            __syncthreads();
        }
    }
}

// ---- 2. connected components of every anchor's block -----------------------------------------------------------------
struct LumpGrid {
    int nx, ny, nz, di, dj, dk;
    i64 P;
};
__device__ __forceinline__ int lump_local(const LumpGrid &g, i64 L, int ai, int aj, int ak) {  // local index in the block or -1
    const int k = (int)(L / g.P);
    const i64 r = L - (i64)k * g.P;
    const int j = (int)(r / g.nx), i = (int)(r - (i64)j * g.nx);
    const int a = i - ai, b = j - aj, c = k - ak;
    if (a < 0 || a >= g.di || b < 0 || b >= g.dj || c < 0 || c >= g.dk) return -1;
    return (c * g.dj + b) * g.di + a;
}
__device__ __forceinline__ bool lump_has_entry(const i64 *Tp, const i64 *Ti, i64 col, i64 row) {  // T[row, col] stored?
    for (i64 q = Tp[col - 1]; q < Tp[col]; ++q)
        if (Ti[q - 1] == row) return true;
    return false;
}

__global__ __launch_bounds__(LS_THREADS) void lump_components_kernel(LumpGrid g, i64 G, const uint8_t *__restrict__ isanchor,
                                                                      const i64 *__restrict__ arank, const uint8_t *__restrict__ wet,
                                                                      const i64 *__restrict__ lwet3d, const i64 *__restrict__ lwet,
                                                                      const i64 *__restrict__ Tp, const i64 *__restrict__ Ti,
                                                                      unsigned short *__restrict__ labels, uint32_t *__restrict__ ncomp,
                                                                      int *flags) {
    const i64 L0 = (i64)blockIdx.x * LS_THREADS + threadIdx.x;
    if (L0 >= G || !isanchor[L0]) return;
    const int bv = g.di * g.dj * g.dk;
    unsigned short *lab = labels + arank[L0] * bv;
    const int ak = (int)(L0 / g.P);
    const i64 r0 = L0 - (i64)ak * g.P;
    const int aj = (int)(r0 / g.nx), ai = (int)(r0 - (i64)aj * g.nx);
    // vertices = wet cells of the block inside the grid (ghost cells of the reference's extension are dry, :48)
    for (int c = 0; c < g.dk; ++c)
        for (int b = 0; b < g.dj; ++b)
            for (int a = 0; a < g.di; ++a) {
                const int l = (c * g.dj + b) * g.di + a;
                const bool inside = ai + a < g.nx && aj + b < g.ny && ak + c < g.nz;
                lab[l] = (inside && wet[L0 + a + (i64)b * g.nx + (i64)c * g.P]) ? (unsigned short)l : (unsigned short)LS_DRY;
            }
    bool asym = false;
    for (bool changed = true, first = true; changed; first = false) {
        changed = false;
        for (int c = 0; c < g.dk; ++c)
            for (int b = 0; b < g.dj; ++b)
                for (int a = 0; a < g.di; ++a) {
                    const int l = (c * g.dj + b) * g.di + a;
                    if (lab[l] == LS_DRY) continue;
                    const i64 col = lwet3d[L0 + a + (i64)b * g.nx + (i64)c * g.P];
                    unsigned short best = lab[l];
                    for (i64 q = Tp[col - 1]; q < Tp[col]; ++q) {
                        const i64 row = Ti[q - 1];
                        const int l2 = lump_local(g, lwet[row - 1] - 1, ai, aj, ak);
                        if (l2 < 0) continue;
                        if (first && !lump_has_entry(Tp, Ti, row, col)) asym = true;  // SimpleGraph(adjmx) wants symmetry
                        const unsigned short o = lab[l2];
                        if (o < best) best = o;
                    }
                    if (best < lab[l]) { lab[l] = best; changed = true; }
                }
    }
    if (asym && flags[0] == 0) atomicExch(&flags[0], 1);
    // component numbers in order of their smallest vertex (= the label they converged to)
    unsigned n = 0;
    for (int l = 0; l < bv; ++l)
        if (lab[l] == l) ++n;
    ncomp[L0] = n;
    // replace labels by component numbers: roots are visited in ascending order, so a root's number is known before any
    // member with a larger index needs it; members never precede their root
    unsigned next = 0;
    for (int l = 0; l < bv; ++l) {
        const unsigned short x = lab[l];
        if (x == LS_DRY) continue;
        if (x == l) lab[l] = (unsigned short)(0x8000u | next++);  // tag: already a component number
        else lab[l] = lab[x];                                      // x < l: its root has been rewritten already
    }
    for (int l = 0; l < bv; ++l)
        if (lab[l] != LS_DRY) lab[l] &= 0x7FFFu;
}

// numbers consumed when the sweep visits a cell: an anchor's components, one for an out-of-mask cell (:76, :80)
__global__ void lump_increment_kernel(i64 G, const uint8_t *__restrict__ mask, const uint8_t *__restrict__ isanchor,
                                      const uint32_t *__restrict__ ncomp, i64 *__restrict__ inc) {
    const i64 L = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= G) return;
    const bool inm = mask ? (mask[L] != 0) : true;
    inc[L] = inm ? (isanchor[L] ? (i64)ncomp[L] : 0) : 1;
}
__global__ void lump_anchor_flag_kernel(i64 G, const uint8_t *__restrict__ isanchor, i64 *__restrict__ out) {
    const i64 L = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L < G) out[L] = isanchor[L];
}

// ---- 4. the number of every wet cell, and which numbers hold wet cells -----------------------------------------------
__global__ void lump_number_kernel(LumpGrid g, i64 N, const i64 *__restrict__ lwet, const uint8_t *__restrict__ mask,
                                   const uint32_t *__restrict__ owner, const i64 *__restrict__ arank, const i64 *__restrict__ cbase,
                                   const unsigned short *__restrict__ labels, i64 *__restrict__ cnum, uint8_t *__restrict__ used) {
    const i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= N) return;
    const i64 L = lwet[w] - 1;
    i64 c;
    if (mask && !mask[L]) {
        c = 2 + cbase[L];  // :55, :80
    } else {
        const i64 A = owner[L];
        const int ak = (int)(A / g.P);
        const i64 r0 = A - (i64)ak * g.P;
        const int aj = (int)(r0 / g.nx), ai = (int)(r0 - (i64)aj * g.nx);
        const int l = lump_local(g, L, ai, aj, ak);
        c = 2 + cbase[A] + labels[arank[A] * (i64)(g.di * g.dj * g.dk) + l];
    }
    cnum[w] = c;
    used[c] = 1;
}
__global__ void lump_widen_kernel(i64 n, const uint8_t *__restrict__ in, i64 *__restrict__ out) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = in[q];
}
__global__ void lump_row_kernel(i64 N, const i64 *__restrict__ cnum, const i64 *__restrict__ newidx, i64 *__restrict__ crow,
                                unsigned *__restrict__ counts) {
    const i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= N) return;
    const i64 I = newidx[cnum[w]];  // 0-based coarse row
    crow[w] = I;
    atomicAdd(&counts[I], 1u);
}

// ---- 5. fill ----------------------------------------------------------------------------------------------------------
__global__ void lump_iota_kernel(i64 n, i64 *__restrict__ a, i64 first) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) a[q] = first + q;
}
__global__ void lump_colptr_kernel(i64 Nc, const i64 *__restrict__ offs, i64 N, i64 *__restrict__ colptr) {
    const i64 I = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (I < Nc) colptr[I] = offs[I] + 1;
    if (I == Nc) colptr[Nc] = N + 1;
}
__global__ void lump_counts_widen_kernel(i64 n, const unsigned *__restrict__ in, i64 *__restrict__ out) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = in[q];
}
// vol_c = LUMP * vol (:96): mul! walks the columns (wet cells) in ascending order and does y[row] += 1 * vol[col]
__global__ void lump_volc_kernel(i64 Nc, const i64 *__restrict__ spray_colptr, const i64 *__restrict__ spray_row,
                                 const double *__restrict__ vol, double *__restrict__ vol_c) {
    const i64 I = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= Nc) return;
    double acc = 0.0;
    for (i64 q = spray_colptr[I]; q < spray_colptr[I + 1]; ++q) acc = acc + 1 * vol[spray_row[q - 1] - 1];
    vol_c[I] = acc;
}
// LUMP = sparse(Diagonal(1 ./ vol_c)) * LUMP * sparse(Diagonal(vol)) (:97), left to right; SPRAY.nzval .= 1 (:102)
__global__ void lump_values_kernel(i64 N, const i64 *__restrict__ crow, const double *__restrict__ vol, const double *__restrict__ vol_c,
                                   i64 *__restrict__ lump_row, double *__restrict__ lump_val, double *__restrict__ spray_val) {
    const i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= N) return;
    const i64 I = crow[w];
    lump_row[w] = I + 1;
    lump_val[w] = ((1.0 / vol_c[I]) * 1) * vol[w];
    spray_val[w] = 1.0;
}
__global__ void lump_plus1_kernel(i64 n, const i64 *__restrict__ in, i64 *__restrict__ out) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = in[q] + 1;
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static int32_t lump_scan(otmb_ctx *ctx, const i64 *in, i64 *out, i64 n) {  // exclusive sum; out[n] is NOT written
    size_t tmp = 0;
    if (rocprim::exclusive_scan(nullptr, tmp, in, out, (i64)0, (size_t)n, rocprim::plus<i64>(), ctx->stream) != hipSuccess)
        return otmb_fail(ctx, OTMB_ERR_HIP, "exclusive_scan (size)");
    int32_t rc;
    if ((rc = otmb_reserve(ctx, ctx->lump[9], tmp + 16))) return rc;
    if (rocprim::exclusive_scan(ctx->lump[9].p, tmp, in, out, (i64)0, (size_t)n, rocprim::plus<i64>(), ctx->stream) != hipSuccess)
        return otmb_fail(ctx, OTMB_ERR_HIP, "exclusive_scan");
    return OTMB_OK;
}
#define GRID(n) dim3((unsigned)(((n) + 255) / 256 > 0 ? ((n) + 255) / 256 : 1)), dim3(256), 0, ctx->stream

extern "C" {

int32_t otmb_lump_and_spray_plan_dev(otmb_ctx *ctx, const uint8_t *wet3d, const uint8_t *mask, const int64_t *lwet3d,
                                     const int64_t *lwet, int64_t n_wet, int64_t nx, int64_t ny, int64_t nz,
                                     const int64_t *t_colptr, const int64_t *t_rowval, int64_t di, int64_t dj, int64_t dk,
                                     int64_t *n_coarse) {
    if (!ctx || !wet3d || !lwet3d || !t_colptr || !n_coarse || (n_wet > 0 && (!lwet || !t_rowval)))
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    ctx->lump_valid = false;
    if (nx < 1 || ny < 1 || nz < 1 || di < 1 || dj < 1 || dk < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "sizes");
    const i64 G = nx * ny * nz, N = n_wet, bv = di * dj * dk;
    if (G >= (1ll << 32) || nx > 4900 || n_wet < 0 || n_wet > G) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid too large (nx <= 4900)");
    if (bv > LS_MAX_BLOCK || di > nx + 4096 || dj > 4096 || dk > 4096) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "di*dj*dk > 4096");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    // scratch: [0] isanchor u8 G | [1] owner u32 G | [2] ncomp u32 G | [3] i64 G+2 (anchor rank, then reused) |
    // [4] i64 G (increments) | [5] cbase i64 G | [6] labels | [7] cnum i64 N, crow i64 N | [8] used u8 / newidx | [9] scan temp
    if ((rc = otmb_reserve(ctx, ctx->lump[0], (size_t)G + 16))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->lump[1], (size_t)G * 4 + 16))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->lump[2], (size_t)G * 4 + 16))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->lump[3], (size_t)(G + 2) * 8))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->lump[4], (size_t)(G + 2) * 8))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->lump[5], (size_t)(G + 2) * 8))) return rc;
    uint8_t *isanchor = (uint8_t *)ctx->lump[0].p;
    uint32_t *owner = (uint32_t *)ctx->lump[1].p, *ncomp = (uint32_t *)ctx->lump[2].p;
    i64 *arank = (i64 *)ctx->lump[3].p, *inc = (i64 *)ctx->lump[4].p, *cbase = (i64 *)ctx->lump[5].p;
    int *dflags = (int *)ctx->flags.p;
    HIP_TRY(ctx, hipMemsetAsync(isanchor, 0, (size_t)G, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dflags, 0, sizeof(int), ctx->stream));
    // 1. sweep
    int *covk = nullptr;
    if (dk > 1) {
        if ((rc = otmb_reserve(ctx, ctx->lump[10], (size_t)nx * ny * 4 + 16))) return rc;
        covk = (int *)ctx->lump[10].p;
        HIP_TRY(ctx, hipMemsetAsync(covk, 0xFF, (size_t)nx * ny * 4, ctx->stream));
    }
    {
        const int per = (dk > 1) ? (int)nz : 1;
        const unsigned groups = (unsigned)((nz + per - 1) / per);
        const size_t lds = (size_t)nx * (4 + 4 * 2 + 1) + 16;
        hipLaunchKernelGGL(lump_sweep_kernel, dim3(groups), dim3(LS_THREADS), lds, ctx->stream, mask, (int)nx, (int)ny, (int)nz, (int)di,
                           (int)dj, (int)dk, per, isanchor, owner, covk);
        HIP_TRY(ctx, hipGetLastError());
    }
    // anchor ranks (exclusive count of anchors before a cell) and their number
    hipLaunchKernelGGL(lump_anchor_flag_kernel, GRID(G), G, (const uint8_t *)isanchor, inc);
    if ((rc = lump_scan(ctx, inc, arank, G))) return rc;
    i64 last[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(&last[0], arank + (G - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&last[1], inc + (G - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const i64 nanchor = last[0] + last[1];
    if ((size_t)nanchor * (size_t)bv * 2 > ((size_t)64 << 30)) return otmb_fail(ctx, OTMB_ERR_ALLOC, "component scratch over 64 GB");
    if ((rc = otmb_reserve(ctx, ctx->lump[6], (size_t)nanchor * bv * 2 + 16))) return rc;
    unsigned short *labels = (unsigned short *)ctx->lump[6].p;
    // 2. components
    LumpGrid g{(int)nx, (int)ny, (int)nz, (int)di, (int)dj, (int)dk, nx * ny};
    hipLaunchKernelGGL(lump_components_kernel, GRID(G), g, G, (const uint8_t *)isanchor, (const i64 *)arank, wet3d, (const i64 *)lwet3d,
                       (const i64 *)lwet, (const i64 *)t_colptr, (const i64 *)t_rowval, labels, ncomp, dflags);
    // 3. the counter c at every cell
    hipLaunchKernelGGL(lump_increment_kernel, GRID(G), G, mask, (const uint8_t *)isanchor, (const uint32_t *)ncomp, inc);
    if ((rc = lump_scan(ctx, inc, cbase, G))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(&last[0], cbase + (G - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&last[1], inc + (G - 1), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_flags[0]) return otmb_fail(ctx, OTMB_ERR_ASYMMETRIC_PATTERN);
    const i64 cmax = 2 + last[0] + last[1];  // numbers in use are 2 .. cmax-1
    // 4. numbers of the wet cells, surviving rows
    if ((rc = otmb_reserve(ctx, ctx->lump[7], (size_t)(2 * N + 2) * 8))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->lump[8], (size_t)(cmax + 2) * 9 + 16))) return rc;
    i64 *cnum = (i64 *)ctx->lump[7].p, *crow = cnum + N;
    i64 *newidx = (i64 *)ctx->lump[8].p;
    uint8_t *used = (uint8_t *)(newidx + cmax + 2);
    HIP_TRY(ctx, hipMemsetAsync(used, 0, (size_t)cmax + 1, ctx->stream));
    hipLaunchKernelGGL(lump_number_kernel, GRID(N), g, N, (const i64 *)lwet, mask, (const uint32_t *)owner, (const i64 *)arank,
                       (const i64 *)cbase, (const unsigned short *)labels, cnum, used);
    // arank / inc are free again: widen the flags and scan them
    if ((rc = otmb_reserve(ctx, ctx->lump[4], (size_t)(cmax + 2) * 8))) return rc;
    inc = (i64 *)ctx->lump[4].p;
    hipLaunchKernelGGL(lump_widen_kernel, GRID(cmax + 1), cmax + 1, (const uint8_t *)used, inc);
    if ((rc = lump_scan(ctx, inc, newidx, cmax + 1))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(&last[0], newidx + cmax, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(&last[1], inc + cmax, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const i64 Nc = last[0] + last[1];
    // coarse row of every wet cell and the size of every coarse cell
    if ((rc = otmb_reserve(ctx, ctx->lump[2], (size_t)(Nc + 2) * 4 + (size_t)G * 4 + 16))) return rc;
    unsigned *counts = (unsigned *)ctx->lump[2].p;
    HIP_TRY(ctx, hipMemsetAsync(counts, 0, (size_t)(Nc + 1) * 4, ctx->stream));
    hipLaunchKernelGGL(lump_row_kernel, GRID(N), N, (const i64 *)cnum, (const i64 *)newidx, crow, counts);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->lump_valid = true;
    ctx->lump_N = N;
    ctx->lump_Nc = Nc;
    *n_coarse = Nc;
    return OTMB_OK;
}

int32_t otmb_lump_and_spray_fill_dev(otmb_ctx *ctx, const double *vol, int64_t *lump_colptr, int64_t *lump_rowval, double *lump_nzval,
                                     int64_t *spray_colptr, int64_t *spray_rowval, double *spray_nzval, double *vol_c) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    if (!ctx->lump_valid) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    const i64 N = ctx->lump_N, Nc = ctx->lump_Nc;
    if (!lump_colptr || !spray_colptr || (N > 0 && (!vol || !lump_rowval || !lump_nzval || !spray_rowval || !spray_nzval || !vol_c)))
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    i64 *cnum = (i64 *)ctx->lump[7].p, *crow = cnum + N;
    unsigned *counts = (unsigned *)ctx->lump[2].p;
    // SPRAY's column pointers: exclusive scan of the coarse cell sizes
    if ((rc = otmb_reserve(ctx, ctx->lump[4], (size_t)(Nc + 2) * 16))) return rc;
    i64 *wide = (i64 *)ctx->lump[4].p, *offs = wide + (Nc + 2);
    hipLaunchKernelGGL(lump_counts_widen_kernel, GRID(Nc + 1), Nc + 1, (const unsigned *)counts, wide);
    if ((rc = lump_scan(ctx, wide, offs, Nc + 1))) return rc;
    hipLaunchKernelGGL(lump_colptr_kernel, GRID(Nc + 1), Nc, (const i64 *)offs, N, (i64 *)spray_colptr);
    hipLaunchKernelGGL(lump_iota_kernel, GRID(N + 1), N + 1, (i64 *)lump_colptr, (i64)1);  // one entry per column
    if (N > 0) {
        // SPRAY = LUMP' (:101): the wet cells sorted by coarse row; the sort is stable, so members stay ascending
        if ((rc = otmb_reserve(ctx, ctx->lump[5], (size_t)N * 8 * 3 + 16))) return rc;
        i64 *keys_out = (i64 *)ctx->lump[5].p, *vals_in = keys_out + N, *vals_out = vals_in + N;
        hipLaunchKernelGGL(lump_iota_kernel, GRID(N), N, vals_in, (i64)1);
        size_t tmp = 0;
        int bits = 1;
        while (bits < 63 && (1ll << bits) <= Nc) ++bits;
        if (rocprim::radix_sort_pairs(nullptr, tmp, (const i64 *)crow, keys_out, (const i64 *)vals_in, vals_out, (size_t)N, 0, bits, ctx->stream) != hipSuccess)
            return otmb_fail(ctx, OTMB_ERR_HIP, "radix_sort_pairs (size)");
        if ((rc = otmb_reserve(ctx, ctx->lump[9], tmp + 16))) return rc;
        if (rocprim::radix_sort_pairs(ctx->lump[9].p, tmp, (const i64 *)crow, keys_out, (const i64 *)vals_in, vals_out, (size_t)N, 0, bits, ctx->stream) != hipSuccess)
            return otmb_fail(ctx, OTMB_ERR_HIP, "radix_sort_pairs");
        HIP_TRY(ctx, hipMemcpyAsync(spray_rowval, vals_out, (size_t)N * 8, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(lump_volc_kernel, GRID(Nc), Nc, (const i64 *)spray_colptr, (const i64 *)spray_rowval, vol, vol_c);
        hipLaunchKernelGGL(lump_values_kernel, GRID(N), N, (const i64 *)crow, vol, (const double *)vol_c, (i64 *)lump_rowval, lump_nzval,
                           spray_nzval);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OTMB_OK;
}

// host pointers: stage everything, rebuild the indices from the wet mask on the device, plan + fill, copy back
__global__ void lump_wet_to_v3d_kernel(i64 G, const uint8_t *__restrict__ wet, double *__restrict__ v) {
    const i64 L = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (L < G) v[L] = wet[L] ? 1.0 : __builtin_nan("");
}

int32_t otmb_lump_and_spray(otmb_ctx *ctx, const uint8_t *wet3d, const uint8_t *mask, int64_t nx, int64_t ny, int64_t nz,
                            const double *vol, int64_t n_wet, const int64_t *t_colptr, const int64_t *t_rowval, int64_t di,
                            int64_t dj, int64_t dk, int64_t *lump_rowval, double *lump_nzval, int64_t *spray_colptr,
                            int64_t *spray_rowval, double *vol_c, int64_t *n_coarse) {
    if (!ctx || !wet3d || !t_colptr || !spray_colptr || !n_coarse || (n_wet > 0 && (!vol || !t_rowval || !lump_rowval || !lump_nzval || !spray_rowval || !vol_c)))
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1 || n_wet < 0) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "sizes");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t G = (size_t)(nx * ny * nz), N = (size_t)n_wet;
    const i64 tnnz = t_colptr[n_wet] - 1;
    if (tnnz < 0) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "T colptr");
    // one allocation: wet, mask, v3d, lwet3d, lwet, wet', vol, Tp, Ti, outputs
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_wet = take(G), o_mask = take(G), o_v = take(G * 8), o_lw3 = take(G * 8), o_lw = take(G * 8), o_wet2 = take(G),
                 o_vol = take(N * 8 + 8), o_tp = take((N + 1) * 8), o_ti = take((size_t)tnnz * 8 + 8), o_lcp = take((N + 1) * 8),
                 o_lrv = take(N * 8 + 8), o_lnz = take(N * 8 + 8), o_scp = take((N + 2) * 8), o_srv = take(N * 8 + 8),
                 o_snz = take(N * 8 + 8), o_vc = take(N * 8 + 8);
    int32_t rc;
    DevBuf &pool = ctx->lump_host;
    if ((rc = otmb_reserve(ctx, pool, off + 256))) return rc;
    char *b = (char *)pool.p;
    HIP_TRY(ctx, hipMemcpyAsync(b + o_wet, wet3d, G, hipMemcpyHostToDevice, ctx->stream));
    if (mask) HIP_TRY(ctx, hipMemcpyAsync(b + o_mask, mask, G, hipMemcpyHostToDevice, ctx->stream));
    if (N) HIP_TRY(ctx, hipMemcpyAsync(b + o_vol, vol, N * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(b + o_tp, t_colptr, (N + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (tnnz) HIP_TRY(ctx, hipMemcpyAsync(b + o_ti, t_rowval, (size_t)tnnz * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(lump_wet_to_v3d_kernel, GRID((i64)G), (i64)G, (const uint8_t *)(b + o_wet), (double *)(b + o_v));
    int64_t n_found = 0;
    if ((rc = otmb_makeindices_dev(ctx, (const double *)(b + o_v), nx, ny, nz, (int64_t *)(b + o_lw3), (int64_t *)(b + o_lw),
                                   (uint8_t *)(b + o_wet2), &n_found)))
        return rc;
    if (n_found != n_wet) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "length(vol) != sum(wet3D)");
    int64_t Nc = 0;
    if ((rc = otmb_lump_and_spray_plan_dev(ctx, (const uint8_t *)(b + o_wet), mask ? (const uint8_t *)(b + o_mask) : nullptr,
                                           (const int64_t *)(b + o_lw3), (const int64_t *)(b + o_lw), n_wet, nx, ny, nz,
                                           (const int64_t *)(b + o_tp), (const int64_t *)(b + o_ti), di, dj, dk, &Nc)))
        return rc;
    if ((rc = otmb_lump_and_spray_fill_dev(ctx, (const double *)(b + o_vol), (int64_t *)(b + o_lcp), (int64_t *)(b + o_lrv),
                                           (double *)(b + o_lnz), (int64_t *)(b + o_scp), (int64_t *)(b + o_srv), (double *)(b + o_snz),
                                           (double *)(b + o_vc))))
        return rc;
    if (N) {
        HIP_TRY(ctx, hipMemcpyAsync(lump_rowval, b + o_lrv, N * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(lump_nzval, b + o_lnz, N * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(spray_rowval, b + o_srv, N * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (Nc) HIP_TRY(ctx, hipMemcpyAsync(vol_c, b + o_vc, (size_t)Nc * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(spray_colptr, b + o_scp, (size_t)(Nc + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_coarse = Nc;
    return OTMB_OK;
}

}  // extern "C"
