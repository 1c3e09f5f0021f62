// otmb_scan.hip -- exclusive scan of per-tile sums (supporting op of the two-phase CSC assembly and of
// makeindices).  Two levels: (1) one workgroup per 1024 tiles scans its tiles and emits its total;
// (2) one workgroup scans the totals in place and adds them back.  nf <= 8 interleaved fields.
#include "otmb_common.h"

#include <cstdlib>

#define SCAN_THREADS OTMB_SCAN_GROUP
#define SCAN_MAXF 8

// block-wide inclusive scan of one i64 per thread; returns the block total through `total`
__device__ __forceinline__ i64 block_incl_scan(i64 x, i64 (*wave_tot)[SCAN_THREADS / 64], int f, i64 &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        i64 y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wave_tot[f][wid] = x;
    __syncthreads();
    i64 before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        const i64 v = wave_tot[f][w];
        if (w < wid) before += v;
        all += v;
    }
    total = all;
    return before + x;
}

// level 1: offs[t][f] = exclusive prefix inside the group of 1024 tiles; gsum[g][f] = group total
__global__ __launch_bounds__(SCAN_THREADS) void tilescan_groups(const uint32_t *__restrict__ sums, i64 *__restrict__ offs,
                                                                 i64 *__restrict__ gsum, i64 ntiles, int nf) {
    __shared__ i64 wave_tot[SCAN_MAXF][SCAN_THREADS / 64];
    const i64 t = (i64)blockIdx.x * SCAN_THREADS + threadIdx.x;
#pragma unroll
    for (int f = 0; f < SCAN_MAXF; ++f) {
        if (f < nf) {  // nf is uniform
            const i64 mine = (t < ntiles) ? (i64)sums[t * nf + f] : 0;
            i64 total;
            const i64 incl = block_incl_scan(mine, wave_tot, f, total);
            if (t < ntiles) offs[t * nf + f] = incl - mine;
            if (threadIdx.x == 0) gsum[(i64)blockIdx.x * nf + f] = total;
        }
    }
}

// level 1 for the five interleaved fields of the transport-matrix tiles: the tile sums are below 2^11 (a tile has 256
// columns of at most 7 rows), so a group's prefixes fit 32 bits -- 32-bit shuffles, all five fields behind ONE barrier.
__global__ __launch_bounds__(SCAN_THREADS) void tilescan_groups5(const uint32_t *__restrict__ sums, i64 *__restrict__ offs,
                                                                  i64 *__restrict__ gsum, i64 ntiles) {
    __shared__ uint32_t wave_tot[5][SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const i64 t = (i64)blockIdx.x * SCAN_THREADS + threadIdx.x;
    uint32_t mine[5], x[5];
#pragma unroll
    for (int f = 0; f < 5; ++f) mine[f] = (t < ntiles) ? sums[t * 5 + f] : 0u;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        x[f] = mine[f];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x[f], d);
            if (lane >= d) x[f] += y;
        }
        if (lane == 63) wave_tot[f][wid] = x[f];
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; ++w) {
            const uint32_t v = wave_tot[f][w];
            if (w < wid) before += v;
            all += v;
        }
        if (t < ntiles) offs[t * 5 + f] = (i64)(before + x[f] - mine[f]);
        if (threadIdx.x == 0) gsum[(i64)blockIdx.x * 5 + f] = (i64)all;
    }
}

// The same level-1 scan for tile counts that facefluxes accumulated as ONE packed 64-bit word per tile (otmb_facefluxes_counts_dev:
// T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10 bits, bit 63 = a flux into land was seen, bit 62 = wrong bases table; the fields that
// depend on the wet mask alone come from the grid's static table): every thread
// takes its tile's word, writes ZERO back (the buffer is clean again for the facefluxes call after next: same thread, same address, so
// the store is ordered behind the load), hands the five counts to the fill pass in the layout the counting pass writes (sums[t][5]) and
// scans them as above.
__global__ __launch_bounds__(SCAN_THREADS) void tilescan_groups5_packed(unsigned long long *__restrict__ packed,
                                                                         const unsigned long long *__restrict__ stat, uint32_t *__restrict__ sums,
                                                                         i64 *__restrict__ offs, i64 *__restrict__ gsum, i64 ntiles,
                                                                         int *__restrict__ flags, unsigned long long keep) {
    __shared__ uint32_t wave_tot[5][SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const i64 t = (i64)blockIdx.x * SCAN_THREADS + threadIdx.x;
    unsigned long long w = 0;
    if (t < ntiles) {
        w = packed[t] + stat[t];  // per time slice (Tadv, TκVML, flag bits) + per grid (T's union, TκH, TκVdeep): disjoint fields
        packed[t] = 0ull;
    }
    if (w & FFC_BAD_FLUX) { if (flags[FLAG_FLUX_INTO_LAND] == 0) atomicExch(&flags[FLAG_FLUX_INTO_LAND], 1); }
    if (w & FFC_BAD_TABLE) { if (flags[FLAG_COUNT_MISMATCH] == 0) atomicExch(&flags[FLAG_COUNT_MISMATCH], 1); }
    w &= keep;  // (matrices that are not materialised -- T alone, given operators -- count nothing: TmParams.skip)
    uint32_t mine[5], x[5];
    mine[0] = (uint32_t)(w & 0x7ff);
    mine[1] = (uint32_t)((w >> 11) & 0x7ff);
    mine[2] = (uint32_t)((w >> 22) & 0x7ff);
    mine[3] = (uint32_t)((w >> 33) & 0x3ff);
    mine[4] = (uint32_t)((w >> 43) & 0x3ff);
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        if (t < ntiles) sums[t * 5 + f] = mine[f];
        x[f] = mine[f];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x[f], d);
            if (lane >= d) x[f] += y;
        }
        if (lane == 63) wave_tot[f][wid] = x[f];
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int q = 0; q < SCAN_THREADS / 64; ++q) {
            const uint32_t v = wave_tot[f][q];
            if (q < wid) before += v;
            all += v;
        }
        if (t < ntiles) offs[t * 5 + f] = (i64)(before + x[f] - mine[f]);
        if (threadIdx.x == 0) gsum[(i64)blockIdx.x * 5 + f] = (i64)all;
    }
}

// level 2: exclusive scan of the group totals (ngroups <= 1024 per pass, looped), totals to tot[f]
__global__ __launch_bounds__(SCAN_THREADS) void tilescan_top(i64 *__restrict__ gsum, i64 *__restrict__ tot, i64 ngroups, int nf) {
    __shared__ i64 wave_tot[SCAN_MAXF][SCAN_THREADS / 64];
#pragma unroll
    for (int f = 0; f < SCAN_MAXF; ++f) {
        if (f < nf) {
            i64 carry = 0;
            for (i64 base = 0; base < ngroups; base += SCAN_THREADS) {
                const i64 g = base + threadIdx.x;
                const i64 mine = (g < ngroups) ? gsum[g * nf + f] : 0;
                i64 total;
                const i64 incl = block_incl_scan(mine, wave_tot, f, total);
                if (g < ngroups) gsum[g * nf + f] = carry + incl - mine;
                carry += total;
                __syncthreads();
            }
            if (threadIdx.x == 0) tot[f] = carry;
        }
    }
}

// level 3: add the group bases (consumers that can do the add themselves skip this)
__global__ __launch_bounds__(256) void tilescan_add(i64 *__restrict__ offs, const i64 *__restrict__ gsum, i64 ntiles, int nf) {
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e < ntiles * nf) {
        const i64 t = e / nf;
        const int f = (int)(e - t * nf);
        offs[e] += gsum[(t / SCAN_THREADS) * nf + f];
    }
}

// Small inputs (up to SCAN_SINGLE_MAX tiles per thread): one workgroup does the whole scan in one launch -- each thread
// sums a contiguous chunk of tiles, the chunk sums are scanned across the workgroup (all fields behind one barrier),
// then each thread writes the exclusive prefixes of its chunk.  Measured at 11920 tiles x 5 fields (12 tiles per
// thread, strided loads): 71 us against 19.5 us for the three launches below, so it only serves inputs of one group.
#define SCAN_SINGLE_MAX 1
template <int NF>
__global__ __launch_bounds__(SCAN_THREADS) void tilescan_single(const uint32_t *__restrict__ sums, i64 *__restrict__ offs,
                                                                 i64 *__restrict__ tot, i64 ntiles, int per) {
    __shared__ i64 wave_tot[NF][SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const i64 t0 = (i64)threadIdx.x * per;
    i64 loc[NF], inc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) loc[f] = 0;
    for (int q = 0; q < per; ++q) {
        const i64 t = t0 + q;
        if (t < ntiles) {
#pragma unroll
            for (int f = 0; f < NF; ++f) loc[f] += (i64)sums[t * NF + f];
        }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        i64 x = loc[f];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            i64 y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        inc[f] = x;
        if (lane == 63) wave_tot[f][wid] = x;
    }
    __syncthreads();
    i64 run[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        i64 before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < SCAN_THREADS / 64; ++w) {
            const i64 v = wave_tot[f][w];
            if (w < wid) before += v;
            all += v;
        }
        run[f] = before + inc[f] - loc[f];
        if (threadIdx.x == 0) tot[f] = all;
    }
    for (int q = 0; q < per; ++q) {
        const i64 t = t0 + q;
        if (t < ntiles) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                offs[t * NF + f] = run[f];
                run[f] += (i64)sums[t * NF + f];
            }
        }
    }
}

// offs: [ntiles][nf] exclusive prefix; tot: [nf]; gsum: scratch for (ntiles/1024 + 1) * nf values
void otmb_launch_tilescan(hipStream_t s, const uint32_t *sums, i64 *offs, i64 *tot, i64 ntiles, int nf, i64 *gsum) {
    const i64 ngroups = (ntiles + SCAN_THREADS - 1) / SCAN_THREADS;
    const char *env = getenv("OTMB_SCAN_SINGLE_MAX");  // tests force the three-launch path on small inputs with 0
    const i64 single_max = env ? atoll(env) : SCAN_SINGLE_MAX;
    if (ngroups <= single_max && ngroups <= SCAN_SINGLE_MAX && (nf == 5 || nf == 1)) {
        if (nf == 5)
            hipLaunchKernelGGL(tilescan_single<5>, dim3(1), dim3(SCAN_THREADS), 0, s, sums, offs, tot, ntiles, (int)ngroups);
        else
            hipLaunchKernelGGL(tilescan_single<1>, dim3(1), dim3(SCAN_THREADS), 0, s, sums, offs, tot, ntiles, (int)ngroups);
        return;
    }
    hipLaunchKernelGGL(tilescan_groups, dim3((unsigned)ngroups), dim3(SCAN_THREADS), 0, s, sums, offs, gsum, ntiles, nf);
    hipLaunchKernelGGL(tilescan_top, dim3(1), dim3(SCAN_THREADS), 0, s, gsum, tot, ngroups, nf);
    if (ngroups > 1) {
        const i64 n = ntiles * nf;
        hipLaunchKernelGGL(tilescan_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, offs, (const i64 *)gsum, ntiles, nf);
    }
}

// Level 1 only: offs holds prefixes inside groups of OTMB_SCAN_GROUP tiles and gsum the group totals; the consumer
// adds the totals of the preceding groups itself (the fill pass of transportmatrix does, for up to a few dozen groups:
// two launches fewer on the critical path).
void otmb_launch_tilescan_groups(hipStream_t s, const uint32_t *sums, i64 *offs, i64 *gsum, i64 ntiles, int nf) {
    const i64 ngroups = (ntiles + SCAN_THREADS - 1) / SCAN_THREADS;
    if (nf == 5)  // (callers pass sums below 2^11 per tile: see tilescan_groups5)
        hipLaunchKernelGGL(tilescan_groups5, dim3((unsigned)ngroups), dim3(SCAN_THREADS), 0, s, sums, offs, gsum, ntiles);
    else
        hipLaunchKernelGGL(tilescan_groups, dim3((unsigned)ngroups), dim3(SCAN_THREADS), 0, s, sums, offs, gsum, ntiles, nf);
}

// Tile counts from facefluxes (one packed word per tile): level 1 always; all_levels adds the group bases to offs and leaves the
// totals in tot, as otmb_launch_tilescan does (the two-phase plan, and grids of more groups than the fill pass adds up itself).
void otmb_launch_tilescan_packed(hipStream_t s, unsigned long long *packed, const unsigned long long *stat, uint32_t *sums, i64 *offs, i64 *tot, i64 *gsum, i64 ntiles,
                                 int *flags, unsigned long long keep, bool all_levels) {
    const i64 ngroups = (ntiles + SCAN_THREADS - 1) / SCAN_THREADS;
    hipLaunchKernelGGL(tilescan_groups5_packed, dim3((unsigned)ngroups), dim3(SCAN_THREADS), 0, s, packed, stat, sums, offs, gsum, ntiles, flags, keep);
    if (!all_levels) return;
    hipLaunchKernelGGL(tilescan_top, dim3(1), dim3(SCAN_THREADS), 0, s, gsum, tot, ngroups, 5);
    if (ngroups > 1) {
        const i64 n = ntiles * 5;
        hipLaunchKernelGGL(tilescan_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, offs, (const i64 *)gsum, ntiles, 5);
    }
}
