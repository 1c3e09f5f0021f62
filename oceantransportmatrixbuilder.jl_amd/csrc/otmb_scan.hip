// otmb_scan.hip -- exclusive scan of per-tile sums (supporting op of the two-phase CSC assembly and of
// makeindices).  Two levels: (1) one workgroup per 1024 tiles scans its tiles and emits its total;
// (2) one workgroup scans the totals in place and adds them back.  nf <= 8 interleaved fields.
#include "otmb_common.h"

#define SCAN_THREADS 1024
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

// offs: [ntiles][nf] exclusive prefix; tot: [nf]; gsum: scratch for (ntiles/1024 + 1) * nf values
void otmb_launch_tilescan(hipStream_t s, const uint32_t *sums, i64 *offs, i64 *tot, i64 ntiles, int nf, i64 *gsum) {
    const i64 ngroups = (ntiles + SCAN_THREADS - 1) / SCAN_THREADS;
    hipLaunchKernelGGL(tilescan_groups, dim3((unsigned)ngroups), dim3(SCAN_THREADS), 0, s, sums, offs, gsum, ntiles, nf);
    hipLaunchKernelGGL(tilescan_top, dim3(1), dim3(SCAN_THREADS), 0, s, gsum, tot, ngroups, nf);
    if (ngroups > 1) {
        const i64 n = ntiles * nf;
        hipLaunchKernelGGL(tilescan_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, offs, (const i64 *)gsum, ntiles, nf);
    }
}
