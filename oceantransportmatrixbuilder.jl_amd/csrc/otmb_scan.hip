// otmb_scan.hip -- exclusive scan of per-tile sums (supporting op of the two-pass CSC assembly
// and of makeindices).  One 1024-thread workgroup walks the tiles 1024 at a time (coalesced),
// wave-scans each field with DPP/LDS-free shuffles and carries the running totals.
#include "otmb_common.h"

#define SCAN_THREADS 1024
#define SCAN_MAXF 8

__global__ __launch_bounds__(SCAN_THREADS) void tilescan_kernel(const uint32_t *__restrict__ sums,
                                                                 i64 *__restrict__ offs,
                                                                 i64 *__restrict__ tot, i64 ntiles, int nf) {
    __shared__ i64 wave_tot[SCAN_MAXF][SCAN_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    i64 carry[SCAN_MAXF];
#pragma unroll
    for (int f = 0; f < SCAN_MAXF; ++f) carry[f] = 0;
    for (i64 base = 0; base < ntiles; base += SCAN_THREADS) {
        const i64 t = base + tid;
        i64 incl[SCAN_MAXF];
#pragma unroll
        for (int f = 0; f < SCAN_MAXF; ++f) {
            i64 x = (f < nf && t < ntiles) ? (i64)sums[t * nf + f] : 0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                i64 y = __shfl_up(x, d);
                if (lane >= d) x += y;
            }
            incl[f] = x;
            if (lane == 63) wave_tot[f][wid] = x;
        }
        __syncthreads();
#pragma unroll
        for (int f = 0; f < SCAN_MAXF; ++f) {
            if (f < nf) {
                i64 before = 0, all = 0;
                for (int w = 0; w < SCAN_THREADS / 64; ++w) {
                    i64 v = wave_tot[f][w];
                    if (w < wid) before += v;
                    all += v;
                }
                if (t < ntiles) offs[t * nf + f] = carry[f] + before + incl[f] - (i64)sums[t * nf + f];
                carry[f] += all;
            }
        }
        __syncthreads();
    }
    if (tid < nf) {
        // carry is identical in every thread
        i64 c = 0;
#pragma unroll
        for (int f = 0; f < SCAN_MAXF; ++f)
            if (f == tid) c = carry[f];
        tot[tid] = c;
    }
}

void otmb_launch_tilescan(hipStream_t s, const uint32_t *sums, i64 *offs, i64 *tot, i64 ntiles, int nf) {
    hipLaunchKernelGGL(tilescan_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, sums, offs, tot, ntiles, nf);
}
