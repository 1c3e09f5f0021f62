// otmb_indices.hip -- makeindices(v3D) on the device (src/matrixbuilding.jl:10-24).
// wet = !isnan(v3D); Lwet = ascending linear indices of wet cells (a stream compaction);
// Lwet3D = wet rank (1-based) or 0 for `missing`; wet3D as bytes.  Count -> tile scan -> write.
#include "otmb_common.h"

#define IX_THREADS 256
#define IX_PER 4
#define IX_TILE (IX_THREADS * IX_PER)

template <bool WRITE>
__global__ __launch_bounds__(IX_THREADS) void indices_kernel(const double *__restrict__ v, i64 G,
                                                              uint32_t *__restrict__ tilesums,
                                                              const i64 *__restrict__ tileoffs,
                                                              i64 *__restrict__ lwet3d, i64 *__restrict__ lwet,
                                                              uint8_t *__restrict__ wet3d) {
    __shared__ unsigned wave_tot[IX_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const i64 tile = blockIdx.x;
    i64 run = WRITE ? tileoffs[tile] : 0;
    for (int ch = 0; ch < IX_PER; ++ch) {
        const i64 L = tile * IX_TILE + (i64)ch * IX_THREADS + tid;
        const bool wet = (L < G) && !isnan(v[L]);  // :15
        const u64 b = __ballot(wet);
        const unsigned inwave = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wid] = __popcll(b);
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < IX_THREADS / 64; ++w) {
            const unsigned t = wave_tot[w];
            if (w < wid) before += t;
            all += t;
        }
        __syncthreads();
        if (WRITE && L < G) {
            const i64 rank = run + before + inwave;  // wet cells before L
            if (lwet3d) lwet3d[L] = wet ? rank + 1 : 0;  // :19-20
            if (wet3d) wet3d[L] = wet ? 1 : 0;           // :17-18
            if (wet && lwet) lwet[rank] = L + 1;          // :15
        }
        run += all;
    }
    if (!WRITE && tid == 0) tilesums[tile] = (uint32_t)run;
}

extern "C" int32_t otmb_makeindices_dev(otmb_ctx *ctx, const double *v3d, int64_t nx, int64_t ny, int64_t nz,
                                        int64_t *lwet3d, int64_t *lwet, uint8_t *wet3d, int64_t *n_wet) {
    if (!ctx || !v3d || !n_wet) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 G = nx * ny * nz;
    const i64 ntiles = (G + IX_TILE - 1) / IX_TILE;
    int32_t rc;
    if ((rc = otmb_reserve(ctx, ctx->blocksums, (size_t)ntiles * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->blockoffs, (size_t)(ntiles + 1) * sizeof(i64) + otmb_scan_scratch(ntiles, 1)))) return rc;
    uint32_t *sums = (uint32_t *)ctx->blocksums.p;
    i64 *offs = (i64 *)ctx->blockoffs.p;
    i64 *dtot = (i64 *)((int *)ctx->flags.p + OTMB_NFLAGS) + 13;  // a totals word of its own (0-7 belong to transportmatrix)
    {
    KernelTimer kt(ctx, K_IDX_COUNT);
    hipLaunchKernelGGL(indices_kernel<false>, dim3((unsigned)ntiles), dim3(IX_THREADS), 0, ctx->stream, v3d, G, sums,
                       (const i64 *)nullptr, (i64 *)nullptr, (i64 *)nullptr, (uint8_t *)nullptr);
    }
    {
        KernelTimer kt(ctx, K_TILESCAN);
        otmb_launch_tilescan(ctx->stream, sums, offs, dtot, ntiles, 1, offs + ntiles + 1);
    }
    if (lwet3d || lwet || wet3d) {
        KernelTimer kt(ctx, K_IDX_WRITE);
        hipLaunchKernelGGL(indices_kernel<true>, dim3((unsigned)ntiles), dim3(IX_THREADS), 0, ctx->stream, v3d, G,
                           (uint32_t *)nullptr, (const i64 *)offs, (i64 *)lwet3d, (i64 *)lwet, wet3d);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot + 13, dtot, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *n_wet = ctx->h_tot[13];
    return OTMB_OK;
}
