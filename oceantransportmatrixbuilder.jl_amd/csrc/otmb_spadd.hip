// otmb_spadd.hip -- C = A + B for SparseMatrixCSC (SparseArrays' map(+, A, B), the `+` of
// src/matrixbuilding.jl:147): per column a sorted merge of the two row lists; a row present in one operand
// only adds +0.0 to it; results that are exactly zero are NOT stored.  Needed when the caller passes
// precomputed operators to transportmatrix (:133-143), where T = ((Tadv + TκH) + TκVML) + TκVdeep is formed
// from arbitrary matrices.  One thread per column, the entries of a workgroup's columns staged in LDS; count ->
// tile scan -> fill; values are recomputed in the fill pass.  Indices are Julia's (1-based Int64).
#include "otmb_common.h"

#define SA_THREADS 256
#define SA_CAP 2040  // entries of one operand that a workgroup's 256 columns may hold for the staged path (4 arrays x 2040 x 8 B: just under 64 KB of static LDS; 7 rows x 256 columns = 1792)

// One merge step of column j: the next result (row, value) of the sorted union of A's and B's rows.  IDX / VAL: how entry k of an operand is read.
#define SA_MERGE_LOOP(AI, AX, BI, BX, BODY)                                               \
    while (ka < ae || kb < be) {                                                          \
        const i64 ra = (ka < ae) ? AI(ka) : INT64_MAX, rb = (kb < be) ? BI(kb) : INT64_MAX; \
        double x;                                                                         \
        i64 r;                                                                            \
        if (ra == rb) { x = AX(ka) + BX(kb); r = ra; ++ka; ++kb; }                        \
        else if (ra < rb) { x = AX(ka) + 0.0; r = ra; ++ka; }                             \
        else { x = 0.0 + BX(kb); r = rb; ++kb; }                                          \
        BODY                                                                              \
    }

// Round 4: the entries of a workgroup's 256 columns are one contiguous piece of each operand: they are brought into LDS by coalesced loads
// and every thread merges ITS column from there.  (Before, a thread walked its column through global memory: a chain of dependent 8-byte
// loads 30-60 bytes apart from its neighbours' -- 1.4 ms for Tadv + TκH on the 1 degree grid, 0.55 TB/s.)  A block of columns with more
// than SA_CAP entries in an operand (not a transport operator) takes the old path.
template <bool FILL>
__global__ __launch_bounds__(SA_THREADS) void spadd_kernel(i64 n, const i64 *__restrict__ Ap, const i64 *__restrict__ Ai,
                                                            const double *__restrict__ Ax, const i64 *__restrict__ Bp,
                                                            const i64 *__restrict__ Bi, const double *__restrict__ Bx,
                                                            uint32_t *__restrict__ tilesums, const i64 *__restrict__ tileoffs,
                                                            i64 *__restrict__ Cp, i64 *__restrict__ Ci, double *__restrict__ Cx) {
    __shared__ unsigned wave_tot[SA_THREADS / 64];
    __shared__ i64 s_ai[SA_CAP], s_bi[SA_CAP];
    __shared__ double s_ax[SA_CAP], s_bx[SA_CAP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const i64 j0 = (i64)blockIdx.x * SA_THREADS, j1 = (j0 + SA_THREADS < n) ? j0 + SA_THREADS : n;
    const i64 j = j0 + tid;
    // the block's piece of each operand (uniform)
    const i64 a0 = Ap[j0] - 1, a1 = Ap[j1] - 1, b0 = Bp[j0] - 1, b1 = Bp[j1] - 1;
    const bool staged = (a1 - a0) <= SA_CAP && (b1 - b0) <= SA_CAP && a1 >= a0 && b1 >= b0;
    if (staged) {
        for (i64 e = tid; e < a1 - a0; e += SA_THREADS) { s_ai[e] = Ai[a0 + e]; s_ax[e] = Ax[a0 + e]; }
        for (i64 e = tid; e < b1 - b0; e += SA_THREADS) { s_bi[e] = Bi[b0 + e]; s_bx[e] = Bx[b0 + e]; }
        __syncthreads();
    }
    unsigned cnt = 0;
    i64 a = 0, ae = 0, b = 0, be = 0;
    if (j < n) {
        a = Ap[j] - 1; ae = Ap[j + 1] - 1; b = Bp[j] - 1; be = Bp[j + 1] - 1;
        i64 ka = a, kb = b;
        if (staged) {
#define SA_SAI(k) s_ai[(k) - a0]
#define SA_SAX(k) s_ax[(k) - a0]
#define SA_SBI(k) s_bi[(k) - b0]
#define SA_SBX(k) s_bx[(k) - b0]
            SA_MERGE_LOOP(SA_SAI, SA_SAX, SA_SBI, SA_SBX, (void)r; cnt += (x != 0.0);)  // count the non-zero results
        } else {
#define SA_GAI(k) Ai[k]
#define SA_GAX(k) Ax[k]
#define SA_GBI(k) Bi[k]
#define SA_GBX(k) Bx[k]
            SA_MERGE_LOOP(SA_GAI, SA_GAX, SA_GBI, SA_GBX, (void)r; cnt += (x != 0.0);)
        }
    }
    unsigned incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < SA_THREADS / 64; ++w) {
        const unsigned v = wave_tot[w];
        if (w < wid) before += v;
        all += v;
    }
    if (!FILL) {
        if (tid == 0) tilesums[blockIdx.x] = all;
        return;
    }
    if (j < n) {
        i64 q = tileoffs[blockIdx.x] + before + incl - cnt;
        Cp[j] = q + 1;
        i64 ka = a, kb = b;
        if (staged) {
            SA_MERGE_LOOP(SA_SAI, SA_SAX, SA_SBI, SA_SBX, if (x != 0.0) { Ci[q] = r; Cx[q] = x; ++q; })
        } else {
            SA_MERGE_LOOP(SA_GAI, SA_GAX, SA_GBI, SA_GBX, if (x != 0.0) { Ci[q] = r; Cx[q] = x; ++q; })
        }
    }
}

__global__ void spadd_finish(i64 *Cp, i64 n, const i64 *tot) {
    if (threadIdx.x == 0) Cp[n] = tot[0] + 1;
}

extern "C" {

// Two-phase like transportmatrix: plan returns nnz(A+B); fill writes Cp (n+1), Ci, Cx (nnz).  Device pointers.
int32_t otmb_spadd_plan_dev(otmb_ctx *ctx, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                            const int64_t *Bp, const int64_t *Bi, const double *Bx, int64_t *nnz_out) {
    if (!ctx || !Ap || !Bp || !nnz_out || n < 0) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 nt = (n + SA_THREADS - 1) / SA_THREADS;
    int32_t rc;
    if ((rc = otmb_reserve(ctx, ctx->blocksums, (size_t)(nt + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->blockoffs, (size_t)(nt + 1) * sizeof(i64) + otmb_scan_scratch(nt, 1)))) return rc;
    i64 *dtot = (i64 *)((int *)ctx->flags.p + OTMB_NFLAGS) + 9;
    HIP_TRY(ctx, hipMemsetAsync(dtot, 0, sizeof(i64), ctx->stream));
    if (nt > 0) {
        hipLaunchKernelGGL(spadd_kernel<false>, dim3((unsigned)nt), dim3(SA_THREADS), 0, ctx->stream, (i64)n, (const i64 *)Ap,
                           (const i64 *)Ai, Ax, (const i64 *)Bp, (const i64 *)Bi, Bx, (uint32_t *)ctx->blocksums.p,
                           (const i64 *)nullptr, (i64 *)nullptr, (i64 *)nullptr, (double *)nullptr);
        otmb_launch_tilescan(ctx->stream, (const uint32_t *)ctx->blocksums.p, (i64 *)ctx->blockoffs.p, dtot, nt, 1,
                             (i64 *)ctx->blockoffs.p + nt + 1);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot + 9, dtot, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *nnz_out = ctx->h_tot[9];
    return OTMB_OK;
}

int32_t otmb_spadd_fill_dev(otmb_ctx *ctx, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                            const int64_t *Bp, const int64_t *Bi, const double *Bx, int64_t *Cp, int64_t *Ci, double *Cx) {
    if (!ctx || !Ap || !Bp || !Cp || n < 0) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 nt = (n + SA_THREADS - 1) / SA_THREADS;
    i64 *dtot = (i64 *)((int *)ctx->flags.p + OTMB_NFLAGS) + 9;
    if (nt > 0)
        hipLaunchKernelGGL(spadd_kernel<true>, dim3((unsigned)nt), dim3(SA_THREADS), 0, ctx->stream, (i64)n, (const i64 *)Ap,
                           (const i64 *)Ai, Ax, (const i64 *)Bp, (const i64 *)Bi, Bx, (uint32_t *)nullptr,
                           (const i64 *)ctx->blockoffs.p, (i64 *)Cp, (i64 *)Ci, Cx);
    hipLaunchKernelGGL(spadd_finish, dim3(1), dim3(64), 0, ctx->stream, (i64 *)Cp, (i64)n, (const i64 *)dtot);
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}

// Host-pointer convenience: C = A + B, caller provides Cp (n+1) and Ci/Cx with capacity nnz(A)+nnz(B).
int32_t otmb_spadd(otmb_ctx *ctx, int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, const int64_t *Bp,
                   const int64_t *Bi, const double *Bx, int64_t *Cp, int64_t *Ci, double *Cx, int64_t *nnz_out) {
    if (!ctx || !Ap || !Bp || !Cp || !nnz_out || n < 0) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 na = Ap[n] - 1, nb = Bp[n] - 1;
    void *d[9];
    const size_t sz[9] = {(size_t)(n + 1) * 8, (size_t)na * 8, (size_t)na * 8, (size_t)(n + 1) * 8, (size_t)nb * 8, (size_t)nb * 8,
                          (size_t)(n + 1) * 8, (size_t)(na + nb) * 8, (size_t)(na + nb) * 8};
    const void *src[6] = {Ap, Ai, Ax, Bp, Bi, Bx};
    for (int k = 0; k < 9; ++k) {
        if (hipMalloc(&d[k], sz[k] ? sz[k] : 8) != hipSuccess) {
            for (int q = 0; q < k; ++q) (void)hipFree(d[q]);
            return otmb_fail(ctx, OTMB_ERR_ALLOC, "hipMalloc");
        }
        if (k < 6 && sz[k]) (void)hipMemcpyAsync(d[k], src[k], sz[k], hipMemcpyHostToDevice, ctx->stream);
    }
    int32_t rc = otmb_spadd_plan_dev(ctx, n, (const int64_t *)d[0], (const int64_t *)d[1], (const double *)d[2], (const int64_t *)d[3],
                                     (const int64_t *)d[4], (const double *)d[5], nnz_out);
    if (!rc) rc = otmb_spadd_fill_dev(ctx, n, (const int64_t *)d[0], (const int64_t *)d[1], (const double *)d[2], (const int64_t *)d[3],
                                      (const int64_t *)d[4], (const double *)d[5], (int64_t *)d[6], (int64_t *)d[7], (double *)d[8]);
    if (!rc) {
        (void)hipMemcpyAsync(Cp, d[6], sz[6], hipMemcpyDeviceToHost, ctx->stream);
        if (*nnz_out > 0) {
            (void)hipMemcpyAsync(Ci, d[7], (size_t)*nnz_out * 8, hipMemcpyDeviceToHost, ctx->stream);
            (void)hipMemcpyAsync(Cx, d[8], (size_t)*nnz_out * 8, hipMemcpyDeviceToHost, ctx->stream);
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) rc = otmb_fail(ctx, OTMB_ERR_HIP, "synchronize");
    }
    for (int k = 0; k < 9; ++k) (void)hipFree(d[k]);
    return rc;
}
}
