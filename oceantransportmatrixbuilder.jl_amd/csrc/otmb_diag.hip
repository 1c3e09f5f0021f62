// otmb_diag.hip -- what THIS box's memory system sustains for the two plainest jobs there are: one HBM read stream and one HBM write
// stream (non-temporal, like the matrices), 2 GiB each, all CUs.  The boxes of the pool differ by up to 15 % on the same kernels
// (profiles/r03, r04); bench.py puts these two rates beside its roofline record so that a kernel time can be read as a fraction of what
// the box it ran on can do.  Diagnostic only: nothing in the hot path calls it.
#include "otmb_common.h"

typedef double d2 __attribute__((ext_vector_type(2)));

// workgroup b reads / writes, in pass q, 8 consecutive pieces of 256 x 16 bytes: eight independent 16-byte accesses per lane in flight
__global__ __launch_bounds__(256) void diag_read_kernel(const d2 *__restrict__ buf, size_t n, int passes, double *sink) {
    double acc = 0;
    for (int q = 0; q < passes; ++q) {
        const size_t e0 = (((size_t)q * gridDim.x + blockIdx.x) * 8) * 256 + threadIdx.x;
        d2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(buf + ((e0 + (size_t)u * 256) & (n - 1)));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x;
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}
__global__ __launch_bounds__(256) void diag_write_kernel(d2 *__restrict__ buf, size_t n, int passes) {
    for (int q = 0; q < passes; ++q) {
        const size_t e0 = (((size_t)q * gridDim.x + blockIdx.x) * 8) * 256 + threadIdx.x;
        const d2 x = {(double)q, (double)e0};
#pragma unroll
        for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(x, buf + ((e0 + (size_t)u * 256) & (n - 1)));
    }
}

extern "C" int32_t otmb_ctx_box_ceilings(otmb_ctx *ctx, double *read_gbs, double *write_gbs) {
    if (!ctx || !read_gbs || !write_gbs) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int ncu = 0;
    HIP_TRY(ctx, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device));
    if (ncu <= 0) ncu = 256;
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(d2);
    char *buf = nullptr;
    if (hipMalloc(&buf, bytes + 64) != hipSuccess) return otmb_fail(ctx, OTMB_ERR_ALLOC, "hipMalloc (box ceilings)");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int32_t rc = OTMB_OK;
    auto fail = [&](const char *what) { rc = otmb_fail(ctx, OTMB_ERR_HIP, what); };
    if (hipMemsetAsync(buf, 0, bytes + 64, ctx->stream) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) fail("box ceilings set-up");
    const int nblk = ncu * 8, passes = (int)(n / ((size_t)nblk * 256 * 8)), reps = 3;  // every byte of the buffer once per launch
    for (int which = 0; which < 2 && rc == OTMB_OK; ++which) {
        auto launch = [&] {
            if (which == 0) hipLaunchKernelGGL(diag_read_kernel, dim3(nblk), dim3(256), 0, ctx->stream, (const d2 *)buf, n, passes, (double *)(buf + bytes));
            else hipLaunchKernelGGL(diag_write_kernel, dim3(nblk), dim3(256), 0, ctx->stream, (d2 *)buf, n, passes);
        };
        launch(); launch();
        float ms = 0.f;
        if (hipEventRecord(e0, ctx->stream) != hipSuccess) { fail("event"); break; }
        for (int r = 0; r < reps; ++r) launch();
        if (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) { fail("box ceilings run"); break; }
        const double gbs = (double)((size_t)nblk * 256 * 8 * passes * sizeof(d2)) * reps / (ms * 1e-3) / 1e9;
        if (which == 0) *read_gbs = gbs; else *write_gbs = gbs;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(buf);
    return rc;
}

// ---- the same question for a kernel's OWN arrays -----------------------------------------------------------------------------------------
// The two plain streams above do not see what makes one box, one process or one set of allocations slow (profiles/r05: 7.0 / 5.2 TB/s on a
// box where the fill pass takes 5.9 ms and on one where it takes 7.3 ms): the fill pass streams through ~30 arrays at once, and it is how
// THOSE lie in the HBM that differs (profiles/r04 section 12).  otmb_ctx_stream_mix runs the plainest kernel there is over the very arrays a
// kernel reads and writes -- every input read once, every output written once (CONTENTS DESTROYED), with the kernel's own granularity: the
// arrays are cut into `tiles` proportional slices, a workgroup takes slice t of every array (contiguous 16-byte-per-lane accesses, all loads
// before the stores), XCD x takes the x-th contiguous eighth of the tiles.  bytes / time is what an ideal streaming kernel with this byte mix
// reaches on these arrays; bench.py reports the fill pass's rate as a fraction of it (roofline.box.frac_of_mix).
#define DIAG_MAXARR 24
struct DiagMixArgs {
    const char *in[DIAG_MAXARR];
    char *out[DIAG_MAXARR];
    unsigned long long in_el[DIAG_MAXARR], out_el[DIAG_MAXARR];  // 16-byte elements
    int n_in, n_out;
    unsigned tiles;
};
__global__ __launch_bounds__(256) void diag_mix_kernel(const DiagMixArgs a, double *sink) {
    // XCD-contiguous eighths of the tile sequence, as the fill pass takes its tiles
    const unsigned nb = a.tiles, q = nb / 8, r = nb % 8, x = blockIdx.x % 8, y = blockIdx.x / 8;
    if (y >= q + (x < r ? 1u : 0u)) return;
    const unsigned long long t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    // slice t of array k: elements [t L_k, (t + 1) L_k) with L_k = ceil(elements / tiles), cut at the array's end.  Every load of a round
    // (one 16-byte element per lane and array) is issued before anything waits for one: up to DIAG_MAXARR requests in flight per lane.
    double acc = 0;
    unsigned long long lmax = 0;
#pragma unroll
    for (int k = 0; k < DIAG_MAXARR; ++k)
        if (k < a.n_in) { const unsigned long long L = (a.in_el[k] + nb - 1) / nb; lmax = L > lmax ? L : lmax; }
    for (unsigned long long it = 0; it < lmax; it += 256) {
        d2 v[DIAG_MAXARR];
#pragma unroll
        for (int k = 0; k < DIAG_MAXARR; ++k) {
            v[k] = d2{0.0, 0.0};
            if (k < a.n_in) {
                const unsigned long long L = (a.in_el[k] + nb - 1) / nb, off = it + threadIdx.x, e = t * L + off;
                if (off < L && e < a.in_el[k]) v[k] = __builtin_nontemporal_load((const d2 *)a.in[k] + e);
            }
        }
#pragma unroll
        for (int k = 0; k < DIAG_MAXARR; ++k) acc += v[k].x;
    }
    const d2 w = {acc, (double)t};
    lmax = 0;
#pragma unroll
    for (int k = 0; k < DIAG_MAXARR; ++k)
        if (k < a.n_out) { const unsigned long long L = (a.out_el[k] + nb - 1) / nb; lmax = L > lmax ? L : lmax; }
    for (unsigned long long it = 0; it < lmax; it += 256) {
#pragma unroll
        for (int k = 0; k < DIAG_MAXARR; ++k) {
            if (k < a.n_out) {
                const unsigned long long L = (a.out_el[k] + nb - 1) / nb, off = it + threadIdx.x, e = t * L + off;
                if (off < L && e < a.out_el[k]) __builtin_nontemporal_store(w, (d2 *)a.out[k] + e);
            }
        }
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}

extern "C" int32_t otmb_ctx_stream_mix(otmb_ctx *ctx, int32_t n_in, const void *const *in, const int64_t *in_bytes, int32_t n_out,
                                       void *const *out, const int64_t *out_bytes, int64_t tiles, double *gbs) {
    if (!ctx || !gbs || n_in < 0 || n_out < 0 || n_in > DIAG_MAXARR || n_out > DIAG_MAXARR || tiles < 8 || tiles >= (1ll << 31) ||
        (n_in && (!in || !in_bytes)) || (n_out && (!out || !out_bytes)))
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "otmb_ctx_stream_mix");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DiagMixArgs a;
    memset(&a, 0, sizeof a);
    a.n_in = n_in; a.n_out = n_out; a.tiles = (unsigned)tiles;
    double total = 0;
    for (int k = 0; k < n_in; ++k) {
        if (!in[k] || in_bytes[k] < 0 || ((size_t)in[k] & 15)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "otmb_ctx_stream_mix: input (16-byte aligned, non-null)");
        a.in[k] = (const char *)in[k]; a.in_el[k] = (unsigned long long)in_bytes[k] / 16; total += 16.0 * (double)a.in_el[k];
    }
    for (int k = 0; k < n_out; ++k) {
        if (!out[k] || out_bytes[k] < 0 || ((size_t)out[k] & 15)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "otmb_ctx_stream_mix: output (16-byte aligned, non-null)");
        a.out[k] = (char *)out[k]; a.out_el[k] = (unsigned long long)out_bytes[k] / 16; total += 16.0 * (double)a.out_el[k];
    }
    double *sink = nullptr;
    if (hipMalloc((void **)&sink, 64) != hipSuccess) return otmb_fail(ctx, OTMB_ERR_ALLOC, "hipMalloc (stream mix)");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int32_t rc = OTMB_OK;
    const unsigned grid = (unsigned)((tiles + 7) / 8 * 8);
    const int reps = 3;
    float ms = 0.f;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = otmb_fail(ctx, OTMB_ERR_HIP, "stream mix set-up");
    if (rc == OTMB_OK) {
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(diag_mix_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, sink);
        (void)hipEventRecord(e0, ctx->stream);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(diag_mix_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, sink);
        if (hipEventRecord(e1, ctx->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess ||
            hipGetLastError() != hipSuccess)
            rc = otmb_fail(ctx, OTMB_ERR_HIP, "stream mix run");
    }
    if (rc == OTMB_OK) *gbs = total * reps / (ms * 1e-3) / 1e9;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(sink);
    return rc;
}
