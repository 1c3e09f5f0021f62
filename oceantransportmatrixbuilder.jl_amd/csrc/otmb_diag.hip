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
