// Micro-benchmark (tools only): does the rate at which ONE buffer can be written / read depend on which allocation it is?  NB buffers of SZ bytes
// (one hipMalloc each, like torch's allocator gives every large tensor), each written front to back (non-temporal, 16 bytes per lane) and read
// front to back by the whole chip, ROUNDS times in turn.  profiles/r04/README.md section 12.
//   hipcc --offload-arch=gfx950 -O3 -o buffer_lottery buffer_lottery.hip && ./buffer_lottery [NB] [MiB per buffer]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void wr(d2 *__restrict__ buf, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (size_t e0 = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; e0 < n; e0 += stride) {
        const d2 x = {(double)e0, 1.0};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + (size_t)u * 256 < n) __builtin_nontemporal_store(x, buf + e0 + (size_t)u * 256);
    }
}
__global__ __launch_bounds__(256) void rd(const d2 *__restrict__ buf, size_t n, double *sink) {
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    double acc = 0;
    for (size_t e0 = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; e0 < n; e0 += stride) {
        d2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = buf[(e0 + (size_t)u * 256 < n) ? e0 + (size_t)u * 256 : e0];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x;
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int nb = argc > 1 ? atoi(argv[1]) : 12;
    const size_t sz = (size_t)(argc > 2 ? atoll(argv[2]) : 3400) << 20;
    std::vector<d2 *> b(nb);
    for (int k = 0; k < nb; ++k) { CK(hipMalloc(&b[k], sz)); CK(hipMemset(b[k], 0, sz)); }
    double *sink;
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n = sz / 16;
    printf("%d buffers of %zu MiB; TB/s written / read, per buffer and round\n", nb, sz >> 20);
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < nb; ++k) {
            float msw = 0, msr = 0;
            hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, b[k], n);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int q = 0; q < 3; ++q) hipLaunchKernelGGL(wr, dim3(2048), dim3(256), 0, 0, b[k], n);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&msw, e0, e1));
            CK(hipEventRecord(e0, 0));
            for (int q = 0; q < 3; ++q) hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, b[k], n, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&msr, e0, e1));
            printf("round %d buffer %2d %p: write %.3f  read %.3f\n", r, k, (void *)b[k], 3.0 * sz / (msw * 1e-3) / 1e12, 3.0 * sz / (msr * 1e-3) / 1e12);
        }
    }
    return 0;
}
