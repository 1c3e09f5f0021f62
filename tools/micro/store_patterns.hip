// Micro-benchmark (tools only, not part of the library): HBM write rate of the store patterns used by the
// facefluxes / fill kernels on gfx950.  hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));
struct Arr { double *a[8]; };

template <int K> __global__ void flat8(Arr A, long n) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
#pragma unroll
        for (int q = 0; q < K; ++q) A.a[q][t] = (double)t;
    }
}
template <int K> __global__ void flat16(Arr A, long n) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * t + 1 < n) {
#pragma unroll
        for (int q = 0; q < K; ++q) *(f64x2 *)(A.a[q] + 2 * t) = f64x2{(double)t, 1.0};
    }
}
// thread per column, marching the levels (facefluxes pattern); KC = levels per thread chunk
template <int K> __global__ void march8(Arr A, long P, int nz, int kc) {
    long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int k0 = blockIdx.y * kc;
    if (s < P) {
        double x = (double)s;
        for (int k = k0 + kc - 1; k >= k0; --k) {
            if (k < nz) {
#pragma unroll
                for (int q = 0; q < K; ++q) A.a[q][(long)k * P + s] = x;
                x += 1.0;
            }
        }
    }
}
template <int K> __global__ void march16(Arr A, long P, int nz, int kc) {
    long s = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
    int k0 = blockIdx.y * kc;
    if (s + 1 < P) {
        double x = (double)s;
        for (int k = k0 + kc - 1; k >= k0; --k) {
            if (k < nz) {
#pragma unroll
                for (int q = 0; q < K; ++q) *(f64x2 *)(A.a[q] + (long)k * P + s) = f64x2{x, x};
                x += 1.0;
            }
        }
    }
}
// read 2 arrays + write K (column march)
template <int K> __global__ void march_rw(Arr A, const double *u, const double *v, long P, int nz, int kc) {
    long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int k0 = blockIdx.y * kc;
    if (s < P) {
        double x = 0;
        for (int k = k0 + kc - 1; k >= k0; --k) {
            if (k < nz) {
                x += u[(long)k * P + s] - v[(long)k * P + s];
#pragma unroll
                for (int q = 0; q < K; ++q) A.a[q][(long)k * P + s] = x;
            }
        }
    }
}

int main() {
    const long P = 360 * 300; const int nz = 50; const long n = P * nz;
    Arr A; double *u, *v;
    for (int q = 0; q < 8; ++q) CK(hipMalloc(&A.a[q], n * 8));
    CK(hipMalloc(&u, n * 8)); CK(hipMalloc(&v, n * 8));
    CK(hipMemset(u, 0, n * 8)); CK(hipMemset(v, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, double bytes, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        (void)hipEventRecord(e0, 0);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) launch();
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-34s %8.4f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6);
    };
    run("flat8  K=1 (256 thr)", n * 8.0, [&] { flat8<1><<<(n + 255) / 256, 256>>>(A, n); });
    run("flat8  K=6 (256 thr)", n * 48.0, [&] { flat8<6><<<(n + 255) / 256, 256>>>(A, n); });
    run("flat16 K=1 (256 thr)", n * 8.0, [&] { flat16<1><<<(n / 2 + 255) / 256, 256>>>(A, n); });
    run("flat16 K=6 (256 thr)", n * 48.0, [&] { flat16<6><<<(n / 2 + 255) / 256, 256>>>(A, n); });
    for (int kc : {50, 25, 10, 5, 1}) {
        char nm[64];
        dim3 g((P + 63) / 64, (nz + kc - 1) / kc);
        snprintf(nm, sizeof nm, "march8  K=6 64thr kc=%d", kc);
        run(nm, n * 48.0, [&] { march8<6><<<g, 64>>>(A, P, nz, kc); });
        dim3 g2((P / 2 + 63) / 64, (nz + kc - 1) / kc);
        snprintf(nm, sizeof nm, "march16 K=6 64thr kc=%d", kc);
        run(nm, n * 48.0, [&] { march16<6><<<g2, 64>>>(A, P, nz, kc); });
        dim3 g3((P + 255) / 256, (nz + kc - 1) / kc);
        snprintf(nm, sizeof nm, "march8  K=6 256thr kc=%d", kc);
        run(nm, n * 48.0, [&] { march8<6><<<g3, 256>>>(A, P, nz, kc); });
        snprintf(nm, sizeof nm, "march_rw K=6 64thr kc=%d", kc);
        run(nm, n * 64.0, [&] { march_rw<6><<<g, 64>>>(A, u, v, P, nz, kc); });
    }
    return 0;
}
