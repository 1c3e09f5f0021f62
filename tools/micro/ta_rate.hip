// Micro-benchmark (tools only): vector memory instruction throughput per CU on gfx950 for 1/4/8/16 bytes per lane,
// loads from an L2-resident buffer and stores to one, coalesced.  Answers: is a kernel with many narrow accesses
// bound by the number of VMEM instructions (address processing) or by bytes?
#include <hip/hip_runtime.h>

#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ float sum(T x);
template <> __device__ float sum(unsigned char x) { return x; }
template <> __device__ float sum(float x) { return x; }
template <> __device__ float sum(f32x2 x) { return x.x + x.y; }
template <> __device__ float sum(f32x4 x) { return x.x + x.y + x.z + x.w; }

// each wave reads `reps` times 64 consecutive elements; the window (elements) keeps the footprint in L2
template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void loads(const T *__restrict__ a, long window, int reps, float *out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0;
    long idx = t % window;
    for (int r = 0; r < reps; r += UNROLL) {
        T v[UNROLL];
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) {
            v[q] = a[idx];
            idx += 64 * 1031;  // a different line each time
            if (idx >= window) idx -= window;
        }
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) acc += sum(v[q]);
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void stores(T *__restrict__ a, long window, int reps, T val) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long idx = t % window;
    for (int r = 0; r < reps; r += UNROLL) {
#pragma unroll
        for (int q = 0; q < UNROLL; ++q) {
            a[idx] = val;
            idx += 64 * 1031;
            if (idx >= window) idx -= window;
        }
    }
}

int main() {
    const size_t bytes = 16u << 20;  // 16 MiB: L2-resident across the 8 XCDs (4 MiB each) is optimistic; MALL-resident surely
    void *buf; float *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double ghz = pr.clockRate * 1e-6;
    printf("CUs %d clock %.2f GHz\n", cus, ghz);
    const int blocks = cus * 8, reps = 2048;
    auto run = [&](const char *name, int bpl, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        (void)hipEventRecord(e0, 0);
        const int n = 5;
        for (int r = 0; r < n; ++r) launch();
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= n;
        const double winstr = (double)blocks * 4 * reps;  // wave-level instructions
        const double cyc_per_instr_per_cu = ms * 1e-3 * ghz * 1e9 / (winstr / cus);
        printf("%-22s %8.3f ms  %6.1f cycles/wave-instr/CU  %8.1f GB/s\n", name, ms, cyc_per_instr_per_cu, winstr * 64 * bpl / ms * 1e-6);
    };
    for (size_t win : {(size_t)(2u << 20), bytes}) {
        printf("window %zu MiB\n", win >> 20);
        run("load  1 B/lane", 1, [&] { loads<unsigned char, 8><<<blocks, 256>>>((const unsigned char *)buf, (long)win, reps, out); });
        run("load  4 B/lane", 4, [&] { loads<float, 8><<<blocks, 256>>>((const float *)buf, (long)(win / 4), reps, out); });
        run("load  8 B/lane", 8, [&] { loads<f32x2, 8><<<blocks, 256>>>((const f32x2 *)buf, (long)(win / 8), reps, out); });
        run("load 16 B/lane", 16, [&] { loads<f32x4, 8><<<blocks, 256>>>((const f32x4 *)buf, (long)(win / 16), reps, out); });
        run("store 1 B/lane", 1, [&] { stores<unsigned char, 8><<<blocks, 256>>>((unsigned char *)buf, (long)win, reps, (unsigned char)1); });
        run("store 4 B/lane", 4, [&] { stores<float, 8><<<blocks, 256>>>((float *)buf, (long)(win / 4), reps, 1.0f); });
        run("store 8 B/lane", 8, [&] { stores<f32x2, 8><<<blocks, 256>>>((f32x2 *)buf, (long)(win / 8), reps, f32x2{1, 2}); });
        run("store 16 B/lane", 16, [&] { stores<f32x4, 8><<<blocks, 256>>>((f32x4 *)buf, (long)(win / 16), reps, f32x4{1, 2, 3, 4}); });
    }
    return 0;
}
