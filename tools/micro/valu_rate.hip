// Micro-benchmark (tools only): issue cost of the f64 instructions the fill pass is made of, on gfx950.
// One kernel per instruction kind: every wave runs REPS x 16 independent instructions; waves per SIMD = 1 and 4.
// Prints cycles per wave-instruction per SIMD (s_memtime ticks of one wave / instructions).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { K_FMA64, K_MUL64, K_ADD64, K_RCP64, K_DIVSCALE, K_DIVFMAS, K_DIVFIXUP, K_CNDMASK, K_CMP64, K_FMA32, K_BCNT, K_DIV64, K_MIN64, K_NK };
static const char *NAMES[K_NK] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64",
                                  "v_cndmask_b32", "v_cmp_lt_f64", "v_fma_f32", "v_bcnt_u32_b32", "a / b (full IEEE f64 division)", "v_min_f64"};

template <int KIND>
__global__ __launch_bounds__(256) void rate(double *out, int reps, unsigned long long *ticks) {
    double x[16];
    const double seed = 1.0 + threadIdx.x * 1e-3;
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = seed + q;
    double y = 1.000001 + blockIdx.x * 1e-9;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (KIND == K_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[q]) : "v"(y));
            if (KIND == K_MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[q]) : "v"(y));
            if (KIND == K_ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[q]) : "v"(y));
            if (KIND == K_MIN64) asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[q]) : "v"(y));
            if (KIND == K_RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[q]));
            if (KIND == K_DIVSCALE) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(x[q]) : "v"(y) : "vcc");
            if (KIND == K_DIVFMAS) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(x[q]) : "v"(y) : "vcc");
            if (KIND == K_DIVFIXUP) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(x[q]) : "v"(y));
            if (KIND == K_CNDMASK) { float f = (float)x[q]; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f) : "v"((float)y) : "vcc"); x[q] = f; }
            if (KIND == K_CMP64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(x[q]), "v"(y) : "vcc");
            if (KIND == K_FMA32) { float f = (float)x[q]; asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"((float)y)); x[q] = f; }
            if (KIND == K_BCNT) { unsigned f = (unsigned)q; asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(f) : "v"(threadIdx.x)); x[q] += f; }
            if (KIND == K_DIV64) x[q] = y / x[q];
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += x[q];
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND> int run(double *out, unsigned long long *ticks, int wgs_per_cu) {
    const int reps = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate<KIND>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, 10, ticks);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate<KIND>, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, reps, ticks);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long t = 0;
    CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    const double n = (double)reps * 16;
    // s_memtime ticks at 100 MHz on this part: convert through the wall time of the launch
    printf("%-34s waves/SIMD %d: %8.2f ns per wave-instruction per SIMD  (kernel %.3f ms, wave0 %llu ticks = %.2f ticks/instr)\n", NAMES[KIND],
           wgs_per_cu, 1e6 * ms / (n * wgs_per_cu), ms, t, (double)t / n);
    return 0;
}

int main() {
    double *out; unsigned long long *ticks;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&ticks, 64));
    for (int w : {1, 4}) {
        run<K_FMA32>(out, ticks, w); run<K_CNDMASK>(out, ticks, w); run<K_BCNT>(out, ticks, w); run<K_CMP64>(out, ticks, w);
        run<K_ADD64>(out, ticks, w); run<K_MUL64>(out, ticks, w); run<K_FMA64>(out, ticks, w); run<K_MIN64>(out, ticks, w);
        run<K_RCP64>(out, ticks, w); run<K_DIVSCALE>(out, ticks, w); run<K_DIVFMAS>(out, ticks, w); run<K_DIVFIXUP>(out, ticks, w);
        run<K_DIV64>(out, ticks, w);
    }
    return 0;
}
