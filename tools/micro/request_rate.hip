// Micro-benchmark (tools only): how many L1 -> L2 REQUESTS per second does the chip sustain, and is the limit a CU's (its vector L1) or
// shared (L2 channels / fabric)?  profiles/r04/README.md section 11: every kernel of the hot path, the traffic model and the ideal
// streaming mix all run at 84-99 G requests/s whatever their bytes per request, occupancy or HBM traffic.
//   reads, 16 bytes per lane  = 1 KB per wave instruction = 8 requests of 128 bytes;
//   reads,  8 bytes per lane  = 512 B per wave instruction = 4 requests of 128 bytes (the fill pass's gathers are of this kind);
//   writes, 16 bytes per lane = 1 KB per wave instruction = 16 requests of 64 bytes (non-temporal, like the matrices);
// each over a footprint that stays in every XCD's L2 (2 MB: hits only) and over 2 GiB (HBM), on all CUs and under two CU masks
// (every second CU id / the lower half of the ids); a probe kernel reports which XCDs and CUs a mask really selects.
//   hipcc --offload-arch=gfx950 -O3 -o request_rate request_rate.hip && ./request_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

// Workgroup b reads, in pass q, UNR consecutive pieces of 256 elements at element ((q * gridDim.x + b) * UNR + u) * 256 of the footprint
// (n elements, a power of two: the walk wraps around it), UNR independent loads in flight per thread: every wave instruction is one
// contiguous piece (64 lanes x sizeof(T)), every element of the footprint is read equally often.
template <typename T, int UNR>
__global__ __launch_bounds__(256) void reads(const T *__restrict__ buf, size_t n, int passes, double *sink) {
    double acc = 0;
    for (int q = 0; q < passes; ++q) {
        const size_t e0 = (((size_t)q * gridDim.x + blockIdx.x) * UNR) * 256 + threadIdx.x;
        T v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = buf[(e0 + (size_t)u * 256) & (n - 1)];
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += ((const double *)&v[u])[0];
    }
    if (acc == 1.2345e-300) sink[0] = acc;
}
template <int UNR>
__global__ __launch_bounds__(256) void writes(d2 *__restrict__ buf, size_t n, int passes) {
    for (int q = 0; q < passes; ++q) {
        const size_t e0 = (((size_t)q * gridDim.x + blockIdx.x) * UNR) * 256 + threadIdx.x;
        const d2 x = {(double)q, (double)e0};
#pragma unroll
        for (int u = 0; u < UNR; ++u) __builtin_nontemporal_store(x, buf + ((e0 + (size_t)u * 256) & (n - 1)));
    }
}
// write sweep: NS streams (regions `gap` elements apart, each written front to back by all workgroups together, like the fill pass's ten
// output arrays), UNR stores of 16 bytes in flight per thread and stream piece, plain or non-temporal
template <bool NT, int UNR, int NS>
__global__ __launch_bounds__(256) void writes_sweep(d2 *__restrict__ buf, size_t n_per_stream, size_t gap, int passes) {
    for (int q = 0; q < passes; ++q) {
        const size_t e0 = (((size_t)q * gridDim.x + blockIdx.x) * UNR) * 256 + threadIdx.x;
        const d2 x = {(double)q, (double)e0};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                d2 *dst = buf + (size_t)s * gap + ((e0 + (size_t)u * 256) & (n_per_stream - 1));
                if (NT) __builtin_nontemporal_store(x, dst); else *dst = x;
            }
        }
    }
}
// which XCD / CU does each workgroup run on?  HW_REG_HW_ID (4): cu_id [11:8], sh_id [12], se_id [15:13]; HW_REG_XCC_ID (20)
__global__ void probe(unsigned *out) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        out[blockIdx.x] = ((xcc & 0xf) << 16) | ((hw >> 8) & 0xff);
    }
    // keep the workgroup alive long enough for every CU to take some
    for (int q = 0; q < 200; ++q) __builtin_amdgcn_s_sleep(10);
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    // ONE mask per invocation (default 0 = all CUs).  On this pool (ROCm 7.2, ordinary user) hipExtStreamCreateWithCUMask is of no use: the
    // mask "every second CU id" was ignored (workgroups still ran on all 256 CUs, same rates) and creating the stream for "lower half of
    // the CU ids" never returned (profiles/r04/call27_request_rate.log) -- masks 1 ... 3 are kept for other systems, run them under `timeout`.
    const int only = argc > 1 ? atoi(argv[1]) : 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("%s: %d CUs, %d MHz\n", prop.name, ncu, prop.clockRate / 1000);
    const size_t big = (size_t)2 << 30, small = (size_t)2 << 20;
    char *buf;
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 0, big));
    double *sink;
    CK(hipMalloc(&sink, 64));
    unsigned *pr;
    const int NPROBE = 4096;
    CK(hipMalloc(&pr, NPROBE * sizeof(unsigned)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    struct Mask { const char *name; std::vector<uint32_t> bits; };
    std::vector<Mask> masks;
    const int nw = (ncu + 31) / 32;
    masks.push_back({"all CUs", {}});
    { std::vector<uint32_t> m(nw, 0x55555555u); masks.push_back({"every second CU id", m}); }
    { std::vector<uint32_t> m(nw, 0u); for (int c = 0; c < ncu / 2; ++c) m[c / 32] |= 1u << (c % 32); masks.push_back({"lower half of the CU ids", m}); }
    { std::vector<uint32_t> m(nw, 0x11111111u); masks.push_back({"every fourth CU id", m}); }

    if (only == 100) {  // write sweep on all CUs: total 8 GiB written per launch over a 8 GiB buffer (far beyond L2 + Infinity Cache)
        char *wb;
        const size_t tot = (size_t)8 << 30;
        CK(hipMalloc(&wb, tot));
        CK(hipMemset(wb, 0, tot));
        auto run = [&](const char *name, int nblk, int unr, int ns, auto launch) {
            for (int r = 0; r < 2; ++r) launch();
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, 0);
            const int REP = 5;
            for (int r = 0; r < REP; ++r) launch();
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  writes %-14s %5d workgroups, %2d stores in flight x %2d streams: %8.3f ms  %6.2f TB/s\n", name, nblk, unr, ns, ms / REP, (double)tot / (ms * 1e-3 / REP) / 1e12);
        };
#define SWEEP(NT, UNR, NS, NBLK) { const size_t nps = tot / 16 / NS; const int passes = (int)(nps / ((size_t)(NBLK) * 256 * UNR)); \
        run(NT ? "non-temporal" : "plain", NBLK, UNR, NS, [&] { hipLaunchKernelGGL((writes_sweep<NT, UNR, NS>), dim3(NBLK), dim3(256), 0, 0, (d2 *)wb, nps, nps, passes); }); }
        for (int nblk : {ncu * 2, ncu * 4, ncu * 8, ncu * 16}) {
            SWEEP(true, 8, 1, nblk) SWEEP(false, 8, 1, nblk)
        }
        SWEEP(true, 2, 1, ncu * 8) SWEEP(false, 2, 1, ncu * 8) SWEEP(true, 16, 1, ncu * 8) SWEEP(false, 16, 1, ncu * 8)
        SWEEP(true, 4, 2, ncu * 8) SWEEP(false, 4, 2, ncu * 8) SWEEP(true, 2, 8, ncu * 8) SWEEP(false, 2, 8, ncu * 8)
        SWEEP(true, 1, 16, ncu * 8) SWEEP(false, 1, 16, ncu * 8) SWEEP(true, 4, 8, ncu * 4) SWEEP(false, 4, 8, ncu * 4)
        SWEEP(true, 2, 8, ncu * 3) SWEEP(false, 2, 8, ncu * 3)
        return 0;
    }
    if (only == 200) {  // scaling with the number of busy CUs: NW persistent workgroups (one per CU while NW <= CUs), the same streams
        char *wb;
        const size_t tot = (size_t)4 << 30;
        CK(hipMalloc(&wb, tot));
        CK(hipMemset(wb, 0, tot));
        for (int nw : {16, 32, 64, 128, 256, 512, 1024, 2048}) {
            // where do NW workgroups of 256 threads land?
            CK(hipMemset(pr, 0xff, NPROBE * sizeof(unsigned)));
            hipLaunchKernelGGL(probe, dim3(nw), dim3(256), 0, 0, pr);
            CK(hipDeviceSynchronize());
            std::vector<unsigned> h(nw);
            CK(hipMemcpy(h.data(), pr, nw * sizeof(unsigned), hipMemcpyDeviceToHost));
            std::set<unsigned> xcds, cus;
            for (unsigned v : h) { xcds.insert(v >> 16); cus.insert(v); }
            const int active = (int)cus.size();
            auto run = [&](const char *name, double bytes, double req_bytes, auto launch) {
                launch();
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, 0);
                const int REP = 3;
                for (int r = 0; r < REP; ++r) launch();
                (void)hipEventRecord(e1, 0);
                (void)hipDeviceSynchronize();
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                const double sec = ms * 1e-3 / REP;
                printf("  %4d workgroups on %3d CUs of %zu XCDs: %-34s %7.2f TB/s  %7.1f G requests/s  %6.3f per busy CU and cycle\n", nw, active, xcds.size(), name, bytes / sec / 1e12,
                       bytes / req_bytes / sec / 1e9, bytes / req_bytes / sec / ((double)active * prop.clockRate * 1e3));
            };
            const int PH = 2048, PB = (int)(((size_t)1 << 30) / ((size_t)nw * 256 * 8 * 16)) > 4096 ? 4096 : (int)(((size_t)1 << 30) / ((size_t)nw * 256 * 8 * 16));
            auto bytes = [&](int passes) { return (double)passes * nw * 256 * 8 * 16; };
            run("reads 16 B/lane, L2 hits (2 MB)", bytes(PH), 128, [&] { hipLaunchKernelGGL((reads<d2, 8>), dim3(nw), dim3(256), 0, 0, (const d2 *)wb, small / 16, PH, sink); });
            run("reads 16 B/lane, HBM (4 GiB)", bytes(PB), 128, [&] { hipLaunchKernelGGL((reads<d2, 8>), dim3(nw), dim3(256), 0, 0, (const d2 *)wb, tot / 16, PB, sink); });
            run("writes 16 B/lane nt, HBM (4 GiB)", bytes(PB), 64, [&] { hipLaunchKernelGGL((writes<8>), dim3(nw), dim3(256), 0, 0, (d2 *)wb, tot / 16, PB); });
        }
        return 0;
    }
    for (size_t mi = 0; mi < masks.size(); ++mi) {
        auto &mk = masks[mi];
        if (only >= 0 && (int)mi != only) continue;
        printf("-- %s: creating the stream --\n", mk.name);
        hipStream_t st;
        if (mk.bits.empty()) CK(hipStreamCreate(&st));
        else CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mk.bits.size(), mk.bits.data()));
        // what the mask selects
        CK(hipMemsetAsync(pr, 0xff, NPROBE * sizeof(unsigned), st));
        printf("-- probe --\n");
        hipLaunchKernelGGL(probe, dim3(NPROBE), dim3(64), 0, st, pr);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> h(NPROBE);
        CK(hipMemcpy(h.data(), pr, NPROBE * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::set<unsigned> xcds, cus;
        for (unsigned v : h) { xcds.insert(v >> 16); cus.insert(v); }
        printf("== %s: workgroups ran on %zu XCDs, %zu distinct (XCD, SE/SH/CU) places ==\n", mk.name, xcds.size(), cus.size());
        const int active = (int)cus.size();
        const int nblk = ncu * 8;
        auto timeit = [&](const char *name, double bytes, double req_bytes, auto launch) {
            for (int r = 0; r < 3; ++r) launch();
            (void)hipStreamSynchronize(st);
            (void)hipEventRecord(e0, st);
            const int REP = 10;
            for (int r = 0; r < REP; ++r) launch();
            (void)hipEventRecord(e1, st);
            (void)hipStreamSynchronize(st);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double s = ms * 1e-3 / REP;
            printf("  %-58s %8.3f ms  %7.2f TB/s  %7.1f G requests/s  %6.3f requests per active CU and cycle\n", name, s * 1e3, bytes / s / 1e12, bytes / req_bytes / s / 1e9,
                   bytes / req_bytes / s / ((double)active * prop.clockRate * 1e3));
        };
        // bytes per launch = passes x workgroups x 256 threads x 8 loads x sizeof(element)
        auto bytes = [&](int passes, int elem) { return (double)passes * nblk * 256 * 8 * elem; };
        const int PS = 64, PB16 = (int)(big / ((size_t)nblk * 256 * 8 * 16)), PB8 = (int)(big / ((size_t)nblk * 256 * 8 * 8));
        timeit("reads, 16 B per lane, 2 MB footprint (L2 hits)", bytes(PS, 16), 128, [&] { hipLaunchKernelGGL((reads<d2, 8>), dim3(nblk), dim3(256), 0, st, (const d2 *)buf, small / 16, PS, sink); });
        timeit("reads,  8 B per lane, 2 MB footprint (L2 hits)", bytes(PS, 8), 128, [&] { hipLaunchKernelGGL((reads<double, 8>), dim3(nblk), dim3(256), 0, st, (const double *)buf, small / 8, PS, sink); });
        timeit("reads, 16 B per lane, 2 GiB (HBM)", bytes(PB16, 16), 128, [&] { hipLaunchKernelGGL((reads<d2, 8>), dim3(nblk), dim3(256), 0, st, (const d2 *)buf, big / 16, PB16, sink); });
        timeit("reads,  8 B per lane, 2 GiB (HBM)", bytes(PB8, 8), 128, [&] { hipLaunchKernelGGL((reads<double, 8>), dim3(nblk), dim3(256), 0, st, (const double *)buf, big / 8, PB8, sink); });
        timeit("writes, 16 B per lane, non-temporal, 2 MB footprint", bytes(PS, 16), 64, [&] { hipLaunchKernelGGL((writes<8>), dim3(nblk), dim3(256), 0, st, (d2 *)buf, small / 16, PS); });
        timeit("writes, 16 B per lane, non-temporal, 2 GiB (HBM)", bytes(PB16, 16), 64, [&] { hipLaunchKernelGGL((writes<8>), dim3(nblk), dim3(256), 0, st, (d2 *)buf, big / 16, PB16); });
        CK(hipStreamDestroy(st));
    }
    return 0;
}
