// Micro-benchmark (tools only, round 4): does WHERE the twenty streams of the fill pass's byte mix live decide how fast an ideal
// kernel moves them?  The same kernel (stream_mix.hip's mix_stream: ten inputs read once, ten outputs written once, 2.25 bytes
// written per byte read, 16 bytes per lane) over buffers obtained in different ways:
//   separate    one hipMalloc per stream (what torch's allocator does for large tensors)
//   arena       ONE hipMalloc, the streams carved out back to back (2 MiB grid)
//   vmm<H>      every stream its own virtual range, mapped from physical handles of H MiB (hipMemCreate / hipMemMap),
//               created stream after stream ("seq") or round-robin over the streams ("rr": physical neighbours belong to different streams)
// bench.py with all arrays of a run carved out of one allocation measured the fill pass 15-19 % slower than with one allocation per
// array, whatever the strides between the arrays (profiles/r04/README.md section 8): is that the platform or the kernel?
//   hipcc --offload-arch=gfx950 -O3 -o placement_mix placement_mix.hip && ./placement_mix [MiB per input stream, default 43]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef long long i64;
typedef double d2 __attribute__((ext_vector_type(2)));
#define NIN 10
#define NOUT 10
struct Mix { const d2 *in[NIN]; d2 *out[NOUT]; i64 n_in; };
static const int CNT_H[5] = {7, 4, 5, 1, 3};

template <bool NT>
__global__ __launch_bounds__(256) void mix_stream(Mix p, i64 nchunks) {
    const int CNT[5] = {7, 4, 5, 1, 3};
    const i64 c = blockIdx.x;
    if (c >= nchunks) return;
    const i64 e0 = c * 1024 + threadIdx.x;
    d2 v[NIN][4];
#pragma unroll
    for (int a = 0; a < NIN; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[a][r] = p.in[a][e0 + r * 256];
    const i64 o0 = c * (i64)256;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r == 2 && threadIdx.x >= 64) continue;
        int e = 0;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
#pragma unroll
            for (int q = 0; q < CNT[m]; ++q) {
                const d2 x = v[(e + r) % NIN][(e + q) & 3];
                const i64 pos = (o0 * 3 + (i64)r * 256) * CNT[m] + (i64)q * 256 + threadIdx.x;
                if (NT) { __builtin_nontemporal_store(x, p.out[2 * m] + pos); __builtin_nontemporal_store(x, p.out[2 * m + 1] + pos); }
                else { p.out[2 * m][pos] = x; p.out[2 * m + 1][pos] = x; }
                ++e;
            }
        }
    }
}

struct Vmm {
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<std::pair<void *, size_t>> ranges;
};

int main(int argc, char **argv) {
    const i64 mib = argc > 1 ? atoll(argv[1]) : 43;
    const i64 n_in = (mib << 20) / 16 / 1024 * 1024, nchunks = n_in / 1024;
    size_t bytes[NIN + NOUT];
    double in_bytes = 0, out_bytes = 0;
    for (int a = 0; a < NIN; ++a) { bytes[a] = (size_t)n_in * 16; in_bytes += n_in * 16.0; }
    for (int m = 0; m < 5; ++m)
        for (int h = 0; h < 2; ++h) {
            bytes[NIN + 2 * m + h] = (size_t)(nchunks * 256 * 3 * CNT_H[m] + 1024) * 16;
            out_bytes += nchunks * 256.0 * 2.25 * CNT_H[m] * 16;
        }
    const size_t M2 = (size_t)2 << 20;
    for (auto &b : bytes) b = (b + M2 - 1) / M2 * M2;
    size_t total = 0;
    for (auto b : bytes) total += b;
    printf("mix: %.3f GB read, %.3f GB written per launch; %d streams, %.2f GB of buffers\n", in_bytes / 1e9, out_bytes / 1e9, NIN + NOUT, total / 1e9);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, const Mix &p) {
        for (int nt = 0; nt < 2; ++nt) {
            auto launch = [&] {
                if (nt) hipLaunchKernelGGL((mix_stream<true>), dim3((unsigned)nchunks), dim3(256), 0, 0, p, nchunks);
                else hipLaunchKernelGGL((mix_stream<false>), dim3((unsigned)nchunks), dim3(256), 0, 0, p, nchunks);
            };
            for (int r = 0; r < 3; ++r) launch();
            (void)hipDeviceSynchronize();
            float best = 1e30f, sum = 0;
            const int rep = 10;
            for (int r = 0; r < rep; ++r) {
                (void)hipEventRecord(e0);
                launch();
                (void)hipEventRecord(e1);
                (void)hipDeviceSynchronize();
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
                sum += ms;
            }
            printf("%-34s %-12s mean %8.4f ms  best %8.4f ms  %6.2f TB/s\n", name, nt ? "nt stores" : "plain stores", sum / rep, best,
                   (in_bytes + out_bytes) / (sum / rep * 1e-3) / 1e12);
            fflush(stdout);
        }
    };
    auto fill = [&](void *ptrs[], Mix &p) {
        for (int a = 0; a < NIN; ++a) p.in[a] = (const d2 *)ptrs[a];
        for (int o = 0; o < NOUT; ++o) p.out[o] = (d2 *)ptrs[NIN + o];
        p.n_in = n_in;
    };
    // ---- separate ----
    {
        void *ptrs[NIN + NOUT];
        for (int s = 0; s < NIN + NOUT; ++s) { CK(hipMalloc(&ptrs[s], bytes[s])); CK(hipMemset(ptrs[s], 0, bytes[s])); }
        Mix p{};
        fill(ptrs, p);
        timeit("separate (one hipMalloc per stream)", p);
        for (auto q : ptrs) CK(hipFree(q));
    }
    // ---- arena ----
    {
        char *base;
        CK(hipMalloc(&base, total));
        CK(hipMemset(base, 0, total));
        void *ptrs[NIN + NOUT];
        size_t off = 0;
        for (int s = 0; s < NIN + NOUT; ++s) { ptrs[s] = base + off; off += bytes[s]; }
        Mix p{};
        fill(ptrs, p);
        timeit("arena (one hipMalloc)", p);
        CK(hipFree(base));
    }
    // ---- arena, after a large dummy allocation (the arena lands elsewhere in the physical address space) ----
    {
        char *dummy, *base;
        CK(hipMalloc(&dummy, (size_t)16 << 30));
        CK(hipMalloc(&base, total));
        CK(hipMemset(base, 0, total));
        void *ptrs[NIN + NOUT];
        size_t off = 0;
        for (int s = 0; s < NIN + NOUT; ++s) { ptrs[s] = base + off; off += bytes[s]; }
        Mix p{};
        fill(ptrs, p);
        timeit("arena behind a 16 GiB allocation", p);
        CK(hipFree(base)); CK(hipFree(dummy));
    }
    // ---- virtual memory management: physical handles of H MiB ----
    int dev = 0;
    CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) {
        printf("virtual memory management: not available on this platform\n");
        return 0;
    }
    printf("virtual memory management: minimum granularity %zu bytes\n", gran);
    for (size_t hm : {(size_t)2, (size_t)32, (size_t)256})
        for (int rr = 0; rr < 2; ++rr) {
            const size_t H = std::max(hm << 20, gran);
            Vmm vm;
            void *ptrs[NIN + NOUT];
            std::vector<size_t> need(NIN + NOUT);
            bool ok = true;
            for (int s = 0; s < NIN + NOUT && ok; ++s) {
                need[s] = (bytes[s] + H - 1) / H * H;
                ok = hipMemAddressReserve(&ptrs[s], need[s], H, nullptr, 0) == hipSuccess;
                if (ok) vm.ranges.push_back({ptrs[s], need[s]});
            }
            // creation order = physical order (roughly): stream after stream, or round-robin over the streams
            std::vector<std::pair<int, size_t>> order;  // (stream, offset)
            if (!rr) {
                for (int s = 0; s < NIN + NOUT; ++s)
                    for (size_t o = 0; o < need[s]; o += H) order.push_back({s, o});
            } else {
                size_t maxn = *std::max_element(need.begin(), need.end());
                for (size_t o = 0; o < maxn; o += H)
                    for (int s = 0; s < NIN + NOUT; ++s)
                        if (o < need[s]) order.push_back({s, o});
            }
            for (auto &so : order) {
                if (!ok) break;
                hipMemGenericAllocationHandle_t h;
                ok = hipMemCreate(&h, H, &prop, 0) == hipSuccess;
                if (!ok) break;
                vm.handles.push_back(h);
                ok = hipMemMap((char *)ptrs[so.first] + so.second, H, 0, h, 0) == hipSuccess;
            }
            hipMemAccessDesc acc = {};
            acc.location.type = hipMemLocationTypeDevice;
            acc.location.id = dev;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            for (auto &r : vm.ranges)
                if (ok) ok = hipMemSetAccess(r.first, r.second, &acc, 1) == hipSuccess;
            char name[96];
            snprintf(name, sizeof name, "vmm %zu MiB handles, %s", H >> 20, rr ? "round-robin" : "stream by stream");
            if (!ok) {
                printf("%-34s failed: %s\n", name, hipGetErrorString(hipGetLastError()));
            } else {
                for (int s = 0; s < NIN + NOUT; ++s) (void)hipMemset(ptrs[s], 0, bytes[s]);
                Mix p{};
                fill(ptrs, p);
                timeit(name, p);
            }
            (void)hipDeviceSynchronize();
            for (auto &r : vm.ranges) { (void)hipMemUnmap(r.first, r.second); (void)hipMemAddressFree(r.first, r.second); }
            for (auto h : vm.handles) (void)hipMemRelease(h);
        }
    return 0;
}
