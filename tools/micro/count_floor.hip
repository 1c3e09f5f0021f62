// Micro-benchmark (tools only): what does a counting pass of the 1-degree grid's shape cost at least?
// 11 920 tiles of 256 wet cells; per cell one 8-byte index load (coalesced) and one dependent 2-byte gather, a wave
// reduction, one barrier, five 4-byte results per tile.  Variants: empty kernel | one tile per workgroup | persistent
// workgroups striding over the tiles (next tile's index loads issued before the current tile is reduced).
//   hipcc --offload-arch=gfx950 -O3 -o count_floor count_floor.hip && ./count_floor
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef long long i64;
typedef unsigned long long u64;

struct Fat { const i64 *lwet; const unsigned short *mask; unsigned *sums; i64 n, G; double pad[48]; };  // a kernarg block as fat as TmParams

__global__ __launch_bounds__(256) void k_empty(Fat p) {}

__device__ __forceinline__ u64 unpack(unsigned word) {
    return (u64)(word & 7u) | ((u64)((word >> 3) & 7u) << 11) | ((u64)((word >> 6) & 7u) << 22) | ((u64)((word >> 9) & 3u) << 33) |
           ((u64)((word >> 11) & 3u) << 43);
}

__device__ __forceinline__ void tile_out(const Fat &p, i64 tile, u64 x, u64 (*wave_tot)[4], int slot) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    if (lane == 0) wave_tot[slot][wid] = x;
    __syncthreads();
    if (tid < 5) {
        const u64 all = wave_tot[slot][0] + wave_tot[slot][1] + wave_tot[slot][2] + wave_tot[slot][3];
        const unsigned sh = (tid == 0) ? 0 : (tid == 1) ? 11 : (tid == 2) ? 22 : (tid == 3) ? 33 : 43;
        p.sums[tile * 5 + tid] = (unsigned)((all >> sh) & 0x7ffu);
    }
}

__global__ __launch_bounds__(256) void k_tile(Fat p) {
    __shared__ u64 wave_tot[1][4];
    const i64 tile = blockIdx.x, w = tile * 256 + threadIdx.x;
    u64 x = 0;
    if (w < p.n) x = unpack(p.mask[p.lwet[w] - 1]);
    tile_out(p, tile, x, wave_tot, 0);
}

__global__ __launch_bounds__(256) void k_persistent(Fat p, i64 ntiles) {
    __shared__ u64 wave_tot[2][4];
    i64 tile = blockIdx.x;
    if (tile >= ntiles) return;
    i64 w = tile * 256 + threadIdx.x;
    i64 L = (w < p.n) ? p.lwet[w] - 1 : -1;
    int slot = 0;
    while (true) {
        const i64 next = tile + gridDim.x;
        i64 Ln = -1;
        if (next < ntiles) {
            const i64 wn = next * 256 + threadIdx.x;
            if (wn < p.n) Ln = p.lwet[wn] - 1;
        }
        const u64 x = (L >= 0) ? unpack(p.mask[L]) : 0;
        tile_out(p, tile, x, wave_tot, slot);
        slot ^= 1;
        if (next >= ntiles) break;
        tile = next;
        L = Ln;
    }
}

int main() {
    const i64 G = 360ll * 300 * 50, n = 3051515, ntiles = (n + 255) / 256;
    std::vector<i64> lwet(n);
    std::vector<unsigned short> mask(G);
    u64 s = 88172645463325252ull;
    i64 L = 0;
    for (i64 w = 0; w < n; ++w) {  // ascending indices with gaps (57 % wet)
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        L += 1 + ((s & 3) == 0) + ((s & 12) == 0) * 2;
        if (L > G) L = G;
        lwet[w] = L;
    }
    for (i64 c = 0; c < G; ++c) mask[c] = (unsigned short)((c * 2654435761u) & 0x1fff);
    Fat p{};
    i64 *d_lwet; unsigned short *d_mask; unsigned *d_sums;
    CK(hipMalloc(&d_lwet, n * 8)); CK(hipMalloc(&d_mask, G * 2)); CK(hipMalloc(&d_sums, ntiles * 5 * 4));
    CK(hipMemcpy(d_lwet, lwet.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_mask, mask.data(), G * 2, hipMemcpyHostToDevice));
    p.lwet = d_lwet; p.mask = d_mask; p.sums = d_sums; p.n = n; p.G = G;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 50;
    auto timeit = [&](const char *name, auto launch) {
        for (int r = 0; r < 5; ++r) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < REP; ++r) launch();
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.4f ms\n", name, ms / REP);
    };
    timeit("empty, 11920 x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(ntiles), dim3(256), 0, 0, p); });
    timeit("empty, 1024 x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, 0, p); });
    timeit("one tile per workgroup", [&] { hipLaunchKernelGGL(k_tile, dim3(ntiles), dim3(256), 0, 0, p); });
    std::vector<unsigned> ref(ntiles * 5), got(ntiles * 5);
    CK(hipMemcpy(ref.data(), d_sums, ntiles * 20, hipMemcpyDeviceToHost));
    for (int wgs : {512, 1024, 2048, 4096}) {
        char name[64];
        snprintf(name, sizeof name, "persistent, %d workgroups", wgs);
        CK(hipMemset(d_sums, 0, ntiles * 20));
        timeit(name, [&] { hipLaunchKernelGGL(k_persistent, dim3(wgs), dim3(256), 0, 0, p, ntiles); });
        CK(hipMemcpy(got.data(), d_sums, ntiles * 20, hipMemcpyDeviceToHost));
        if (got != ref) printf("   MISMATCH\n");
    }
    return 0;
}
