// Micro-benchmark (tools only): what this chip does with the fill pass's BYTE MIX when the access pattern is ideal --
// ten dense input streams read once with 16 bytes per lane, ten output streams written once with 16 bytes per lane,
// 2.25 bytes written per byte read (0.25 degree grid: 9.3 GB in, 22.2 GB out) -- and the plain read / write / copy rates
// beside it.  It is the ceiling any formulation of the fill pass can reach, and (under rocprofv3 --pmc FETCH_SIZE /
// WRITE_SIZE) the calibration of those two counters on known byte counts: a streaming read of 8 and of 16 bytes per
// lane over 2 GiB (far beyond the 256 MiB Infinity Cache), and an 8-byte-per-lane GATHER through an index list with
// the gaps of a wet mask (the fill pass's own pattern).
//   hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip && ./stream_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef long long i64;
typedef double d2 __attribute__((ext_vector_type(2)));

#define NIN 10
#define NOUT 10
struct Mix {
    const d2 *in[NIN];
    d2 *out[NOUT];
    i64 n_in;  // d2 elements per input stream
};
static const int CNT_H[5] = {7, 4, 5, 1, 3};

__global__ __launch_bounds__(256) void calib_read8(const double *__restrict__ a, i64 n, double *sink) {
    const i64 stride = (i64)gridDim.x * 256;
    double s = 0;
    for (i64 q = (i64)blockIdx.x * 256 + threadIdx.x; q < n; q += stride) s += a[q];
    if (s == 1.2345e300) *sink = s;
}
__global__ __launch_bounds__(256) void calib_read16(const d2 *__restrict__ a, i64 n, double *sink) {
    const i64 stride = (i64)gridDim.x * 256;
    double s = 0;
    for (i64 q = (i64)blockIdx.x * 256 + threadIdx.x; q < n; q += stride) { const d2 x = a[q]; s += x.x + x.y; }
    if (s == 1.2345e300) *sink = s;
}
// one pass, one element per thread (the fill pass's own launch shape: many short workgroups)
__global__ __launch_bounds__(256) void calib_read8_onepass(const double *__restrict__ a, i64 n, double *sink) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q < n && a[q] == 1.2345e300) *sink = 1;
}
// non-temporal variants (the fill pass reads its six flux arrays this way): does a streaming load fetch whole 128-byte lines?
__global__ __launch_bounds__(256) void calib_read8_nt(const double *__restrict__ a, i64 n, double *sink) {
    const i64 stride = (i64)gridDim.x * 256;
    double s = 0;
    for (i64 q = (i64)blockIdx.x * 256 + threadIdx.x; q < n; q += stride) s += __builtin_nontemporal_load(a + q);
    if (s == 1.2345e300) *sink = s;
}
__global__ __launch_bounds__(256) void calib_gather8_nt(const double *__restrict__ a, const i64 *__restrict__ idx, i64 n, double *sink) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q < n && __builtin_nontemporal_load(a + idx[q]) == 1.2345e300) *sink = 1;
}
__global__ __launch_bounds__(256) void calib_gather8(const double *__restrict__ a, const i64 *__restrict__ idx, i64 n, double *sink) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q < n && a[idx[q]] == 1.2345e300) *sink = 1;
}
__global__ __launch_bounds__(256) void calib_write16(d2 *__restrict__ a, i64 n) {
    const i64 stride = (i64)gridDim.x * 256;
    for (i64 q = (i64)blockIdx.x * 256 + threadIdx.x; q < n; q += stride) a[q] = d2{1.0, 2.0};
}
// write shapes: one pass, every workgroup writes a contiguous 16 KB block (4 stores of 16 B per lane, 4 KB apart)
template <bool NT>
__global__ __launch_bounds__(256) void write16_block(d2 *__restrict__ a, i64 n) {
    const i64 q0 = (i64)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const i64 q = q0 + r * 256;
        if (q < n) {
            if (NT) __builtin_nontemporal_store(d2{1.0, 2.0}, a + q);
            else a[q] = d2{1.0, 2.0};
        }
    }
}
__global__ __launch_bounds__(256) void write8_block(double *__restrict__ a, i64 n) {
    const i64 q0 = (i64)blockIdx.x * 2048 + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const i64 q = q0 + r * 256;
        if (q < n) a[q] = 1.0;
    }
}
// ten output streams at once: workgroup b writes a 4 KB piece (one store per lane) into each of ten regions of the buffer
__global__ __launch_bounds__(256) void write16_ten_streams(d2 *__restrict__ a, i64 n) {
    const i64 per = n / 10;
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q < per) {
#pragma unroll
        for (int s = 0; s < 10; ++s) a[s * per + q] = d2{1.0, 2.0};
    }
}
// reads beside writes at a chosen ratio: a workgroup reads RD blocks of 4 KB and writes WR blocks of 4 KB
template <int RD, int WR>
__global__ __launch_bounds__(256) void rw_ratio(const d2 *__restrict__ a, d2 *__restrict__ b, i64 nblk) {
    const i64 blk = blockIdx.x;
    if (blk >= nblk) return;
    d2 acc = {0, 0};
#pragma unroll
    for (int r = 0; r < RD; ++r) acc += a[(blk * RD + r) * 256 + threadIdx.x];
#pragma unroll
    for (int w = 0; w < WR; ++w) b[(blk * WR + w) * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void calib_copy16(const d2 *__restrict__ a, d2 *__restrict__ b, i64 n) {
    const i64 stride = (i64)gridDim.x * 256;
    for (i64 q = (i64)blockIdx.x * 256 + threadIdx.x; q < n; q += stride) b[q] = a[q];
}

// the mix: a workgroup takes chunks of 4 x 256 d2 elements of every input stream (4 loads of 16 B per lane and stream =
// 40 KB... per wave 10 KB per round), then writes 2.25 x as many bytes: stream pair m gets CNT[m] 16-byte stores per
// lane and round for two rounds, plus a quarter round.  PERSIST: grid-stride over chunks; else one chunk per workgroup.
template <bool PERSIST>
__global__ __launch_bounds__(256) void mix_stream(Mix p, i64 nchunks) {
    const int CNT[5] = {7, 4, 5, 1, 3};
    for (i64 c = blockIdx.x; c < nchunks; c += PERSIST ? gridDim.x : nchunks) {
        const i64 e0 = c * 1024 + threadIdx.x;
        d2 v[NIN][4];
#pragma unroll
        for (int a = 0; a < NIN; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[a][r] = p.in[a][e0 + r * 256];
        // 20 entries of 16 bytes per lane and round; rounds 0, 1 full, round 2 a quarter of the lanes: 45 stores per 40 loads
        const i64 o0 = c * (i64)256;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (r == 2 && threadIdx.x >= 64) continue;  // a quarter round: one wave's worth, contiguous
            int e = 0;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
#pragma unroll
                for (int q = 0; q < CNT[m]; ++q) {
                    const d2 x = v[(e + r) % NIN][(e + q) & 3];
                    const i64 pos = (o0 * 3 + (i64)r * 256) * CNT[m] + (i64)q * 256 + threadIdx.x;
                    p.out[2 * m][pos] = x;
                    p.out[2 * m + 1][pos] = x;
                    ++e;
                }
            }
        }
    }
}

int main() {
    const i64 NB = 2ll << 30;  // calibration buffer: 2 GiB
    double *buf, *buf2, *sink;
    CK(hipMalloc(&buf, NB)); CK(hipMalloc(&buf2, NB)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, NB)); CK(hipMemset(buf2, 0, NB));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes, int rep, auto launch) {
        for (int r = 0; r < 3; ++r) launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < rep; ++r) launch();
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.3f ms  %7.1f GB  %6.2f TB/s\n", name, ms / rep, bytes / 1e9, bytes / (ms / rep * 1e-3) / 1e12);
        fflush(stdout);
    };
    const i64 n8 = NB / 8, n16 = NB / 16;
    timeit("calib_read8   (8 B/lane stream, 2 GiB, grid-stride)", (double)NB, 10, [&] { hipLaunchKernelGGL(calib_read8, dim3(256 * 16), dim3(256), 0, 0, buf, n8, sink); });
    timeit("calib_read16  (16 B/lane stream, 2 GiB, grid-stride)", (double)NB, 10, [&] { hipLaunchKernelGGL(calib_read16, dim3(256 * 16), dim3(256), 0, 0, (const d2 *)buf, n16, sink); });
    timeit("calib_read8_nt (8 B/lane stream, non-temporal loads)", (double)NB, 10, [&] { hipLaunchKernelGGL(calib_read8_nt, dim3(256 * 16), dim3(256), 0, 0, buf, n8, sink); });
    timeit("calib_read8_onepass (8 B/lane, one element per thread)", (double)NB, 10, [&] { hipLaunchKernelGGL(calib_read8_onepass, dim3((unsigned)(n8 / 256)), dim3(256), 0, 0, buf, n8, sink); });
    // gather through an index list with a wet mask's gaps: 54 % of the cells, runs and holes of random length
    {
        std::vector<i64> idx;
        idx.reserve(n8);
        unsigned long long s = 88172645463325252ull;
        i64 L = 0;
        while (true) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const int run = 1 + (int)(s & 15), hole = (int)((s >> 8) & 15) * 14 / 15;  // mean 8.5 wet, 7 land
            for (int q = 0; q < run && L < n8; ++q) idx.push_back(L++);
            L += hole;
            if (L >= n8) break;
        }
        const i64 ng = (i64)idx.size() / 256 * 256;
        i64 *didx;
        CK(hipMalloc(&didx, ng * 8));
        CK(hipMemcpy(didx, idx.data(), ng * 8, hipMemcpyHostToDevice));
        printf("gather: %lld of %lld cells (%.1f %%)\n", (long long)ng, (long long)n8, 100.0 * ng / n8);
        timeit("calib_gather8 (8 B/lane through an index list; bytes = whole buffer + index)", (double)NB + ng * 8.0, 10,
               [&] { hipLaunchKernelGGL(calib_gather8, dim3((unsigned)(ng / 256)), dim3(256), 0, 0, buf, didx, ng, sink); });
        timeit("calib_gather8_nt (the same gather, non-temporal loads)", (double)NB + ng * 8.0, 10,
               [&] { hipLaunchKernelGGL(calib_gather8_nt, dim3((unsigned)(ng / 256)), dim3(256), 0, 0, buf, didx, ng, sink); });
        CK(hipFree(didx));
    }
    timeit("calib_write16 (16 B/lane stream, 2 GiB)", (double)NB, 10, [&] { hipLaunchKernelGGL(calib_write16, dim3(256 * 16), dim3(256), 0, 0, (d2 *)buf, n16); });
    timeit("write16_block (one pass, 16 KB contiguous per workgroup)", (double)NB, 10, [&] { hipLaunchKernelGGL((write16_block<false>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (d2 *)buf, n16); });
    timeit("write16_block, nontemporal", (double)NB, 10, [&] { hipLaunchKernelGGL((write16_block<true>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (d2 *)buf, n16); });
    timeit("write8_block  (one pass, 8 B/lane, 16 KB per workgroup)", (double)NB, 10, [&] { hipLaunchKernelGGL(write8_block, dim3((unsigned)(n8 / 2048)), dim3(256), 0, 0, buf, n8); });
    timeit("write16_ten_streams (4 KB into each of ten regions)", (double)(n16 / 10 * 10) * 16, 10, [&] { hipLaunchKernelGGL(write16_ten_streams, dim3((unsigned)((n16 / 10 + 255) / 256)), dim3(256), 0, 0, (d2 *)buf, n16); });
    timeit("rw_ratio<4,4>  (copy, one pass)", 2.0 * (n16 / 1024 * 1024) * 16, 10, [&] { hipLaunchKernelGGL((rw_ratio<4, 4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const d2 *)buf, (d2 *)buf2, n16 / 1024); });
    timeit("rw_ratio<2,4>  (1 read : 2 written)", 1.5 * (n16 / 1024 * 1024) * 16, 10, [&] { hipLaunchKernelGGL((rw_ratio<2, 4>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const d2 *)buf, (d2 *)buf2, n16 / 1024); });
    timeit("rw_ratio<4,2>  (2 read : 1 written)", 1.5 * (n16 / 1024 * 1024) * 16, 10, [&] { hipLaunchKernelGGL((rw_ratio<4, 2>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const d2 *)buf, (d2 *)buf2, n16 / 1024); });
    timeit("rw_ratio<4,1>  (4 read : 1 written)", 1.25 * (n16 / 1024 * 1024) * 16, 10, [&] { hipLaunchKernelGGL((rw_ratio<4, 1>), dim3((unsigned)(n16 / 1024)), dim3(256), 0, 0, (const d2 *)buf, (d2 *)buf2, n16 / 1024); });
    timeit("calib_copy16  (2 GiB -> 2 GiB)", 2.0 * NB, 10, [&] { hipLaunchKernelGGL(calib_copy16, dim3(256 * 16), dim3(256), 0, 0, (const d2 *)buf, (d2 *)buf2, n16); });
    CK(hipFree(buf)); CK(hipFree(buf2));

    // the mix: 10 inputs of 400 MB, outputs 2.25 x
    Mix p{};
    const i64 n_in = (400ll << 20) / 16 / 1024 * 1024, nchunks = n_in / 1024;
    double in_bytes = 0, out_bytes = 0;
    for (int a = 0; a < NIN; ++a) { d2 *x; CK(hipMalloc(&x, n_in * 16)); CK(hipMemset(x, 0, n_in * 16)); p.in[a] = x; in_bytes += n_in * 16.0; }
    for (int m = 0; m < 5; ++m)
        for (int h = 0; h < 2; ++h) {
            const i64 ne = nchunks * 256 * 3 * CNT_H[m] + 1024;
            CK(hipMalloc(&p.out[2 * m + h], ne * 16));
            out_bytes += nchunks * 256.0 * 2.25 * CNT_H[m] * 16;
        }
    p.n_in = n_in;
    printf("mix: %.2f GB read, %.2f GB written\n", in_bytes / 1e9, out_bytes / 1e9);
    timeit("mix_stream<one chunk per workgroup>", in_bytes + out_bytes, 10, [&] { hipLaunchKernelGGL((mix_stream<false>), dim3((unsigned)nchunks), dim3(256), 0, 0, p, nchunks); });
    for (int wg : {256 * 2, 256 * 4, 256 * 8})  {
        char name[80];
        snprintf(name, sizeof name, "mix_stream<persistent, %d workgroups>", wg);
        timeit(name, in_bytes + out_bytes, 10, [&] { hipLaunchKernelGGL((mix_stream<true>), dim3(wg), dim3(256), 0, 0, p, nchunks); });
    }
    return 0;
}
