// Micro-benchmark (tools only): a MODEL of the fill pass's traffic and arithmetic, to ask one question -- can the part of the
// kernel's time that does not overlap with its memory traffic (profiles/r02/README.md: time = 0.167 ms + bytes / 6.3 TB/s)
// be made to overlap by restructuring a wave's life?
//   per wet cell (one lane): an index load, then NLOAD gathers of 8 bytes around it (same cell, +-1, +-nx, +-P in NARR 3-D
//   arrays, plus 2-D arrays), ~NFMA dependent f64 FMAs, then 320 bytes of output written as contiguous 16-byte-per-lane stores
//   into ten streams (20 entries of 16 bytes per cell: 7 + 4 + 5 + 1 + 3 like T, Tadv, TkH, TkVML, TkVdeep).
// Variants: plain (one tile per workgroup) with / without loads, arithmetic, stores; PIPE: persistent workgroups that issue
// tile t+1's loads before tile t's arithmetic and stores.
//   hipcc --offload-arch=gfx950 -O3 -o fill_model fill_model.hip && ./fill_model
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef long long i64;
typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));

#define NARR 10  // 3-D arrays: v, rho, Lwet3D read at 7 stencil points, thk at 5, the six fluxes at one neighbour each
#define N2D 10   // 2-D arrays read at the cell, two of them also at 4 neighbours
#define NVAL (7 + 7 + 7 + 5 + 6 + N2D + 8)
static const int CNT[5] = {7, 4, 5, 1, 3};

struct Params {
    const i64 *lwet;
    const double *a[NARR];
    const double *b[N2D];
    double *out[10];
    i64 n, G, P;
    int nx, nfma, shift8;
};

template <bool LOADS, bool NOEW = false>
__device__ __forceinline__ void load_cell(const Params &p, i64 w, double (&v)[NVAL]) {
    if (!LOADS) {
#pragma unroll
        for (int q = 0; q < NVAL; ++q) v[q] = 1.0 + 1e-9 * (double)(w + q);
        return;
    }
    const i64 wc = w < p.n ? w : p.n - 1;
    const i64 L = p.lwet[wc];
    const i64 s = L % p.P;
    const i64 off[7] = {0, 1, -1, p.nx, -(i64)p.nx, p.P, -p.P};
    int q = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 0; o < 7; ++o) {
            if (k == 3 && o >= 5) continue;
            if (NOEW && (o == 1 || o == 2)) { v[q++] = 1.0; continue; }
            i64 x = L + off[o];
            x = x < 0 ? L : (x >= p.G ? L : x);
            v[q++] = p.a[k][x];
        }
    }
#pragma unroll
    for (int k = 4; k < NARR; ++k) {
        i64 x = L + off[k - 3];
        x = x < 0 ? L : (x >= p.G ? L : x);
        v[q++] = p.a[k][x];
    }
#pragma unroll
    for (int k = 0; k < N2D; ++k) v[q++] = p.b[k][s];
    const i64 o2[4] = {1, -1, p.nx, -(i64)p.nx};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            i64 x = s + o2[o];
            x = x < 0 ? s : (x >= p.P ? s : x);
            v[q++] = p.b[k][x];
        }
    }
}

template <bool MATH>
__device__ __forceinline__ void math(const Params &p, const double (&v)[NVAL], double (&r)[20]) {
#pragma unroll
    for (int e = 0; e < 20; ++e) r[e] = v[e % NVAL] + v[(e + 20) % NVAL] * v[(e + 40) % NVAL];
    if (MATH) {
        for (int it = 0; it < p.nfma; ++it) {
#pragma unroll
            for (int e = 0; e < 20; ++e) r[e] = __builtin_fma(r[e], 1.0000001, v[e % 8]);
        }
    }
}

// a wave's 64 cells own 64 * CNT[m] consecutive entries of matrix m: written as 16-byte-per-lane stores (two 8-byte entries)
template <bool STORES, bool NT = false, bool ALIGNCHUNK = false>
__device__ __forceinline__ void store_tile(const Params &p, i64 w0wave, int lane, const double (&r)[20]) {
    if (!STORES) {
        double s = 0;
#pragma unroll
        for (int e = 0; e < 20; ++e) s += r[e];
        if (s == 12345.678) p.out[0][0] = s;
        return;
    }
    int e0 = 0;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const i64 base = w0wave * CNT[m] + p.shift8;  // entries (shift8: runs start off the cache-line grid)
        if (ALIGNCHUNK) {
            // chunks of 128 entries on the absolute 1 KB grid of the array: every store instruction covers whole cache lines,
            // the first and last chunk of the run are partially masked (one chunk more per run than the unaligned walk)
            const i64 first = base & ~(i64)127, end = base + 64 * CNT[m];
#pragma unroll
            for (int c = 0; c <= CNT[m]; c += 2) {
                const i64 ent = first + (i64)c * 64 + lane * 2;
                if (ent >= base && ent + 1 < end + 1) {
                    const int cc = c < CNT[m] ? c : CNT[m] - 1;
                    d2 x = {r[e0 + cc], r[e0 + (cc + 1 < CNT[m] ? cc + 1 : cc)]};
                    *(d2 *)(p.out[2 * m] + ent) = x;
                    *(d2 *)(p.out[2 * m + 1] + ent) = x;
                }
            }
            e0 += CNT[m];
            continue;
        }
#pragma unroll
        for (int c = 0; c < CNT[m]; c += 2) {
            // entries [c*64, c*64 + 128) of the wave's run, 2 per lane; an odd last column is a half store
            const i64 ent = base + (i64)c * 64 + lane * 2;
            if (c + 1 < CNT[m] || lane < 32) {
                d2 x = {r[e0 + c], r[e0 + (c + 1 < CNT[m] ? c + 1 : c)]};
                if (NT) {
                    __builtin_nontemporal_store(x, (d2 *)(p.out[2 * m] + ent));
                    __builtin_nontemporal_store(x, (d2 *)(p.out[2 * m + 1] + ent));
                } else {
                    *(d2 *)(p.out[2 * m] + ent) = x;
                    *(d2 *)(p.out[2 * m + 1] + ent) = x;
                }
            }
        }
        e0 += CNT[m];
    }
}

template <bool LOADS, bool MATH, bool STORES, bool NOEW = false, bool NT = false, bool ALIGNCHUNK = false>
__global__ __launch_bounds__(256) void plain(Params p) {
    const i64 w = (i64)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double v[NVAL], r[20];
    load_cell<LOADS, NOEW>(p, w, v);
    math<MATH>(p, v, r);
    store_tile<STORES, NT, ALIGNCHUNK>(p, w - lane, lane, r);
}

// ROWS4: the four waves of a workgroup take four chunks of 64 wet cells that lie in four ADJACENT ROWS at about the same i (chunk
// ids `stride` apart, stride = wet cells per row / 64) instead of four consecutive chunks: the south / north lines of one wave are
// the own lines of its neighbour wave -- one L1 miss instead of two.
__global__ __launch_bounds__(256) void plain_rows4(Params p, int stride, i64 nchunks) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const i64 b = blockIdx.x;
    const i64 chunk = (b / stride) * (4 * stride) + (b % stride) + (i64)wid * stride;
    if (chunk >= nchunks) return;
    const i64 w = chunk * 64 + lane;
    double v[NVAL], r[20];
    load_cell<true>(p, w, v);
    math<true>(p, v, r);
    store_tile<true>(p, w - lane, lane, r);
}

// occupancy: the same kernel with LDSB bytes of (unused) LDS per workgroup, i.e. 160 KB / LDSB workgroups of 4 waves per CU
template <int LDSB>
__global__ __launch_bounds__(256) void plain_occ(Params p) {
    __shared__ char pad[LDSB];
    const i64 w = (i64)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double v[NVAL], r[20];
    load_cell<true>(p, w, v);
    if (p.nfma < 0) pad[threadIdx.x] = (char)w;  // keep the array alive
    math<true>(p, v, r);
    if (p.nfma < 0) r[0] += pad[(threadIdx.x + 1) & 255];
    store_tile<true>(p, w - lane, lane, r);
}

// the real kernel's extras: a workgroup barrier between arithmetic and stores (block scan of the counts), and the entries staged
// through LDS (28 KB per workgroup in the real kernel) before the 16-byte stores
template <bool BARRIER, bool STAGE>
__global__ __launch_bounds__(256) void plain_extras(Params p) {
    __shared__ double stage[4][7 * 64 + 8];
    __shared__ unsigned long long tot[4];
    const i64 w = (i64)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double v[NVAL], r[20];
    load_cell<true>(p, w, v);
    math<true>(p, v, r);
    if (BARRIER) {
        unsigned long long x = (unsigned long long)(r[0] != 0.0) + 5;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { unsigned long long y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == 63) tot[wid] = x;
        __syncthreads();
        r[1] += (double)(tot[0] + tot[1] + tot[2] + tot[3] == 12345);
    }
    if (STAGE) {  // per matrix: column-major scatter into the wave's LDS area, read back as consecutive pairs
        int e0 = 0;
#pragma unroll
        for (int m = 0; m < 5; ++m) {
#pragma unroll
            for (int c = 0; c < CNT[m]; ++c) stage[wid][lane * CNT[m] + c] = r[e0 + c];
#pragma unroll
            for (int c = 0; c < CNT[m]; c += 2) {
                r[e0 + c] = stage[wid][c * 64 + lane * 2];
                if (c + 1 < CNT[m]) r[e0 + c + 1] = stage[wid][c * 64 + lane * 2 + 1];
            }
            e0 += CNT[m];
        }
    }
    store_tile<true>(p, w - lane, lane, r);
}

// persistent: the loads of the next tile are in flight while this tile is computed and stored
__global__ __launch_bounds__(256) void piped(Params p, i64 ntiles) {
    const int lane = threadIdx.x & 63;
    i64 tile = blockIdx.x;
    if (tile >= ntiles) return;
    double v[NVAL], vn[NVAL], r[20];
    load_cell<true>(p, tile * 256 + threadIdx.x, v);
    while (true) {
        const i64 next = tile + gridDim.x;
        if (next < ntiles) load_cell<true>(p, next * 256 + threadIdx.x, vn);
        math<true>(p, v, r);
        store_tile<true>(p, tile * 256 + threadIdx.x - lane, lane, r);
        if (next >= ntiles) break;
#pragma unroll
        for (int q = 0; q < NVAL; ++q) v[q] = vn[q];
        tile = next;
    }
}

// k-march model: a wave is 64 consecutive i of one row and walks the levels; the levels above / below stay in registers, east /
// west would come by DPP: per level only the own level of the ten 3-D arrays and the south / north rows of five of them are
// loaded (4 cache lines per load instead of ~8), the 2-D arrays once per column.  57 % of the lanes hold a wet cell.
__global__ __launch_bounds__(256) void march(Params p, int ny, int nz, int nseg, int kparts) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const i64 wave = (i64)blockIdx.x * 4 + wid;
    const int kp = (int)(wave % kparts);  // the levels are cut into kparts segments: more waves, shorter marches
    const i64 col = wave / kparts;
    const int j = (int)(col / nseg), seg = (int)(col % nseg);
    if (j >= ny) return;
    const int k_lo = (int)((i64)nz * kp / kparts), k_hi = (int)((i64)nz * (kp + 1) / kparts);
    const int i = seg * 64 + lane;
    const bool in = i < p.nx;
    const i64 s = (i64)j * p.nx + (in ? i : p.nx - 1);
    const i64 sS = j > 0 ? s - p.nx : s, sN = j + 1 < ny ? s + p.nx : s;
    double c2[N2D + 8];
#pragma unroll
    for (int k = 0; k < N2D; ++k) c2[k] = p.b[k][s];
#pragma unroll
    for (int k = 0; k < 4; ++k) { c2[N2D + 2 * k] = p.b[k][sS]; c2[N2D + 2 * k + 1] = p.b[k][sN]; }
    unsigned h = (unsigned)(s * 2654435761u);
    double prev[4] = {0, 0, 0, 0};
    for (int k = k_lo; k < k_hi; ++k) {
        const i64 o = (i64)k * p.P;
        double v[NARR + 10];
#pragma unroll
        for (int a = 0; a < NARR; ++a) v[a] = p.a[a][o + s];
#pragma unroll
        for (int a = 0; a < 5; ++a) { v[NARR + 2 * a] = p.a[a][o + sS]; v[NARR + 2 * a + 1] = p.a[a][o + sN]; }
        h = h * 1664525u + 1013904223u;
        const bool wet = in && ((h >> 16) % 100u) < 57u;
        const unsigned long long m = __ballot(wet);
        const int cnt = __popcll(m);
        double r[20];
#pragma unroll
        for (int e = 0; e < 20; ++e) r[e] = v[e % (NARR + 10)] + c2[e % (N2D + 8)] * prev[e % 4];
        for (int it = 0; it < p.nfma; ++it) {
#pragma unroll
            for (int e = 0; e < 20; ++e) r[e] = __builtin_fma(r[e], 1.0000001, v[e % 8]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) prev[q] = v[q];
        // the wave's run: cnt * CNT[m] entries, contiguous, 16 bytes per lane and store
        const i64 runbase = (col * nz + k) * 64;
        int e0 = 0;
#pragma unroll
        for (int mm = 0; mm < 5; ++mm) {
            const int nent = cnt * CNT[mm];
#pragma unroll
            for (int c = 0; c < CNT[mm]; c += 2) {
                const int first = c * 64 + lane * 2;
                if (first < nent) {
                    d2 x = {r[e0 + c], r[e0 + (c + 1 < CNT[mm] ? c + 1 : c)]};
                    const i64 ent = runbase * CNT[mm] + first;
                    *(d2 *)(p.out[2 * mm] + ent) = x;
                    *(d2 *)(p.out[2 * mm + 1] + ent) = x;
                }
            }
            e0 += CNT[mm];
        }
    }
}

// 3-D block model (round 4): a workgroup owns 64 consecutive i x J rows x K levels.  Phase 1: the rows of the block and of its halo
// (south / north rows, levels above / below; no corners) of the four stencil arrays, and the (J + 2) rows of the ten 2-D arrays, come in
// by coalesced row loads -- every line ONCE per block -- and are parked in LDS.  Phase 2 (after one barrier): a wave per (row, level),
// a lane per cell; the stencil is read from LDS, only the six flux arrays (read at one point per cell: nothing to share) come from
// global memory; wet lanes store a contiguous run per (row, level) like the march model.  wetpct % of the lanes hold a wet cell.
// Requests per wet cell against the gather kernel's 3.6 (1 degree) / 3.15 (0.25 degree): J = K = 4 -> 2.4 / 1.9.
template <int J, int K, int UNR>
__global__ __launch_bounds__(256) void block_lds(Params p, int ny, int nz, int nseg, int njg, int nkg, unsigned wetpct, int xcd) {
    constexpr int W = 66, NR3 = (J + 2) * (K + 2), NR2 = J + 2;
    extern __shared__ double lds[];
    double *s3 = lds;                      // [4][NR3][W]
    double *s2 = lds + 4 * NR3 * W;        // [N2D][NR2][W]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned b = blockIdx.x;
    if (xcd) {  // XCD x takes the x-th contiguous eighth of the blocks
        const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8, y = b / 8;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int kg = (int)(b % nkg), seg = (int)((b / nkg) % nseg), jg = (int)(b / nkg / nseg);
    if (jg >= njg) return;
    const int i0 = seg * 64, j0 = jg * J, k0 = kg * K;
    // ---- phase 1: row loads, UNR rows of a wave in flight at a time ----
    constexpr int NROWS = 4 * NR3 + N2D * NR2;
    for (int base = wid; base < NROWS; base += 4 * UNR) {
        double x0[UNR], x1[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = base + 4 * u;
            x0[u] = 0; x1[u] = 0;
            if (r < 4 * NR3) {
                const int a = r / NR3, q = r % NR3, jj = q % (J + 2), kk = q / (J + 2);
                const bool corner = (jj == 0 || jj == J + 1) && (kk == 0 || kk == K + 1);
                int j = j0 + jj - 1, k = k0 + kk - 1;
                j = j < 0 ? 0 : (j >= ny ? ny - 1 : j);
                k = k < 0 ? 0 : (k >= nz ? nz - 1 : k);
                const double *row = p.a[a] + ((i64)k * ny + j) * p.nx;
                int i = i0 - 1 + lane; i = i < 0 ? 0 : (i >= p.nx ? p.nx - 1 : i);
                int i2 = i0 + 63 + lane; i2 = i2 >= p.nx ? p.nx - 1 : i2;
                if (!corner) { x0[u] = row[i]; if (lane < 2) x1[u] = row[i2]; }
            } else if (r < NROWS) {
                const int q = r - 4 * NR3, a = q / NR2, jj = q % NR2;
                int j = j0 + jj - 1; j = j < 0 ? 0 : (j >= ny ? ny - 1 : j);
                const double *row = p.b[a] + (i64)j * p.nx;
                int i = i0 - 1 + lane; i = i < 0 ? 0 : (i >= p.nx ? p.nx - 1 : i);
                int i2 = i0 + 63 + lane; i2 = i2 >= p.nx ? p.nx - 1 : i2;
                x0[u] = row[i]; if (lane < 2) x1[u] = row[i2];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int r = base + 4 * u;
            if (r < NROWS) {
                double *dst = (r < 4 * NR3) ? s3 + (i64)r * W : s2 + (i64)(r - 4 * NR3) * W;
                dst[lane] = x0[u];
                if (lane < 2) dst[64 + lane] = x1[u];
            }
        }
    }
    __syncthreads();
    // ---- phase 2: a wave per (row, level) of the block ----
    for (int rl = wid; rl < J * K; rl += 4) {
        const int jj = rl % J, kk = rl / J, j = j0 + jj, k = k0 + kk, i = i0 + lane;
        if (j >= ny || k >= nz) continue;  // (wave-uniform)
        const bool in = i < p.nx;
        const i64 s = (i64)j * p.nx + (in ? i : p.nx - 1), L = (i64)k * p.P + s;
        double v[NVAL];
        int q = 0;
        const int c3 = ((kk + 1) * (J + 2) + (jj + 1)) * W + lane + 1;
        const int o3[7] = {0, 1, -1, W, -W, (J + 2) * W, -(J + 2) * W};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int o = 0; o < 7; ++o) {
                if (a == 3 && o >= 5) continue;
                v[q++] = s3[a * NR3 * W + c3 + o3[o]];
            }
        }
        const i64 off[7] = {0, 1, -1, p.nx, -(i64)p.nx, p.P, -p.P};
#pragma unroll
        for (int a = 4; a < NARR; ++a) {
            i64 x = L + off[a - 3];
            x = x < 0 ? L : (x >= p.G ? L : x);
            v[q++] = __builtin_nontemporal_load(p.a[a] + x);
        }
        const int c2 = (jj + 1) * W + lane + 1;
#pragma unroll
        for (int a = 0; a < N2D; ++a) v[q++] = s2[a * NR2 * W + c2];
        const int o2[4] = {1, -1, W, -W};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int o = 0; o < 4; ++o) v[q++] = s2[a * NR2 * W + c2 + o2[o]];
        }
        double r[20];
        math<true>(p, v, r);
        unsigned h = (unsigned)(L * 2654435761u);
        h = h * 1664525u + 1013904223u;
        const bool wet = in && ((h >> 16) % 100u) < wetpct;
        // compacted run of the wave's wet cells: position of a wet lane = number of wet lanes below it
        const unsigned long long m = __ballot(wet);
        const int cnt = __popcll(m);
        const int pos = __popcll(m & ((1ull << lane) - 1ull));
        const i64 runbase = (((i64)seg * ny + j) * nz + k) * 64;
        // (model: every lane writes its own entries at its compacted position, 16 bytes per lane and store, like store_tile)
        int e0 = 0;
#pragma unroll
        for (int mm = 0; mm < 5; ++mm) {
            const int nent = cnt * CNT[mm];
#pragma unroll
            for (int c = 0; c < CNT[mm]; c += 2) {
                const int first = c * 64 + lane * 2;
                if (first < nent) {
                    d2 x = {r[e0 + c] + pos, r[e0 + (c + 1 < CNT[mm] ? c + 1 : c)]};
                    const i64 ent = runbase * CNT[mm] + first;
                    __builtin_nontemporal_store(x, (d2 *)(p.out[2 * mm] + ent));
                    __builtin_nontemporal_store(x, (d2 *)(p.out[2 * mm + 1] + ent));
                }
            }
            e0 += CNT[mm];
        }
    }
}

// The same block, pipelined as well as a single workgroup can be: phase 1 is ONE burst of asynchronous `global_load_lds_dwordx4` row loads
// (34 lanes x 16 bytes = the 64 cells of the segment + 2 halo cells on either side, from an even index so that the 16 bytes are aligned;
// no VGPR staging, every row of the wave in flight at once), the flux loads of all the wave's (row, level)s are issued before the barrier
// too, and phase 2 touches global memory only to store.  Only two of the ten 2-D arrays carry halo rows (as in the real stencil).
template <int J, int K>
__global__ __launch_bounds__(256) void block_async(Params p, int ny, int nz, int nseg, int njg, int nkg, unsigned wetpct, int xcd) {
    constexpr int W = 68, NR3 = (J + 2) * (K + 2), NR2 = 2 * (J + 2) + (N2D - 2) * J, ITER = (J * K + 3) / 4;
    extern __shared__ double lds[];
    double *s3 = lds;                  // [4][NR3][W]
    double *s2 = lds + 4 * NR3 * W;    // arrays 0, 1: [J + 2][W] each; arrays 2 ..: [J][W] each
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned b = blockIdx.x;
    if (xcd) {
        const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8, y = b / 8;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int kg = (int)(b % nkg), seg = (int)((b / nkg) % nseg), jg = (int)(b / nkg / nseg);
    if (jg >= njg) return;
    const int i0 = seg * 64, j0 = jg * J, k0 = kg * K;
    // the 16-byte piece of a row this lane fetches: cells i0 - 2 + 2 * lane, +1 (clamped into the row; lanes >= 34 idle)
    int ia = i0 - 2 + 2 * lane;
    ia = ia < 0 ? 0 : (ia > p.nx - 2 ? p.nx - 2 : ia);
    constexpr int NROWS = 4 * NR3 + NR2;
    for (int r = wid; r < NROWS; r += 4) {  // (wave-uniform trip count and row)
        const double *row;
        double *dst;
        bool skip = false;
        if (r < 4 * NR3) {
            const int a = r / NR3, q = r % NR3, jj = q % (J + 2), kk = q / (J + 2);
            skip = (jj == 0 || jj == J + 1) && (kk == 0 || kk == K + 1);
            int j = j0 + jj - 1, k = k0 + kk - 1;
            j = j < 0 ? 0 : (j >= ny ? ny - 1 : j);
            k = k < 0 ? 0 : (k >= nz ? nz - 1 : k);
            row = p.a[a] + ((i64)k * ny + j) * p.nx;
            dst = s3 + (i64)r * W;
        } else {
            const int q = r - 4 * NR3;
            int a, jj;
            if (q < 2 * (J + 2)) { a = q / (J + 2); jj = q % (J + 2) - 1; } else { a = 2 + (q - 2 * (J + 2)) / J; jj = (q - 2 * (J + 2)) % J; }
            int j = j0 + jj; j = j < 0 ? 0 : (j >= ny ? ny - 1 : j);
            row = p.b[a] + (i64)j * p.nx;
            dst = s2 + (i64)q * W;
        }
        if (!skip && lane < 34)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(row + ia), (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    }
    // flux loads of every (row, level) of this wave
    double phi[ITER][6];
    const i64 off[7] = {0, 1, -1, p.nx, -(i64)p.nx, p.P, -p.P};
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int rl = wid + 4 * it, jj = rl % J, kk = rl / J;
        int j = j0 + jj, k = k0 + kk, i = i0 + lane;
        j = j >= ny ? ny - 1 : j; k = k >= nz ? nz - 1 : k; i = i >= p.nx ? p.nx - 1 : i;
        const i64 L = (i64)k * p.P + (i64)j * p.nx + i;
#pragma unroll
        for (int a = 4; a < NARR; ++a) {
            i64 x = L + off[a - 3];
            x = x < 0 ? L : (x >= p.G ? L : x);
            phi[it][a - 4] = __builtin_nontemporal_load(p.a[a] + x);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int rl = wid + 4 * it;
        if (rl >= J * K) break;
        const int jj = rl % J, kk = rl / J, j = j0 + jj, k = k0 + kk, i = i0 + lane;
        if (j >= ny || k >= nz) continue;
        const bool in = i < p.nx;
        const i64 L = (i64)k * p.P + (i64)j * p.nx + (in ? i : p.nx - 1);
        double v[NVAL];
        int q = 0;
        const int c3 = ((kk + 1) * (J + 2) + (jj + 1)) * W + lane + 2;
        const int o3[7] = {0, 1, -1, W, -W, (J + 2) * W, -(J + 2) * W};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int o = 0; o < 7; ++o) {
                if (a == 3 && o >= 5) continue;
                v[q++] = s3[a * NR3 * W + c3 + o3[o]];
            }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) v[q++] = phi[it][a];
        const int ch = (jj + 1) * W + lane + 2;  // arrays with halo rows
        const int cn = jj * W + lane + 2;        // arrays without
#pragma unroll
        for (int a = 0; a < N2D; ++a) v[q++] = (a < 2) ? s2[a * (J + 2) * W + ch] : s2[(2 * (J + 2) + (a - 2) * J) * W + cn];
        const int o2[4] = {1, -1, W, -W};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int o = 0; o < 4; ++o) v[q++] = s2[a * (J + 2) * W + ch + o2[o]];
        }
        double r[20];
        math<true>(p, v, r);
        unsigned h = (unsigned)(L * 2654435761u);
        h = h * 1664525u + 1013904223u;
        const bool wet = in && ((h >> 16) % 100u) < wetpct;
        const unsigned long long m = __ballot(wet);
        const int cnt = __popcll(m);
        const int pos = __popcll(m & ((1ull << lane) - 1ull));
        const i64 runbase = (((i64)seg * ny + j) * nz + k) * 64;
        int e0 = 0;
#pragma unroll
        for (int mm = 0; mm < 5; ++mm) {
            const int nent = cnt * CNT[mm];
#pragma unroll
            for (int c = 0; c < CNT[mm]; c += 2) {
                const int first = c * 64 + lane * 2;
                if (first < nent) {
                    d2 x = {r[e0 + c] + pos, r[e0 + (c + 1 < CNT[mm] ? c + 1 : c)]};
                    const i64 ent = runbase * CNT[mm] + first;
                    __builtin_nontemporal_store(x, (d2 *)(p.out[2 * mm] + ent));
                    __builtin_nontemporal_store(x, (d2 *)(p.out[2 * mm + 1] + ent));
                }
            }
            e0 += CNT[mm];
        }
    }
}

template <int J, int K>
static void run_block_async(Params p, int ny, int nz, int nseg, unsigned wetpct, hipEvent_t e0, hipEvent_t e1) {
    const int njg = (ny + J - 1) / J, nkg = (nz + K - 1) / K;
    const size_t ldsb = (size_t)(4 * (J + 2) * (K + 2) + 2 * (J + 2) + (N2D - 2) * J) * 68 * 8;
    (void)hipFuncSetAttribute((const void *)block_async<J, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    for (int xcd = 1; xcd < 2; ++xcd) {
        auto launch = [&] { hipLaunchKernelGGL((block_async<J, K>), dim3((unsigned)(nseg * njg * nkg)), dim3(256), ldsb, 0, p, ny, nz, nseg, njg, nkg, wetpct, xcd); };
        for (int r = 0; r < 10; ++r) launch();
        (void)hipDeviceSynchronize();
        hipError_t err = hipGetLastError();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 30; ++r) launch();
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("block model, asynchronous row loads: %d rows x %d levels, LDS %zu KB (%d workgroups per CU), XCD eighths, wet %u %%: %.4f ms%s\n", J, K, ldsb / 1024,
               (int)(160 * 1024 / ldsb), wetpct, ms / 30, err == hipSuccess ? "" : "  (LAUNCH ERROR)");
    }
}

template <int J, int K, int UNR>
static void run_block(const char *tag, Params p, int ny, int nz, int nseg, unsigned wetpct, hipEvent_t e0, hipEvent_t e1) {
    const int njg = (ny + J - 1) / J, nkg = (nz + K - 1) / K;
    const size_t ldsb = (size_t)(4 * (J + 2) * (K + 2) + N2D * (J + 2)) * 66 * 8;
    (void)hipFuncSetAttribute((const void *)block_lds<J, K, UNR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    for (int xcd = 0; xcd < 2; ++xcd) {
        auto launch = [&] { hipLaunchKernelGGL((block_lds<J, K, UNR>), dim3((unsigned)(nseg * njg * nkg)), dim3(256), ldsb, 0, p, ny, nz, nseg, njg, nkg, wetpct, xcd); };
        for (int r = 0; r < 10; ++r) launch();
        (void)hipDeviceSynchronize();
        hipError_t err = hipGetLastError();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 30; ++r) launch();
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("block model %s: %d rows x %d levels, %d rows in flight per wave, LDS %zu KB, %s, wet %u %%: %.4f ms%s\n", tag, J, K, UNR, ldsb / 1024,
               xcd ? "XCD eighths" : "blockIdx order", wetpct, ms / 30, err == hipSuccess ? "" : "  (LAUNCH ERROR)");
    }
}

int main(int argc, char **argv) {
    // default: the 1 degree grid; `fill_model 1440 1080 75 63526216` = the 0.25 degree grid
    const int nx = argc > 3 ? atoi(argv[1]) : 360, ny = argc > 3 ? atoi(argv[2]) : 300, nz = argc > 3 ? atoi(argv[3]) : 50;
    const int nseg = (nx + 63) / 64;
    const i64 P = (i64)nx * ny, G = P * nz, n = (argc > 4 ? atoll(argv[4]) : 3051515) / 256 * 256, ntiles = n / 256;
    const bool quick = argc > 5;  // fewer variants
    std::vector<i64> lwet(n);
    unsigned long long s = 88172645463325252ull;
    i64 L = 0;
    for (i64 w = 0; w < n; ++w) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        L += 1 + ((s & 3) == 0) + ((s & 12) == 0) * 2 + (((s >> 8) % 100) < (unsigned long long)(100.0 * ((double)G / n - 1.75) > 0 ? 100.0 * ((double)G / n - 1.75) : 0));
        if (L >= G) L = G - 1;
        lwet[w] = L;
    }
    Params p{};
    i64 *d_lwet;
    CK(hipMalloc(&d_lwet, n * 8));
    CK(hipMemcpy(d_lwet, lwet.data(), n * 8, hipMemcpyHostToDevice));
    p.lwet = d_lwet;
    for (int k = 0; k < NARR; ++k) { double *x; CK(hipMalloc(&x, G * 8)); CK(hipMemset(x, 0, G * 8)); p.a[k] = x; }
    for (int k = 0; k < N2D; ++k) { double *x; CK(hipMalloc(&x, P * 8)); CK(hipMemset(x, 0, P * 8)); p.b[k] = x; }
    double out_bytes = 0;
    for (int m = 0; m < 5; ++m)
        for (int h = 0; h < 2; ++h) { CK(hipMalloc(&p.out[2 * m + h], ((i64)nseg * ny * nz * 64 * CNT[m] + 256) * 8)); out_bytes += (double)n * CNT[m] * 8; }
    p.n = n; p.G = G; p.P = P; p.nx = nx;
    const double in_bytes = (double)G * 8 * NARR + (double)n * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 30;
    auto timeit = [&](const char *name, auto launch) {
        for (int r = 0; r < 20; ++r) launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < REP; ++r) launch();
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.4f ms\n", name, ms / REP);
        return ms / REP;
    };
    printf("model: %lld cells, reads >= %.0f MB, writes %.0f MB\n", (long long)n, in_bytes / 1e6, out_bytes / 1e6);
    for (int nf : {0}) {
        p.nfma = nf;
        printf("-- %d rounds of 20 dependent FMAs per cell --\n", nf);
        timeit("plain: loads + math + stores", [&] { hipLaunchKernelGGL((plain<true, true, true>), dim3(ntiles), dim3(256), 0, 0, p); });
        if (!quick) timeit("plain: 8 east/west loads fewer", [&] { hipLaunchKernelGGL((plain<true, true, true, true>), dim3(ntiles), dim3(256), 0, 0, p); });
        if (!quick) {
        timeit("plain: nontemporal stores", [&] { hipLaunchKernelGGL((plain<true, true, true, false, true>), dim3(ntiles), dim3(256), 0, 0, p); });
        timeit("plain, 5 workgroups per CU (32 KB LDS)", [&] { hipLaunchKernelGGL((plain_occ<32 * 1024>), dim3(ntiles), dim3(256), 0, 0, p); });
        timeit("plain, 4 workgroups per CU (40 KB LDS)", [&] { hipLaunchKernelGGL((plain_occ<40 * 1024>), dim3(ntiles), dim3(256), 0, 0, p); });
        timeit("plain, 3 workgroups per CU (52 KB LDS)", [&] { hipLaunchKernelGGL((plain_occ<52 * 1024>), dim3(ntiles), dim3(256), 0, 0, p); });
        timeit("plain, 2 workgroups per CU (64 KB LDS)", [&] { hipLaunchKernelGGL((plain_occ<64 * 1024>), dim3(ntiles), dim3(256), 0, 0, p); });
        for (int sh : {1, 2, 3, 4, 8, 9}) {
            char name[80];
            snprintf(name, sizeof name, "plain, every run shifted by %d bytes", sh * 8);
            p.shift8 = sh;
            timeit(name, [&] { hipLaunchKernelGGL((plain<true, true, true>), dim3(ntiles), dim3(256), 0, 0, p); });
            if (sh % 2 == 0) {
                snprintf(name, sizeof name, "   ... written in line-aligned chunks");
                timeit(name, [&] { hipLaunchKernelGGL((plain<true, true, true, false, false, true>), dim3(ntiles), dim3(256), 0, 0, p); });
            }
        }
        p.shift8 = 0;
        timeit("plain + barrier", [&] { hipLaunchKernelGGL((plain_extras<true, false>), dim3(ntiles), dim3(256), 0, 0, p); });
        timeit("plain + LDS staging", [&] { hipLaunchKernelGGL((plain_extras<false, true>), dim3(ntiles), dim3(256), 0, 0, p); });
        timeit("plain + barrier + LDS staging", [&] { hipLaunchKernelGGL((plain_extras<true, true>), dim3(ntiles), dim3(256), 0, 0, p); });
        }
        {
            const int stride = (int)((double)n / ((double)ny * nz) / 64.0 + 0.5);  // chunks per row
            char name[96];
            snprintf(name, sizeof name, "plain, waves of a workgroup in 4 adjacent rows (stride %d)", stride);
            timeit(name, [&] { hipLaunchKernelGGL(plain_rows4, dim3(ntiles + 4 * stride), dim3(256), 0, 0, p, stride, n / 64); });
            timeit("plain (again)", [&] { hipLaunchKernelGGL((plain<true, true, true>), dim3(ntiles), dim3(256), 0, 0, p); });
        }
        timeit("plain: no stores", [&] { hipLaunchKernelGGL((plain<true, true, false>), dim3(ntiles), dim3(256), 0, 0, p); });
        timeit("plain: no loads", [&] { hipLaunchKernelGGL((plain<false, true, true>), dim3(ntiles), dim3(256), 0, 0, p); });
        timeit("plain: math only", [&] { hipLaunchKernelGGL((plain<false, true, false>), dim3(ntiles), dim3(256), 0, 0, p); });
        for (int kparts : {1, 2, 5, 10, 25}) {
            char name[96];
            snprintf(name, sizeof name, "march model: dense rows, %d level segments (%d waves)", kparts, nseg * ny * kparts);
            timeit(name, [&] { hipLaunchKernelGGL(march, dim3((nseg * ny * kparts + 3) / 4), dim3(256), 0, 0, p, ny, nz, nseg, kparts); });
        }
        {
            const unsigned wetpct = (unsigned)(100.0 * (double)n / (double)G + 0.5);
            run_block<4, 4, 8>("a", p, ny, nz, nseg, wetpct, e0, e1);
            run_block<3, 3, 8>("b", p, ny, nz, nseg, wetpct, e0, e1);
            run_block<2, 4, 8>("c", p, ny, nz, nseg, wetpct, e0, e1);
            run_block<4, 2, 8>("d", p, ny, nz, nseg, wetpct, e0, e1);
            run_block<2, 2, 8>("e", p, ny, nz, nseg, wetpct, e0, e1);
            run_block<8, 2, 8>("f", p, ny, nz, nseg, wetpct, e0, e1);
            run_block<3, 3, 4>("g", p, ny, nz, nseg, wetpct, e0, e1);
            run_block<3, 3, 12>("h", p, ny, nz, nseg, wetpct, e0, e1);
            run_block_async<3, 3>(p, ny, nz, nseg, wetpct, e0, e1);
            run_block_async<2, 2>(p, ny, nz, nseg, wetpct, e0, e1);
            run_block_async<2, 3>(p, ny, nz, nseg, wetpct, e0, e1);
            run_block_async<3, 2>(p, ny, nz, nseg, wetpct, e0, e1);
            run_block_async<4, 2>(p, ny, nz, nseg, wetpct, e0, e1);
            run_block_async<4, 4>(p, ny, nz, nseg, wetpct, e0, e1);
            timeit("plain: nontemporal stores (again)", [&] { hipLaunchKernelGGL((plain<true, true, true, false, true>), dim3(ntiles), dim3(256), 0, 0, p); });
        }
        for (int wgs : {768})  {
            char name[64];
            snprintf(name, sizeof name, "piped, %d persistent workgroups", wgs);
            timeit(name, [&] { hipLaunchKernelGGL(piped, dim3(wgs), dim3(256), 0, 0, p, ntiles); });
        }
    }
    return 0;
}
