// otmb_facefluxes.hip -- facefluxesfrommasstransport / facefluxes / nofluxboundaries! on the
// device (src/velocities.jl:118-130, :190-255, :154-179), fused into one pass:
// one thread per (i,j) water column marches k = nz..1 (the continuity recurrence :236-243 is a
// sequential chain in k with a fixed association, :242), reading umo/vmo once (plus the west and
// south neighbours' values, which are L1/L2 hits of the same rows) and writing the six ϕ arrays
// once.  Lanes run along i, so every access is coalesced.
#include "otmb_common.h"

#define FF_THREADS 64
#ifndef FF_KB
#define FF_KB 6  // levels per chunk (two chunks are live: the one being processed and the one in flight)
#endif

// replace(x, NaN => 0.0, FillValue => 0.0) -- isequal semantics (:203, :215)
__device__ __forceinline__ double ff_replace(double x, double fill) {
    return (isnan(x) || __double_as_longlong(x) == __double_as_longlong(fill)) ? 0.0 : x;
}

// The inputs of FF_KB consecutive levels of one water column (registers; every index is a compile-time constant)
template <typename T>
struct FfChunk {
    T u[FF_KB], v[FF_KB], uw[FF_KB], vs[FF_KB];
    unsigned char wc[FF_KB], wE[FF_KB], wW[FF_KB], wS[FF_KB], wN[FF_KB];  // (FLAGS: wc holds the cell's five flags, the others are unused)
};
// wet flags of a cell and its four horizontal neighbours (otmb_wetflags_dev): the boundary rule of nofluxboundaries!
// (:161-175) needs five wet bytes per cell and level -- grid constants -- which this byte replaces by one load
enum { WF_C = 1, WF_E = 2, WF_W = 4, WF_S = 8, WF_N = 16 };

// lane offsets are 32-bit BYTE offsets (8*nx*ny < 2^31, checked on the host) from level bases that are uniform:
// scalar base + vector offset addressing, no 64-bit address arithmetic per access
template <typename T> __device__ __forceinline__ T ff_ld(const T *base, unsigned idx) {
    return *(const T *)((const char *)base + idx * (unsigned)sizeof(T));
}
// NT: non-temporal stores.  The six ϕ arrays are read back by transportmatrix: on a 1 degree grid (259 MB) partly out of the Infinity
// Cache, which ordinary stores favour; on grids whose fluxes are far larger than any cache they only displace what the kernel itself
// re-reads (the west / south neighbours' rows): 1.63 -> 1.46 ms at 0.25 degree, no change at 1 degree (chosen by size on the host).
template <bool NT, typename T> __device__ __forceinline__ void ff_st(T *base, unsigned idx, T x) {
    if (NT) __builtin_nontemporal_store(x, (T *)((char *)base + idx * (unsigned)sizeof(T)));
    else *(T *)((char *)base + idx * (unsigned)sizeof(T)) = x;
}
struct FfCol {
    unsigned s, sE, sW, cS, cN;  // this column and its (clamped) neighbour columns inside a level
    bool hS, hN;
};

// load_vs (wave-uniform): false when the wave's south row comes from the neighbouring wave through LDS (ff_south_from_lds)
template <typename T, bool FLAGS>
__device__ __forceinline__ void ff_load(FfChunk<T> &c, const T *__restrict__ umo, const T *__restrict__ vmo,
                                        const uint8_t *__restrict__ wet, const FfCol &col, i64 P, int k0, bool load_vs = true) {
#pragma unroll
    for (int q = 0; q < FF_KB; ++q) {
        const int k = (k0 - q >= 0) ? k0 - q : 0;  // clamped: loads are unconditional, the level is skipped later
        const T *ul = umo + (i64)k * P, *vl = vmo + (i64)k * P;
        const uint8_t *wl = wet + (i64)k * P;
        c.u[q] = ff_ld(ul, col.s); c.v[q] = ff_ld(vl, col.s);
        c.uw[q] = ff_ld(ul, col.sW);
        if (load_vs) c.vs[q] = ff_ld(vl, col.cS);
        c.wc[q] = ff_ld(wl, col.s);
        if (!FLAGS) {
            c.wE[q] = ff_ld(wl, col.sE); c.wW[q] = ff_ld(wl, col.sW); c.wS[q] = ff_ld(wl, col.cS);
            c.wN[q] = ff_ld(wl, col.cN);
        }
    }
}

template <typename T, bool FLAGS, bool NT>
__device__ __forceinline__ void ff_levels(const FfChunk<T> &c, const FfCol &col, i64 P, int k0, double fill, double &topbelow,
                                          bool &uvalid, bool &vvalid, double *__restrict__ east, double *__restrict__ west,
                                          double *__restrict__ north, double *__restrict__ south, double *__restrict__ top,
                                          double *__restrict__ bottom, uint16_t *__restrict__ push_mask, bool active = true) {
#pragma unroll
    for (int q = 0; q < FF_KB; ++q) {
        const int k = k0 - q;
        if (k >= 0 && active) {  // (active: lanes beyond the row / rows beyond the grid of a four-row workgroup march along for the barriers only)
            const i64 o = (i64)k * P;
            const unsigned f = c.wc[q];
            const bool wc = FLAGS ? (f & WF_C) != 0 : c.wc[q] != 0, wE = FLAGS ? (f & WF_E) != 0 : c.wE[q] != 0,
                       wW = FLAGS ? (f & WF_W) != 0 : c.wW[q] != 0;
            const bool wS = FLAGS ? (f & WF_S) != 0 : (col.hS && c.wS[q] != 0), wN = FLAGS ? (f & WF_N) != 0 : (col.hN && c.wN[q] != 0);
            double u = (double)c.u[q], v = (double)c.v[q];  // Array{Float64}(umo), :125-126
            // nofluxboundaries!, :167-173
            if (!wc || !wE) u = 0.0;
            if (!wc || !wN) v = 0.0;
            uvalid |= !(isnan(u) || u == fill);  // :199
            vvalid |= !(isnan(v) || v == fill);  // :200
            const double e = ff_replace(u, fill), n = ff_replace(v, fill);
            // ϕwest[c] = ϕeast[i₋₁(c)] (:206-211): the west cell's east flux after ITS boundary rule
            double uw = (double)c.uw[q];
            if (!wW || !wc) uw = 0.0;
            const double w = ff_replace(uw, fill);
            // ϕsouth[c] = ϕnorth[j₋₁(c)], 0 at j == 1 (:219-224); j₊₁ of the south cell is c
            double vs = (double)c.vs[q];
            if (!wS || !wc) vs = 0.0;
            const double so = col.hS ? ff_replace(vs, fill) : 0.0;
            const double b = topbelow;                  // :238-240
            const double t = (((b + w) + so) - e) - n;  // :242
            ff_st<NT>(east + o, col.s, e); ff_st<NT>(west + o, col.s, w); ff_st<NT>(north + o, col.s, n); ff_st<NT>(south + o, col.s, so);
            ff_st<NT>(top + o, col.s, t); ff_st<NT>(bottom + o, col.s, b);
            if (push_mask) ff_st<false>(push_mask + o, col.s, (uint16_t)otmb_push_bits(w, e, so, n, b, t, wc));
            topbelow = t;
        }
    }
}

// Software pipeline over chunks of FF_KB levels: while chunk A is turned into fluxes and stored, the loads of the
// next chunk B are already in flight.  A column is one thread and the grid has few columns (1.7 waves per SIMD at
// 1 degree), so nothing else hides the memory latency of a chunk.
// ROWS: 1 = a workgroup is one wave, 64 consecutive columns of the (nx, ny) plane in linear order; 4 = a workgroup is four waves
// over the SAME 64 values of i in four consecutive rows j: a wave's south row (vmo[s - nx]) is then the row of the wave next to it
// in the same workgroup, marching through the same levels at the same time -- an L1 hit (or a ride on the fill already in flight)
// instead of one more request to the L2 (4 of the ~14 read requests per level and wave).  Large grids only: the last chunk of a row
// is partly idle (nx = 1440: 2 % more waves), and a grid with fewer waves than the chip has slots wants them spread singly.
template <typename T, bool FLAGS, bool NT, int ROWS>
__global__ __launch_bounds__(FF_THREADS * ROWS) void facefluxes_kernel(
    const T *__restrict__ umo, const T *__restrict__ vmo, const uint8_t *__restrict__ wet, double fill, int nx,
    int ny, int nz, int topo, i64 P, double *__restrict__ east, double *__restrict__ west,
    double *__restrict__ north, double *__restrict__ south, double *__restrict__ top, double *__restrict__ bottom,
    const double *__restrict__ top_below, uint16_t *__restrict__ push_mask, int *uv, int gen, int xcd_chunks, int lds_south) {
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  In blockIdx order a wave's south row
    // (vmo[s - nx], nx / 64 blocks back) and the west cell of its first lane (the previous block) belong to workgroups of OTHER XCDs:
    // every XCD's L2 then fetches vmo twice and a quarter of umo again (profiles/r03: 3.27 GB fetched for 1.98 GB of inputs at
    // 0.25 degree).  Give XCD x the x-th contiguous eighth of the column blocks instead, as the fill pass does: those neighbours
    // are then lines the same L2 has just fetched.  Speed only: any bijection of the blocks is correct.
    unsigned cb = blockIdx.x;
    if (xcd_chunks) {
        const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = blockIdx.x % 8, y = blockIdx.x / 8;
        cb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    unsigned s, i, j;
    bool inside;
    if (ROWS == 1) {
        s = cb * FF_THREADS + threadIdx.x;
        j = s / (unsigned)nx;
        i = s - j * (unsigned)nx;
        inside = s < (unsigned)P;
    } else {
        const unsigned nchunk = ((unsigned)nx + FF_THREADS - 1) / FF_THREADS, grp = cb / nchunk, chunk = cb - grp * nchunk;
        i = chunk * FF_THREADS + (threadIdx.x & (FF_THREADS - 1));
        j = grp * ROWS + threadIdx.x / FF_THREADS;
        inside = i < (unsigned)nx && j < (unsigned)ny;
        s = j * (unsigned)nx + i;
    }
    bool uvalid = false, vvalid = false;
    if (ROWS > 1 && lds_south) {
        // Four-row workgroups, south row through LDS: wave r's south row IS wave r-1's own row, so every wave posts the vmo values of
        // its chunk in LDS and takes its south values from the wave below it -- no load at all for three waves out of four (the L1 kept
        // missing on them: the waves drift apart by more than it holds; profiles/r04/README.md section 4).  Every wave takes part in
        // the workgroup barriers, so lanes beyond the row and rows beyond the grid march along on clamped addresses and store nothing.
        __shared__ T s_v[2][ROWS][FF_KB][FF_THREADS];
        const unsigned lane = threadIdx.x & (FF_THREADS - 1), r = threadIdx.x / FF_THREADS;
        const unsigned ic = (i < (unsigned)nx) ? i : (unsigned)nx - 1, jc = (j < (unsigned)ny) ? j : (unsigned)ny - 1;
        const unsigned sc = jc * (unsigned)nx + ic, row = jc * (unsigned)nx;
        FfCol col;
        col.s = sc;
        col.sE = row + ((ic + 1 < (unsigned)nx) ? ic + 1 : 0);
        col.sW = row + ((ic > 0) ? ic - 1 : nx - 1);
        col.hS = jc > 0;
        const bool fold = (jc + 1 >= (unsigned)ny) && (topo == OTMB_TRIPOLAR);
        col.hN = (jc + 1 < (unsigned)ny) || fold;
        col.cS = col.hS ? sc - nx : sc;
        col.cN = (jc + 1 < (unsigned)ny) ? sc + nx : (fold ? row + (nx - 1 - ic) : sc);
        double topbelow = top_below ? top_below[sc] : 0.0;
        const bool own_vs = r == 0;  // (wave-uniform) the first row of the workgroup has its south row in another workgroup
        FfChunk<T> A, B;
        int k0 = nz - 1, buf = 0;
        auto south_from_lds = [&](FfChunk<T> &c) {
#pragma unroll
            for (int q = 0; q < FF_KB; ++q) s_v[buf][r][q][lane] = c.v[q];
            __syncthreads();
            if (!own_vs) {
#pragma unroll
                for (int q = 0; q < FF_KB; ++q) c.vs[q] = s_v[buf][r - 1][q][lane];
            }
            buf ^= 1;  // (the next chunk writes the other half: one barrier per chunk is enough, see the ordering argument in DESIGN.md 3.2)
        };
        ff_load<T, FLAGS>(A, umo, vmo, wet, col, P, k0, own_vs);
        while (k0 >= 0) {
            ff_load<T, FLAGS>(B, umo, vmo, wet, col, P, k0 - FF_KB, own_vs);
            south_from_lds(A);
            ff_levels<T, FLAGS, NT>(A, col, P, k0, fill, topbelow, uvalid, vvalid, east, west, north, south, top, bottom, push_mask, inside);
            k0 -= FF_KB;
            if (k0 < 0) break;
            ff_load<T, FLAGS>(A, umo, vmo, wet, col, P, k0 - FF_KB, own_vs);
            south_from_lds(B);
            ff_levels<T, FLAGS, NT>(B, col, P, k0, fill, topbelow, uvalid, vvalid, east, west, north, south, top, bottom, push_mask, inside);
            k0 -= FF_KB;
        }
    } else if (inside) {
        const unsigned row = j * (unsigned)nx;
        FfCol col;
        col.s = s;
        col.sE = row + ((i + 1 < (unsigned)nx) ? i + 1 : 0);  // i₊₁, gridtopology.jl:57
        col.sW = row + ((i > 0) ? i - 1 : nx - 1);            // i₋₁, :58
        col.hS = j > 0;                                        // j₋₁, :63
        const bool fold = (j + 1 >= (unsigned)ny) && (topo == OTMB_TRIPOLAR);
        col.hN = (j + 1 < (unsigned)ny) || fold;               // j₊₁, :62; tripolar fold :94
        col.cS = col.hS ? s - nx : s;                          // clamped neighbour columns: loads stay unconditional
        col.cN = (j + 1 < (unsigned)ny) ? s + nx : (fold ? row + (nx - 1 - i) : s);
        // seafloor ϕbottom is zero (:238); for a depth slab that is not the deepest, the plane handed up
        // by the slab below (its ϕtop at its first level) continues the chain without re-association
        double topbelow = top_below ? top_below[s] : 0.0;
        FfChunk<T> A, B;
        int k0 = nz - 1;
        ff_load<T, FLAGS>(A, umo, vmo, wet, col, P, k0);
        while (k0 >= 0) {
            ff_load<T, FLAGS>(B, umo, vmo, wet, col, P, k0 - FF_KB);
            ff_levels<T, FLAGS, NT>(A, col, P, k0, fill, topbelow, uvalid, vvalid, east, west, north, south, top, bottom, push_mask);
            k0 -= FF_KB;
            if (k0 < 0) break;
            ff_load<T, FLAGS>(A, umo, vmo, wet, col, P, k0 - FF_KB);
            ff_levels<T, FLAGS, NT>(B, col, P, k0, fill, topbelow, uvalid, vvalid, east, west, north, south, top, bottom, push_mask);
            k0 -= FF_KB;
        }
    }
    // "some value is valid" = the call's pair of words (ring slot gen % OTMB_RING) holds this call's number: no reset
    // pass between calls, and the verdict on call g survives the next OTMB_RING - 1 calls
    if (__any(uvalid) && (threadIdx.x & 63) == 0 && uv[0] != gen) atomicExch(&uv[0], gen);
    if (__any(vvalid) && (threadIdx.x & 63) == 0 && uv[1] != gen) atomicExch(&uv[1], gen);
}

// one thread per cell: the cell's wet byte and those of its east / west / south / north neighbours (periodic in i, closed in
// j, tripolar fold at j == ny: src/gridtopology.jl:57-65,94) folded into one byte
__global__ __launch_bounds__(256) void wetflags_kernel(const uint8_t *__restrict__ wet, int nx, int ny, i64 P, i64 G, int topo,
                                                       uint8_t *__restrict__ flags) {
    const i64 L = (i64)blockIdx.x * 256 + threadIdx.x;
    if (L >= G) return;
    const i64 k = L / P, s = L - k * P;
    const int j = (int)(s / nx), i = (int)(s - (i64)j * nx);
    const i64 row = L - i;
    const bool fold = (j + 1 >= ny) && (topo == OTMB_TRIPOLAR);
    unsigned f = wet[L] ? WF_C : 0u;
    if (wet[row + ((i + 1 < nx) ? i + 1 : 0)]) f |= WF_E;
    if (wet[row + ((i > 0) ? i - 1 : nx - 1)]) f |= WF_W;
    if (j > 0 && wet[L - nx]) f |= WF_S;
    if ((j + 1 < ny) ? wet[L + nx] != 0 : (fold && wet[row + (nx - 1 - i)] != 0)) f |= WF_N;
    flags[L] = (uint8_t)f;
}

static int32_t facefluxes_impl(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                               const uint8_t *wet3d, double fill, int64_t nx, int64_t ny, int64_t nz,
                               int32_t topology, double *const phi[6], const double *top_below, uint16_t *push_mask,
                               bool check_missing, bool flags = false) {
    if (!ctx || !umo || !vmo || !wet3d || !phi) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    for (int f = 0; f < 6; ++f)
        if (!phi[f]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
    if (nx < 1 || ny < 1 || nz < 1 || nx * ny >= (1ll << 28)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    if (topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);  // :163 -> gridtopology.jl:111
    if (topology != OTMB_BIPOLAR && topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "topology");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 P = nx * ny;
    if (ctx->ff_gen == 0x7fffffff) { ctx->ff_gen = 0; ctx->ff_first = 1; }
    ctx->ff_gen += 1;
    int *dflags = otmb_ring_ff((int *)ctx->ring.p, ctx->ff_gen);
    // four-row workgroups on grids with many more waves than the chip has slots (otmb_ctx: ff_rows = 0 lets the size decide)
    const int rows = ctx->ff_rows > 0 ? (ctx->ff_rows >= 4 ? 4 : 1) : ((P >= (1ll << 19) && nx >= 2 * FF_THREADS) ? 4 : 1);
    const unsigned nb = rows == 1 ? (unsigned)((P + FF_THREADS - 1) / FF_THREADS)
                                  : (unsigned)(((nx + FF_THREADS - 1) / FF_THREADS) * ((ny + rows - 1) / rows));
    {
    KernelTimer kt(ctx, K_FACEFLUXES);
    const bool nt = (i64)48 * P * nz > (1ll << 30);  // six Float64 arrays beyond a gigabyte: streaming stores
#define FF_LAUNCH(T, FL, NTS, R)                                                                                                       \
    hipLaunchKernelGGL((facefluxes_kernel<T, FL, NTS, R>), dim3(nb), dim3(FF_THREADS * R), 0, ctx->stream, (const T *)umo, (const T *)vmo, wet3d, \
                       fill, (int)nx, (int)ny, (int)nz, (int)topology, P, phi[OTMB_EAST], phi[OTMB_WEST], phi[OTMB_NORTH],           \
                       phi[OTMB_SOUTH], phi[OTMB_TOP], phi[OTMB_BOTTOM], top_below, push_mask, dflags, ctx->ff_gen, ctx->ff_xcd_chunks, ctx->ff_lds_south)
#define FF_LAUNCH2(T, FL) do { if (rows == 4) { if (nt) FF_LAUNCH(T, FL, true, 4); else FF_LAUNCH(T, FL, false, 4); } \
                               else { if (nt) FF_LAUNCH(T, FL, true, 1); else FF_LAUNCH(T, FL, false, 1); } } while (0)
    if (src_is_f32) { if (flags) FF_LAUNCH2(float, true); else FF_LAUNCH2(float, false); }
    else { if (flags) FF_LAUNCH2(double, true); else FF_LAUNCH2(double, false); }
#undef FF_LAUNCH2
#undef FF_LAUNCH
    }
    HIP_TRY(ctx, hipGetLastError());
    // @assert !all(missing) (:199-200): needs the whole pass, so it is reported after the kernel
    if (!check_missing) return OTMB_OK;  // slab / asynchronous: otmb_facefluxes_slab_flags fetches the flags when asked
    int32_t u = 0, v = 0, rc;
    if ((rc = otmb_facefluxes_slab_flags(ctx, &u, &v))) return rc;
    if (!u || !v) return otmb_fail(ctx, OTMB_ERR_ALL_MISSING);
    return OTMB_OK;
}

extern "C" int32_t otmb_facefluxes_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                                       const uint8_t *wet3d, double fill, int64_t nx, int64_t ny, int64_t nz,
                                       int32_t topology, double *const phi[6]) {
    return facefluxes_impl(ctx, umo, vmo, src_is_f32, wet3d, fill, nx, ny, nz, topology, phi, nullptr, nullptr, true);
}

// Depth-slab variant: the levels handed in are levels [k0,k1) of a deeper grid.  top_below (nx*ny, may
// be NULL for the deepest slab) is ϕtop of level k1 computed by the slab below.  Asynchronous: the
// "all values missing" assertion (:199-200) concerns the whole grid, so the two validity flags are
// returned through otmb_facefluxes_slab_flags after a synchronize and combined by the caller.
extern "C" int32_t otmb_facefluxes_slab_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                                            const uint8_t *wet3d, double fill, int64_t nx, int64_t ny, int64_t nz,
                                            int32_t topology, double *const phi[6], const double *top_below,
                                            uint16_t *push_mask) {
    return facefluxes_impl(ctx, umo, vmo, src_is_f32, wet3d, fill, nx, ny, nz, topology, phi, top_below, push_mask, false);
}

extern "C" int32_t otmb_wetflags_dev(otmb_ctx *ctx, const uint8_t *wet3d, int64_t nx, int64_t ny, int64_t nz, int32_t topology,
                                     uint8_t *wetflags) {
    if (!ctx || !wet3d || !wetflags) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    if (topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);
    if (topology != OTMB_BIPOLAR && topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "topology");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 P = nx * ny, G = P * nz;
    hipLaunchKernelGGL(wetflags_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, ctx->stream, wet3d, (int)nx, (int)ny, P, G,
                       (int)topology, wetflags);
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}

extern "C" int32_t otmb_facefluxes_flags_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                                             const uint8_t *wetflags, double fill, int64_t nx, int64_t ny, int64_t nz,
                                             int32_t topology, double *const phi[6], const double *top_below,
                                             uint16_t *push_mask) {
    return facefluxes_impl(ctx, umo, vmo, src_is_f32, wetflags, fill, nx, ny, nz, topology, phi, top_below, push_mask, false, true);
}

// Push mask of existing ϕ arrays (include/otmb.h): one thread per cell of [first, first + count).
__global__ __launch_bounds__(256) void push_mask_kernel(const double *__restrict__ pe, const double *__restrict__ pw,
                                                        const double *__restrict__ pn, const double *__restrict__ ps,
                                                        const double *__restrict__ pt, const double *__restrict__ pb,
                                                        const i64 *__restrict__ lw, i64 first, i64 count,
                                                        uint16_t *__restrict__ mask) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= count) return;
    const i64 L = first + q;
    mask[L] = (uint16_t)otmb_push_bits(pw[L], pe[L], ps[L], pn[L], pb[L], pt[L], lw[L] != 0);
}

int32_t otmb_launch_push_mask(otmb_ctx *ctx, const double *const phi[6], const int64_t *lwet3d, int64_t first, int64_t count,
                              uint16_t *push_mask) {
    if (count <= 0) return OTMB_OK;
    KernelTimer kt(ctx, K_PUSHMASK);
    hipLaunchKernelGGL(push_mask_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, phi[OTMB_EAST],
                       phi[OTMB_WEST], phi[OTMB_NORTH], phi[OTMB_SOUTH], phi[OTMB_TOP], phi[OTMB_BOTTOM], (const i64 *)lwet3d,
                       (i64)first, (i64)count, push_mask);
    return OTMB_OK;
}

extern "C" int32_t otmb_push_mask_dev(otmb_ctx *ctx, const double *const phi[6], const int64_t *lwet3d, int64_t first,
                                      int64_t count, uint16_t *push_mask) {
    if (!ctx || !phi || !lwet3d || !push_mask) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    for (int f = 0; f < 6; ++f)
        if (!phi[f]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "phi");
    if (first < 0 || count < 0 || count >= (1ll << 39)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int32_t rc = otmb_launch_push_mask(ctx, phi, lwet3d, first, count, push_mask);
    if (rc) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}

static int32_t fetch_ff_ring(otmb_ctx *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(otmb_ring_ff(ctx->h_ring, 0), otmb_ring_ff((int *)ctx->ring.p, 0), OTMB_RING * 2 * sizeof(int),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OTMB_OK;
}

extern "C" int32_t otmb_facefluxes_slab_flags(otmb_ctx *ctx, int32_t *u_valid, int32_t *v_valid) {
    if (!ctx || !u_valid || !v_valid) return OTMB_ERR_INVALID_ARG;
    int32_t rc;
    if ((rc = fetch_ff_ring(ctx))) return rc;
    const int *uv = otmb_ring_ff(ctx->h_ring, ctx->ff_gen);
    *u_valid = ctx->ff_gen > 0 && uv[0] == ctx->ff_gen;
    *v_valid = ctx->ff_gen > 0 && uv[1] == ctx->ff_gen;
    return OTMB_OK;
}

// Validity flags of EVERY facefluxes call since the previous call of this function, oldest first (at most the
// OTMB_RING most recent ones: older verdicts have been overwritten, callers drain the pipeline before that).
extern "C" int32_t otmb_facefluxes_pending_flags(otmb_ctx *ctx, int32_t capacity, int32_t *u_valid, int32_t *v_valid,
                                                 int32_t *n_calls) {
    if (!ctx || !u_valid || !v_valid || !n_calls || capacity < 0) return OTMB_ERR_INVALID_ARG;
    int32_t rc;
    if ((rc = fetch_ff_ring(ctx))) return rc;
    int first = ctx->ff_first;
    if (ctx->ff_gen - first + 1 > OTMB_RING) first = ctx->ff_gen - OTMB_RING + 1;
    int n = 0;
    for (int g = first; g <= ctx->ff_gen && n < capacity; ++g, ++n) {
        const int *uv = otmb_ring_ff(ctx->h_ring, g);
        u_valid[n] = uv[0] == g;
        v_valid[n] = uv[1] == g;
    }
    *n_calls = n;
    ctx->ff_first = first + n;
    return OTMB_OK;
}
