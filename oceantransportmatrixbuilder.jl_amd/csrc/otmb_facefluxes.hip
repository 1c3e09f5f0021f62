// otmb_facefluxes.hip -- facefluxesfrommasstransport / facefluxes / nofluxboundaries! on the
// device (src/velocities.jl:118-130, :190-255, :154-179), fused into one pass:
// one thread per (i,j) water column marches k = nz..1 (the continuity recurrence :236-243 is a
// sequential chain in k with a fixed association, :242), reading umo/vmo once (plus the west and
// south neighbours' values, which are L1/L2 hits of the same rows) and writing the six ϕ arrays
// once.  Lanes run along i, so every access is coalesced.
#include "otmb_common.h"

#define FF_THREADS 64
#ifndef FF_KB
// levels per chunk (two chunks are live: the one being processed and the one in flight).  Round 6: 6 -> 3.  The counting kernels needed 184-202 VGPRs
// with six levels (two waves per SIMD) and 119-128 with three (four waves); the march is a chain of dependent instructions, so the waves next to it are
// what fills the issue slots: the fused step's facefluxes 0.897 -> 0.64 ms at 0.25 degree, everything else unchanged (profiles/r06/call12_ffkb_*.jsonl).
#define FF_KB 3
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
    unsigned char wup;  // (COUNTS) the flags of the level above the chunk's last level
    T vn[FF_KB];        // (COUNTS, waves with lanes on the tripolar seam row) vmo of the fold cell
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

// ---- counts in facefluxes (otmb_facefluxes_counts_dev) ----------------------------------------------------------------------------
// Which rows the four operators hold in column c (src/matrixbuilding.jl:244-296, :348-415, :450-477) is a function of the wet mask,
// the mixed-layer mask and the sign of the six fluxes the NEIGHBOURS push towards c -- and for fluxes that this kernel writes those are
// c's OWN six fluxes: ϕwest[E] = ϕeast[c], ϕeast[W] = ϕwest[c], ϕnorth[S] = ϕsouth[c], ϕsouth[N] = ϕnorth[c] (src/velocities.jl:206-224),
// ϕbottom[A] = ϕtop[c], ϕtop[B] = ϕbottom[c] (:238-240).  Of the five per-tile row counts that transportmatrix's counting pass derives
// from the push mask (tm_count_kernel), three -- TκH, TκVdeep and T's reserved union -- depend on the WET MASK alone (a flux is non-zero
// only towards a wet cell, :167-173, so the union is "every wet neighbour + the diagonal"): they are summed once per grid
// (ff_static_counts_kernel).  The other two -- Tadv (flux signs) and TκVML (zt[k] < mlotst, :85) -- are accumulated HERE, from registers:
// a wave's 64 cells of one level are consecutive in linear order, hence consecutive in wet rank -- rank = bases[level][wave segment] +
// (wet lanes before) -- and fall into at most two tiles of 256 columns: one wave sum (DPP adds) of a packed word and at most two packed
// 64-bit atomic adds per wave and level (integers: the sums do not depend on the order).  The fill pass still compares every tile's
// counts with what it builds (FLAG_COUNT_MISMATCH): counts that do not describe the fluxes cannot corrupt anything.
// (A first version took 19 ballots and 38 scalar popcounts per level and all five counts: the CU's one scalar unit then bounded the
// kernel, 1.38 -> 1.91 ms at 0.25 degree; profiles/r05.)
// The tripolar seam row (j = ny - 1): the north neighbour is the fold cell (nx - 1 - i, ny - 1), which pushes with ITS north flux
// (:271-278 seen from the other side) -- the one flux a cell does not own: waves with lanes on that row prefetch vmo of the fold cell
// with the chunk -- and lies in the cell's own matrix row block, where it can coincide with the east / west row-mate or the cell itself
// (the slot logic of build_column / general_presence, reduced to "the N bit lands on the E or W bit, or nowhere").
// Bits of a neighbour set: E 2, W 4, S 8, N 16 (as in the five-flag byte); A 32, B 64.
struct FfCountArgs {
    const uint32_t *bases;     // [0] header, then [nseg][nz]: 0-based wet rank of the first wet cell of the segment at each level (otmb_count_tables_dev)
    const double *mlotst, *zt;
    unsigned long long *sums;  // packed per-tile counts (T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10): this kernel adds Tadv and TκVML
    int upwind, only_t;
    int nseg;                  // wave segments per level
    int halo;                  // depth slab: bit 0 -- a level above the first one exists (wet flags at level -1, zt[-1]), bit 1 -- one below the last
};
#define FFC_MAX_NZ 128  // a wave keeps its per-level table entries and zt in registers, lane l = level l and level 64 + l
struct FfCountState {
    double mld;
    unsigned seg;
    unsigned base_lo, base_hi;  // lane l: 0-based wet rank of the first wet cell of this wave's segment at level l / 64 + l
    double zt_lo, zt_hi;        // lane l: zt[l - ha] / zt[64 + l - ha] (ha = 1 when a halo level lies above: its depth is needed too)
    bool ha, hb;                // (uniform) a halo level above the first / below the last level: depth slab of a deeper grid
    bool om_cur, om_below;      // zt[k] < mlotst for the level being counted / the level below it (a window that moves up with the march)
    unsigned amask;            // seam row: flag bit (E / W) of the row-mate the fold neighbour coincides with, else 0
    bool aclear;               // seam row: the fold neighbour is the east / west row-mate or the cell itself: it has no slot of its own
    bool fold;                 // the lane's cell is on the tripolar seam row
    bool has_fold;             // (wave-uniform) so is some lane of this wave
    bool act, wet_below;
};
#define FFC_HEADER(rows, nseg) (((unsigned)(rows) << 28) ^ (unsigned)(nseg))
__device__ __forceinline__ unsigned ff_mbcnt(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ bool ff_nonzero(double x) { return (x > 0.0) | (x < 0.0); }
// the seam row's slot rule on a neighbour set: the fold neighbour's bit (N) moves onto the row-mate it coincides with, or vanishes
// into the diagonal
__device__ __forceinline__ unsigned ff_fold_alias(unsigned bits, unsigned amask) { return (bits & 16u) ? ((bits | amask) & ~16u) : bits; }
// Sum of one 32-bit value per lane over the wave (every lane active): four row_shr adds put each row's total in its last lane,
// row_bcast:15 / :31 carry them to lane 63.
__device__ __forceinline__ unsigned ff_wave_sum(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// a per-level value kept one level per lane (k is wave-uniform: a v_readlane, no memory access in the march)
__device__ __forceinline__ unsigned ff_level_u32(unsigned lo, unsigned hi, int k) {
    return (unsigned)((k < 64) ? __builtin_amdgcn_readlane((int)lo, k) : __builtin_amdgcn_readlane((int)hi, k - 64));
}
__device__ __forceinline__ double ff_level_f64(double lo, double hi, int k) {
    const double v = (k < 64) ? lo : hi;
    const int l = k & 63;
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// One level of one wave (all 64 lanes execute).  wc / f: the cell's wet flag and five-flag byte, fa: the flag byte of the cell above,
// vn: vmo of the fold cell (seam-row lanes of waves with has_fold).
__device__ __forceinline__ void ff_count_level(const FfCountArgs &cnt, FfCountState &cs, int k, int nz, bool wc, unsigned f, unsigned fa,
                                               double e, double w, double so, double n, double b, double t, double vn, double fill) {
    const bool wet = cs.act & wc;
    const u64 wetb = __builtin_amdgcn_ballot_w64(wet);
    if (wetb != 0) {  // (uniform)
        const bool hA = k > 0 || cs.ha, hB = k + 1 < nz || cs.hb;
        const unsigned vw = ((hA && (fa & WF_C)) ? 32u : 0u) | ((hB && cs.wet_below) ? 64u : 0u);  // wet cells above / below
        // the flux the north neighbour pushes with, signed so that "towards c" is positive: ϕsouth[N] = ϕnorth[c]; through the seam -ϕnorth[fold]
        double nn = n;
        if (cs.has_fold) {  // (uniform)
            const double nf = (f & WF_N) ? ff_replace(vn, fill) : 0.0;  // the fold cell's ϕnorth after ITS boundary rule (its north neighbour is c)
            nn = cs.fold ? -nf : n;
        }
        unsigned ab;   // which neighbours push mass into c: max(ϕ,0) / min(ϕ,0) / ϕ/2 non-zero (:244-289), exactly otmb_push_bits
        unsigned own;  // c's own vertical pushes (bottom 64, top 32): they must land in a wet cell (the reference indexes Lwet3D[C𝑗]
                       // unconditionally); the horizontal fluxes are zero towards land by construction (:167-173)
        if (cnt.upwind) {
            ab = (e > 0.0 ? 2u : 0u) | (w < 0.0 ? 4u : 0u) | (so < 0.0 ? 8u : 0u) | (nn > 0.0 ? 16u : 0u) | (t > 0.0 ? 32u : 0u) | (b < 0.0 ? 64u : 0u);
            own = (b > 0.0 ? 64u : 0u) | ((hA && t < 0.0) ? 32u : 0u);
        } else {
            const double he = e / 2, hw = w / 2, hs = so / 2, hn = nn / 2, hb = b / 2, ht = t / 2;
            const unsigned vb = ff_nonzero(hb) ? 64u : 0u, vt = ff_nonzero(ht) ? 32u : 0u;
            ab = (ff_nonzero(he) ? 2u : 0u) | (ff_nonzero(hw) ? 4u : 0u) | (ff_nonzero(hs) ? 8u : 0u) | (ff_nonzero(hn) ? 16u : 0u) | vt | vb;
            own = vb | (hA ? vt : 0u);
        }
        const bool bad = (own & ~vw) != 0;
        ab &= vw | 0x1eu;  // the cells above / below push only if they are wet
        const unsigned anyA = ab != 0;  // the diagonal's bit
        if (cs.has_fold && cs.aclear) ab = ff_fold_alias(ab, cs.amask);
        unsigned x = wet ? __popc(ab) + anyA : 0u;  // Tadv rows of this column
        const unsigned base = ff_level_u32(cs.base_lo, cs.base_hi, k);
        const bool in0 = ff_mbcnt(wetb) < 256u - (base & 255u);  // this lane's column still belongs to the wave's first tile
        const unsigned sx = ff_wave_sum(in0 ? x : (x << 16));    // both tiles' sums in one word (<= 7 * 64 each)
        u64 c0 = (u64)(sx & 0xffffu) << 11, c1 = (u64)(sx >> 16) << 11;
        // TκVML: only within the mixed layer -- Ω[c] = zt[k] < mlotst (:85; NaN compares false), and the same for the cells above / below
        const bool omC = wet & cs.om_cur;
        if (__builtin_amdgcn_ballot_w64(omC) != 0) {  // (uniform)
            const bool om_above = ff_level_f64(cs.zt_lo, cs.zt_hi, (hA ? k - 1 : k) + (int)cs.ha) < cs.mld;
            const unsigned ml = ((om_above ? 32u : 0u) | (cs.om_below ? 64u : 0u)) & vw;
            const unsigned y = omC ? __popc(ml) + (ml != 0) : 0u;
            const unsigned sy = ff_wave_sum(in0 ? y : (y << 16));
            c0 |= (u64)(sy & 0xffffu) << 33;
            c1 |= (u64)(sy >> 16) << 33;
        }
        if (cnt.only_t) c0 = c1 = 0;
        const bool anybad = __builtin_amdgcn_ballot_w64(wet & bad) != 0;
        if ((threadIdx.x & 63) == 0) {  // one lane: the sums are wave-uniform
            const unsigned t0 = base >> FFC_TILE_SHIFT;
            if (c0) __hip_atomic_fetch_add(cnt.sums + t0, c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c1) __hip_atomic_fetch_add(cnt.sums + t0 + 1, c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (anybad) __hip_atomic_fetch_or(cnt.sums + t0, FFC_BAD_FLUX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    cs.wet_below = wc;
    // the mixed-layer window moves up one level
    cs.om_below = cs.om_cur;
    cs.om_cur = ff_level_f64(cs.zt_lo, cs.zt_hi, k > 0 ? k - 1 + (int)cs.ha : 0) < cs.mld;
}

// Before the march: the column's mixed-layer depth, the window at the deepest level and -- depth slab with a halo level below -- that
// level's wet flag and its place in the mixed layer.
__device__ __forceinline__ void ff_count_begin(const FfCountArgs &cnt, FfCountState &cs, const uint8_t *__restrict__ wet, unsigned s, i64 P, int nz) {
    cs.mld = cnt.mlotst[s];
    cs.om_cur = ff_level_f64(cs.zt_lo, cs.zt_hi, nz - 1 + (int)cs.ha) < cs.mld;
    if (cs.hb) {  // (uniform)
        cs.wet_below = (wet[(i64)nz * P + s] & WF_C) != 0;
        cs.om_below = ff_level_f64(cs.zt_lo, cs.zt_hi, nz + (int)cs.ha) < cs.mld;
    }
}

// load_vs (wave-uniform): false when the wave's south row comes from the neighbouring wave through LDS (ff_south_from_lds)
template <typename T, bool FLAGS, bool COUNTS = false>
__device__ __forceinline__ void ff_load(FfChunk<T> &c, const T *__restrict__ umo, const T *__restrict__ vmo,
                                        const uint8_t *__restrict__ wet, const FfCol &col, i64 P, int k0, bool load_vs = true,
                                        bool load_vn = false, int kup_min = 0) {
#pragma unroll
    for (int q = 0; q < FF_KB; ++q) {
        const int k = (k0 - q >= 0) ? k0 - q : 0;  // clamped: loads are unconditional, the level is skipped later
        const T *ul = umo + (i64)k * P, *vl = vmo + (i64)k * P;
        const uint8_t *wl = wet + (i64)k * P;
        c.u[q] = ff_ld(ul, col.s); c.v[q] = ff_ld(vl, col.s);
        c.uw[q] = ff_ld(ul, col.sW);
        if (load_vs) c.vs[q] = ff_ld(vl, col.cS);
        c.wc[q] = ff_ld(wl, col.s);
        if (COUNTS && load_vn) c.vn[q] = ff_ld(vl, col.cN);
        if (!FLAGS) {
            c.wE[q] = ff_ld(wl, col.sE); c.wW[q] = ff_ld(wl, col.sW); c.wS[q] = ff_ld(wl, col.cS);
            c.wN[q] = ff_ld(wl, col.cN);
        }
    }
    if (COUNTS) {  // the wet flag above the chunk's last level (the next chunk's first level: not waited for here)
        const int ku = (k0 - FF_KB >= 0) ? k0 - FF_KB : kup_min;  // (kup_min = -1: the halo level above a depth slab)
        c.wup = ff_ld(wet + (i64)ku * P, col.s);
    }
}

template <typename T, bool FLAGS, bool NT, bool COUNTS = false, bool TOPONLY = false>
__device__ __forceinline__ void ff_levels(const FfChunk<T> &c, const FfCol &col, i64 P, int k0, double fill, double &topbelow,
                                          bool &uvalid, bool &vvalid, double *__restrict__ east, double *__restrict__ west,
                                          double *__restrict__ north, double *__restrict__ south, double *__restrict__ top,
                                          double *__restrict__ bottom, uint16_t *__restrict__ push_mask, bool active,
                                          const FfCountArgs &cnt, FfCountState &cs, int nz) {
    static_assert(!COUNTS || FLAGS, "the counts read the five-flag byte");
#pragma unroll
    for (int q = 0; q < FF_KB; ++q) {
        const int k = k0 - q;
        if (k >= 0) {
            // (every lane computes: lanes beyond the row / rows beyond the grid of a four-row workgroup, and lanes beyond the plane of a counting
            // kernel, march along on clamped -- valid -- addresses; only their STORES and their validity flags are switched off.  A branch around
            // the arithmetic cost a save / restore of the execution mask and six zero-initialised doubles per level.)
            const unsigned f = c.wc[q];
            const i64 o = (i64)k * P;
            const bool wc = FLAGS ? (f & WF_C) != 0 : c.wc[q] != 0;
            const bool wE = FLAGS ? (f & WF_E) != 0 : c.wE[q] != 0, wW = FLAGS ? (f & WF_W) != 0 : c.wW[q] != 0;
            const bool wS = FLAGS ? (f & WF_S) != 0 : (col.hS && c.wS[q] != 0), wN = FLAGS ? (f & WF_N) != 0 : (col.hN && c.wN[q] != 0);
            double u = (double)c.u[q], v = (double)c.v[q];  // Array{Float64}(umo), :125-126
            // nofluxboundaries!, :167-173
            if (!wc || !wE) u = 0.0;
            if (!wc || !wN) v = 0.0;
            uvalid |= active & !(isnan(u) || u == fill);  // :199
            vvalid |= active & !(isnan(v) || v == fill);  // :200
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
            if (active) {
                if (!TOPONLY) {
                    ff_st<NT>(east + o, col.s, e); ff_st<NT>(west + o, col.s, w); ff_st<NT>(north + o, col.s, n); ff_st<NT>(south + o, col.s, so);
                    ff_st<NT>(bottom + o, col.s, b);
                }
                ff_st<NT>(top + o, col.s, t);  // (TOPONLY -- the fused step, otmb_step_dev: the other five are re-derived by the fill pass from umo / vmo / ϕtop)
                if (push_mask && !COUNTS) ff_st<false>(push_mask + o, col.s, (uint16_t)otmb_push_bits(w, e, so, n, b, t, wc));
            }
            topbelow = t;
            if (COUNTS) {
                // the cell above (level k - 1; at k == 0 it is the halo level above a depth slab -- c.wup -- or unused)
                const unsigned fa = (q + 1 < FF_KB && k > 0) ? c.wc[q + 1] : c.wup;
                ff_count_level(cnt, cs, k, nz, wc, f, fa, e, w, so, n, b, t, cs.has_fold ? (double)c.vn[q] : 0.0, fill);
            }
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
#ifndef FF_COUNTS_WAVES
#define FF_COUNTS_WAVES 1
#endif
template <typename T, bool FLAGS, bool NT, int ROWS, bool COUNTS, bool TOPONLY = false>
__global__ __launch_bounds__(FF_THREADS * ROWS) __attribute__((amdgpu_waves_per_eu(COUNTS ? FF_COUNTS_WAVES : 1))) void facefluxes_kernel(
    const T *__restrict__ umo, const T *__restrict__ vmo, const uint8_t *__restrict__ wet, double fill, int nx,
    int ny, int nz, int topo, i64 P, double *__restrict__ east, double *__restrict__ west,
    double *__restrict__ north, double *__restrict__ south, double *__restrict__ top, double *__restrict__ bottom,
    const double *__restrict__ top_below, uint16_t *__restrict__ push_mask, int *uv, int gen, int xcd_chunks, int lds_south,
    int j_begin, int j_end,  // rows [j_begin, j_end) of the plane are computed (the whole plane: 0, ny; a piece: the chain over depth slabs, piece by piece)
    // (COUNTS) FfCountArgs, member by member: only a `const __restrict__` kernel argument lets the compiler read the wave-uniform table
    // entries and zt through the scalar cache -- as a vector load the table entry made every level wait for ALL outstanding vector memory
    // operations (s_waitcnt vmcnt(0)), i.e. for the next chunk's prefetch
    const uint32_t *__restrict__ c_bases, const double *__restrict__ c_mlotst, const double *__restrict__ c_zt, unsigned long long *c_sums,
    int c_upwind, int c_only_t, int c_nseg, int c_halo) {
    FfCountArgs cnt;
    cnt.bases = c_bases; cnt.mlotst = c_mlotst; cnt.zt = c_zt; cnt.sums = c_sums; cnt.upwind = c_upwind; cnt.only_t = c_only_t; cnt.nseg = c_nseg;
    cnt.halo = c_halo;
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
    unsigned s, i, j, seg;
    bool inside;
    if (ROWS == 1) {
        seg = cb;  // (COUNTS: whole planes only, j_begin == 0)
        s = (unsigned)j_begin * (unsigned)nx + cb * FF_THREADS + threadIdx.x;
        inside = s < (unsigned)j_end * (unsigned)nx;
        if (COUNTS && !inside) s = (unsigned)j_end * (unsigned)nx - 1;
        j = s / (unsigned)nx;
        i = s - j * (unsigned)nx;
    } else {
        const unsigned nchunk = ((unsigned)nx + FF_THREADS - 1) / FF_THREADS, grp = cb / nchunk, chunk = cb - grp * nchunk;
        i = chunk * FF_THREADS + (threadIdx.x & (FF_THREADS - 1));
        j = (unsigned)j_begin + grp * ROWS + threadIdx.x / FF_THREADS;
        inside = i < (unsigned)nx && j < (unsigned)j_end;
        s = j * (unsigned)nx + i;
        seg = ((j < (unsigned)ny) ? j : (unsigned)ny - 1) * nchunk + chunk;
    }
    bool uvalid = false, vvalid = false;
    const int kup_min = (COUNTS && (c_halo & 1)) ? -1 : 0;
    FfCountState cs;
    cs.mld = 0.0; cs.seg = 0; cs.base_lo = cs.base_hi = 0; cs.zt_lo = cs.zt_hi = 0.0; cs.om_cur = cs.om_below = false; cs.amask = 0; cs.aclear = false; cs.fold = false; cs.has_fold = false; cs.act = false; cs.wet_below = false;
    cs.ha = cs.hb = false;
    if (COUNTS) {
        cs.seg = (unsigned)__builtin_amdgcn_readfirstlane((int)seg);
        cs.act = inside;
        cs.fold = inside && topo == OTMB_TRIPOLAR && j + 1 == (unsigned)ny;
        cs.has_fold = __builtin_amdgcn_ballot_w64(cs.fold) != 0;
        const unsigned ifd = (unsigned)nx - 1 - i, ie = (i + 1 < (unsigned)nx) ? i + 1 : 0, iw = (i > 0) ? i - 1 : (unsigned)nx - 1;
        cs.amask = cs.fold ? ((ifd == ie) ? (unsigned)WF_E : ((ifd == iw) ? (unsigned)WF_W : 0u)) : 0u;
        cs.aclear = cs.fold && (ifd == ie || ifd == iw || ifd == i);
        {   // this wave's table entries and zt, one level per lane (nz <= FFC_MAX_NZ, checked on the host)
            const unsigned lane = threadIdx.x & 63, l0 = (lane < (unsigned)nz) ? lane : (unsigned)nz - 1, l1 = (64 + lane < (unsigned)nz) ? 64 + lane : (unsigned)nz - 1;
            const uint32_t *mine = cnt.bases + 1 + (size_t)cs.seg * (size_t)nz;
            cs.base_lo = mine[l0]; cs.base_hi = mine[l1];
            // zt of levels -ha ... nz - 1 + hb (a depth slab's halo levels take part in the mixed-layer test of their neighbours, :85)
            cs.ha = (cnt.halo & 1) != 0; cs.hb = (cnt.halo & 2) != 0;
            const int zmax = nz - 1 + (int)cs.hb, z0 = (int)lane - (int)cs.ha, z1 = 64 + (int)lane - (int)cs.ha;
            cs.zt_lo = cnt.zt[z0 < zmax ? z0 : zmax]; cs.zt_hi = cnt.zt[z1 < zmax ? z1 : zmax];
        }
        if (cnt.bases[0] != FFC_HEADER(ROWS, cnt.nseg)) {  // a table for another wave geometry: count nothing, and say so
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_or(cnt.sums, FFC_BAD_TABLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cs.act = false;
        }
    }
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
        if (COUNTS) ff_count_begin(cnt, cs, wet, sc, P, nz);
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
        ff_load<T, FLAGS, COUNTS>(A, umo, vmo, wet, col, P, k0, own_vs, cs.has_fold, kup_min);
        while (k0 >= 0) {
            ff_load<T, FLAGS, COUNTS>(B, umo, vmo, wet, col, P, k0 - FF_KB, own_vs, cs.has_fold, kup_min);
            south_from_lds(A);
            ff_levels<T, FLAGS, NT, COUNTS, TOPONLY>(A, col, P, k0, fill, topbelow, uvalid, vvalid, east, west, north, south, top, bottom, push_mask, inside, cnt, cs, nz);
            k0 -= FF_KB;
            if (k0 < 0) break;
            ff_load<T, FLAGS, COUNTS>(A, umo, vmo, wet, col, P, k0 - FF_KB, own_vs, cs.has_fold, kup_min);
            south_from_lds(B);
            ff_levels<T, FLAGS, NT, COUNTS, TOPONLY>(B, col, P, k0, fill, topbelow, uvalid, vvalid, east, west, north, south, top, bottom, push_mask, inside, cnt, cs, nz);
            k0 -= FF_KB;
        }
    } else if (inside || COUNTS) {  // (COUNTS: every lane takes part in the wave sums; lanes beyond the plane march along on the last column and store nothing)
        if (COUNTS && ROWS > 1 && !inside) {  // (ROWS == 1 clamped s above) lanes beyond the row or the plane: clamped addresses, as in the LDS branch
            i = (i < (unsigned)nx) ? i : (unsigned)nx - 1;
            j = (j < (unsigned)ny) ? j : (unsigned)ny - 1;
            s = j * (unsigned)nx + i;
        }
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
        if (COUNTS) ff_count_begin(cnt, cs, wet, s, P, nz);
        FfChunk<T> A, B;
        int k0 = nz - 1;
        ff_load<T, FLAGS, COUNTS>(A, umo, vmo, wet, col, P, k0, true, cs.has_fold, kup_min);
        while (k0 >= 0) {
            ff_load<T, FLAGS, COUNTS>(B, umo, vmo, wet, col, P, k0 - FF_KB, true, cs.has_fold, kup_min);
            ff_levels<T, FLAGS, NT, COUNTS, TOPONLY>(A, col, P, k0, fill, topbelow, uvalid, vvalid, east, west, north, south, top, bottom, push_mask, inside, cnt, cs, nz);
            k0 -= FF_KB;
            if (k0 < 0) break;
            ff_load<T, FLAGS, COUNTS>(A, umo, vmo, wet, col, P, k0 - FF_KB, true, cs.has_fold, kup_min);
            ff_levels<T, FLAGS, NT, COUNTS, TOPONLY>(B, col, P, k0, fill, topbelow, uvalid, vvalid, east, west, north, south, top, bottom, push_mask, inside, cnt, cs, nz);
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

// four-row workgroups on grids with many more waves than the chip has slots (otmb_ctx: ff_rows = 0 lets the size decide)
static int ff_rows_for(const otmb_ctx *ctx, i64 nx, i64 ny) {
    const i64 P = nx * ny;
    return ctx->ff_rows > 0 ? (ctx->ff_rows >= 4 ? 4 : 1) : ((P >= (1ll << 19) && nx >= 2 * FF_THREADS) ? 4 : 1);
}
// wave segments per level: what one wave of facefluxes_kernel covers at one level (64 consecutive cells of the plane, or of one row)
static i64 ff_nseg(int rows, i64 nx, i64 ny) {
    return rows == 1 ? (nx * ny + FF_THREADS - 1) / FF_THREADS : ((nx + FF_THREADS - 1) / FF_THREADS) * ny;
}

// one wave per (level, segment): the 0-based wet rank of the segment's first wet cell (Lwet3D is the wet rank, src/matrixbuilding.jl:19-20)
// (depth slab: lw points at the slab's first owned level, rank_base is the global rank of its first owned wet cell)
__global__ __launch_bounds__(256) void ff_count_bases_kernel(const i64 *__restrict__ lw, int nx, int ny, int nz, i64 P, int rows, int nseg,
                                                             i64 rank_base, uint32_t *__restrict__ bases) {
    const i64 gw = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    if (gw == 0 && lane == 0) bases[0] = FFC_HEADER(rows, nseg);
    if (gw >= (i64)nz * nseg) return;
    const i64 k = gw / nseg, seg = gw - k * nseg;
    i64 s;
    bool valid;
    if (rows == 1) {
        s = seg * FF_THREADS + lane;
        valid = s < P;
    } else {
        const i64 nchunk = (nx + FF_THREADS - 1) / FF_THREADS, j = seg / nchunk, chunk = seg - j * nchunk, i = chunk * FF_THREADS + lane;
        valid = i < nx;
        s = j * nx + i;
    }
    const i64 v = valid ? lw[k * P + s] : 0;
    const u64 b = __builtin_amdgcn_ballot_w64(v != 0);
    const int first = b ? __builtin_ctzll(b) : 0;
    const i64 r = __shfl(v, first);
    if (lane == 0) bases[1 + seg * nz + k] = b ? (uint32_t)(r - 1 - rank_base) : 0u;  // [segment][level]: a wave of facefluxes reads its nz entries in one go
}

// The counts that depend on the wet mask alone, per tile of 256 consecutive wet cells (one workgroup per tile, one thread per column):
// TκH holds the wet horizontal neighbours + the diagonal (:348-415), TκVdeep the wet cells above / below + the diagonal (:450-477), and
// T's reserved union every wet neighbour + the diagonal (Tadv and TκVML rows are among them: a flux is non-zero only towards a wet cell).
// Packed as the block scan packs them; the Tadv and TκVML fields stay zero (facefluxes_kernel<COUNTS> adds those per time slice).
__global__ __launch_bounds__(256) void ff_static_counts_kernel(const i64 *__restrict__ lwet, const uint8_t *__restrict__ flags, i64 n_wet,
                                                               int nx, int ny, int nz, i64 P, int topo, unsigned long long *__restrict__ stat) {
    __shared__ unsigned long long part[4];
    const i64 w = (i64)blockIdx.x * 256 + threadIdx.x;
    unsigned long long mine = 0;
    if (w < n_wet) {
        const i64 L = lwet[w] - 1;
        const i64 k = L / P, s = L - k * P;
        const int j = (int)(s / nx), i = (int)(s - (i64)j * nx);
        unsigned hb = flags[L] & 0x1eu;
        const unsigned vw = ((k > 0 && (flags[L - P] & WF_C)) ? 32u : 0u) | ((k + 1 < nz && (flags[L + P] & WF_C)) ? 64u : 0u);
        const unsigned anyH = hb != 0, anyU = (hb | vw) != 0;
        if (topo == OTMB_TRIPOLAR && j == ny - 1) {  // the fold neighbour's slot (see ff_fold_alias)
            const int ifd = nx - 1 - i, ie = (i + 1 < nx) ? i + 1 : 0, iw = (i > 0) ? i - 1 : nx - 1;
            if (ifd == ie || ifd == iw || ifd == i) hb = ff_fold_alias(hb, (ifd == ie) ? (unsigned)WF_E : ((ifd == iw) ? (unsigned)WF_W : 0u));
        }
        const unsigned h = __popc(hb), d = __popc(vw);
        mine = (unsigned long long)(h + d + anyU) | ((unsigned long long)(h + anyH) << 22) | ((unsigned long long)(d + (vw != 0)) << 43);
    }
#pragma unroll
    for (int q = 32; q >= 1; q >>= 1) mine += __shfl_xor(mine, q);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) stat[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// layout of the caller's table: [header | nz * nseg bases] uint32, padded to 8 bytes, then one packed uint64 per tile
static size_t ff_tables_static_offset(int rows, i64 nx, i64 ny, i64 nz) {
    return ((size_t)(1 + nz * ff_nseg(rows, nx, ny)) * sizeof(uint32_t) + 7) / 8 * 8;
}

static int32_t facefluxes_impl(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                               const uint8_t *wet3d, double fill, int64_t nx, int64_t ny, int64_t nz,
                               int32_t topology, double *const phi[6], const double *top_below, uint16_t *push_mask,
                               bool check_missing, bool flags = false, const otmb_ff_counts *counts = nullptr, int64_t j0 = 0, int64_t j1 = -1,
                               bool same_call = false, bool top_only = false, const otmb_ff_slab *slab = nullptr) {
    // slab (with counts): the nz levels are levels [k_own0, k_own0 + nz) of an extended local grid of nz_ext levels; wet3d (the flag bytes),
    // counts->zt and counts->lwet3d are arrays of THAT grid, everything else holds the owned levels only
    // j0, j1: rows [j0, j1) of the plane only (j1 < 0: all of them); same_call: a further piece of the facefluxes call that the
    // previous piece began -- the validity flags of the pieces accumulate in ONE pair of words
    if (!ctx || !umo || !vmo || !wet3d || !phi) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    for (int f = 0; f < 6; ++f)
        if (!phi[f] && !(top_only && f != OTMB_TOP)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
    if (nx < 1 || ny < 1 || nz < 1 || nx * ny >= (1ll << 28)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    if (topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);  // :163 -> gridtopology.jl:111
    if (topology != OTMB_BIPOLAR && topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "topology");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 P = nx * ny;
    if (j1 < 0) j1 = ny;
    if (j0 < 0 || j0 >= j1 || j1 > ny) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "row range");
    if (!same_call || ctx->ff_gen == 0) {
        if (ctx->ff_gen == 0x7fffffff) { ctx->ff_gen = 0; ctx->ff_first = 1; }
        ctx->ff_gen += 1;
    }
    int *dflags = otmb_ring_ff((int *)ctx->ring.p, ctx->ff_gen);
    const int rows = ff_rows_for(ctx, nx, ny);
    const unsigned nb = rows == 1 ? (unsigned)(((j1 - j0) * nx + FF_THREADS - 1) / FF_THREADS)
                                  : (unsigned)(((nx + FF_THREADS - 1) / FF_THREADS) * ((j1 - j0 + rows - 1) / rows));
    // any facefluxes call on this context ends the validity of counts an earlier call left behind (they are keyed to ff_gen), and a full
    // mask written over a partial one makes that array an ordinary push mask again
    const bool later_piece = same_call && counts && ctx->ffc.pieces_open;  // a further row band of a counting call: same buffer, same key
    if (!later_piece) ctx->ffc.valid = false;
    ctx->ffc.pieces_open = false;
    if (push_mask && push_mask == ctx->ffc_partial_mask && !counts) ctx->ffc_partial_mask = nullptr;
    FfCountArgs ca;
    memset(&ca, 0, sizeof ca);
    const bool with_counts = counts != nullptr;
    const i64 k_own0 = slab ? slab->k_own0 : 0, nz_ext = slab ? slab->nz_ext : nz;
    if (with_counts) {
        // (the caller -- otmb_facefluxes_counts_dev / _slab_counts_dev -- has checked that this grid can be counted here)
        const i64 ntiles = (counts->n_wet + (1ll << FFC_TILE_SHIFT) - 1) >> FFC_TILE_SHIFT;
        const size_t need = (size_t)(ntiles + 2) * sizeof(unsigned long long);
        int buf;
        if (later_piece) {
            buf = ctx->ffc.buf;
        } else {
            buf = ctx->ffc_next;
            ctx->ffc_next ^= 1;
            if (!ctx->ffc_sums[buf].p || ctx->ffc_sums[buf].cap < need) {
                int32_t rc;
                if ((rc = otmb_reserve(ctx, ctx->ffc_sums[buf], need))) return rc;
                ctx->ffc_dirty[buf] = true;
            }
            if (ctx->ffc_dirty[buf]) HIP_TRY(ctx, hipMemsetAsync(ctx->ffc_sums[buf].p, 0, need, ctx->stream));
            ctx->ffc_dirty[buf] = true;  // until a scan has consumed (and zeroed) it
        }
        ca.bases = (const uint32_t *)counts->tables; ca.mlotst = counts->mlotst; ca.zt = counts->zt + k_own0;
        ca.sums = (unsigned long long *)ctx->ffc_sums[buf].p;
        ca.upwind = counts->upwind; ca.only_t = counts->only_t;
        ca.nseg = (int)ff_nseg(rows, nx, ny);
        ca.halo = (k_own0 > 0 ? 1 : 0) | (k_own0 + nz < nz_ext ? 2 : 0);
        otmb_ctx::FfCountsKey &key = ctx->ffc;
        key.buf = buf; key.gen = ctx->ff_gen;
        for (int f = 0; f < 6; ++f) key.phi[f] = phi[f];
        key.stat = (const char *)counts->tables + ff_tables_static_offset(rows, nx, ny, nz);
        key.mask = push_mask; key.mlotst = counts->mlotst; key.zt = counts->zt; key.lwet3d = counts->lwet3d;
        key.nx = nx; key.ny = ny; key.nz = nz_ext; key.n_wet = counts->n_wet;
        key.k_own0 = k_own0; key.wet_base = slab ? slab->wet_base : 0;
        key.topo = topology; key.upwind = counts->upwind != 0; key.only_t = counts->only_t != 0;
        ctx->ffc_partial_mask = push_mask ? push_mask - k_own0 * P : nullptr;  // (as the transportmatrix will name it: the extended grid's array)
        wet3d += k_own0 * P;  // the kernel marches over the owned levels; a halo level is level -1 / nz to it
    }
    {
    KernelTimer kt(ctx, K_FACEFLUXES);
    const bool nt = ctx->ff_nt >= 0 ? ctx->ff_nt != 0 : (i64)48 * P * nz > (1ll << 30);  // six Float64 arrays beyond a gigabyte: streaming stores
#define FF_LAUNCH(T, FL, NTS, R, CN)                                                                                                   \
    FF_LAUNCH_T(T, FL, NTS, R, CN, false)
#define FF_LAUNCH_T(T, FL, NTS, R, CN, TO)                                                                                             \
    hipLaunchKernelGGL((facefluxes_kernel<T, FL, NTS, R, CN, TO>), dim3(nb), dim3(FF_THREADS * R), 0, ctx->stream, (const T *)umo, (const T *)vmo, wet3d, \
                       fill, (int)nx, (int)ny, (int)nz, (int)topology, P, phi[OTMB_EAST], phi[OTMB_WEST], phi[OTMB_NORTH],           \
                       phi[OTMB_SOUTH], phi[OTMB_TOP], phi[OTMB_BOTTOM], top_below, push_mask, dflags, ctx->ff_gen, ctx->ff_xcd_chunks, ctx->ff_lds_south, \
                       (int)j0, (int)j1, ca.bases, ca.mlotst, ca.zt, ca.sums, ca.upwind, ca.only_t, ca.nseg, ca.halo)
#define FF_LAUNCH2(T, FL, CN) do { if (rows == 4) { if (nt) FF_LAUNCH(T, FL, true, 4, CN); else FF_LAUNCH(T, FL, false, 4, CN); } \
                                   else { if (nt) FF_LAUNCH(T, FL, true, 1, CN); else FF_LAUNCH(T, FL, false, 1, CN); } } while (0)
#define FF_LAUNCH2T(T) do { if (rows == 4) { if (nt) FF_LAUNCH_T(T, true, true, 4, true, true); else FF_LAUNCH_T(T, true, false, 4, true, true); } \
                            else { if (nt) FF_LAUNCH_T(T, true, true, 1, true, true); else FF_LAUNCH_T(T, true, false, 1, true, true); } } while (0)
    if (with_counts && top_only) { if (src_is_f32) FF_LAUNCH2T(float); else FF_LAUNCH2T(double); }
    else if (with_counts) { if (src_is_f32) FF_LAUNCH2(float, true, true); else FF_LAUNCH2(double, true, true); }
    else if (src_is_f32) { if (flags) FF_LAUNCH2(float, true, false); else FF_LAUNCH2(float, false, false); }
    else { if (flags) FF_LAUNCH2(double, true, false); else FF_LAUNCH2(double, false, false); }
#undef FF_LAUNCH2T
#undef FF_LAUNCH2
#undef FF_LAUNCH_T
#undef FF_LAUNCH
    }
    if (with_counts) { ctx->ffc.valid = true; ctx->ffc.pieces_open = true; }  // (valid for the caller once the last row band is enqueued)
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

// One row band of the plane: rows [j0, j1) of every level (0-based).  The chain over depth slabs runs piece by piece (slab s
// piece c waits for slab s + 1 piece c only): a cell's fluxes depend on its own column's top_below and on INPUTS of its west / south
// neighbours, never on another cell's outputs, so any partition of the rows gives the same arrays.  first = 1 for the first piece of a
// field (the pieces of one field share one pair of validity flags).
extern "C" int32_t otmb_facefluxes_rows_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32, const uint8_t *wet3d,
                                            double fill, int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *const phi[6],
                                            const double *top_below, uint16_t *push_mask, int64_t j0, int64_t j1, int32_t first) {
    if (j0 < 0 || j1 > ny || j0 >= j1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "row range");
    return facefluxes_impl(ctx, umo, vmo, src_is_f32, wet3d, fill, nx, ny, nz, topology, phi, top_below, push_mask, false, false, nullptr, j0, j1,
                           first == 0);
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

extern "C" int64_t otmb_count_tables_bytes(const otmb_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int64_t n_wet) {
    if (!ctx || nx < 1 || ny < 1 || nz < 1 || n_wet < 0) return 0;
    const i64 ntiles = (n_wet + (1ll << FFC_TILE_SHIFT) - 1) >> FFC_TILE_SHIFT;
    return (int64_t)(ff_tables_static_offset(ff_rows_for(ctx, nx, ny), nx, ny, nz) + (size_t)(ntiles + 1) * sizeof(unsigned long long));
}

// nz levels from level k_own0 of a local grid of nz_ext levels (the whole grid: 0, nz); wet_base: the global 0-based wet rank of the first
// owned wet cell.  lwet3d / wetflags: the local grid's arrays; lwet: the owned cells' local linear indices.
static int32_t count_tables_impl(otmb_ctx *ctx, const int64_t *lwet3d, const int64_t *lwet, const uint8_t *wetflags, int64_t n_wet, int64_t nx,
                                 int64_t ny, int64_t nz, int32_t topology, int64_t k_own0, int64_t nz_ext, int64_t wet_base, void *tables) {
    if (!ctx || !lwet3d || !wetflags || !tables || (n_wet > 0 && !lwet)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1 || n_wet < 0 || nx * ny >= (1ll << 28) || nx * ny * nz_ext >= (1ll << 32)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    if (k_own0 < 0 || k_own0 + nz > nz_ext || nz_ext > nz + 2 || wet_base < 0) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "depth slab");
    if (topology != OTMB_BIPOLAR && topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "topology");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rows = ff_rows_for(ctx, nx, ny);
    const i64 nseg = ff_nseg(rows, nx, ny), nw = nz * nseg, ntiles = (n_wet + (1ll << FFC_TILE_SHIFT) - 1) >> FFC_TILE_SHIFT;
    KernelTimer kt(ctx, K_FF_BASES);
    hipLaunchKernelGGL(ff_count_bases_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, ctx->stream, (const i64 *)lwet3d + k_own0 * nx * ny,
                       (int)nx, (int)ny, (int)nz, nx * ny, rows, (int)nseg, (i64)wet_base, (uint32_t *)tables);
    if (ntiles > 0)
        hipLaunchKernelGGL(ff_static_counts_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, (const i64 *)lwet, wetflags, (i64)n_wet,
                           (int)nx, (int)ny, (int)nz_ext, nx * ny, (int)topology,
                           (unsigned long long *)((char *)tables + ff_tables_static_offset(rows, nx, ny, nz)));
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}

extern "C" int32_t otmb_count_tables_dev(otmb_ctx *ctx, const int64_t *lwet3d, const int64_t *lwet, const uint8_t *wetflags, int64_t n_wet,
                                         int64_t nx, int64_t ny, int64_t nz, int32_t topology, void *tables) {
    return count_tables_impl(ctx, lwet3d, lwet, wetflags, n_wet, nx, ny, nz, topology, 0, nz, 0, tables);
}
extern "C" int32_t otmb_count_tables_slab_dev(otmb_ctx *ctx, const int64_t *lwet3d, const int64_t *lwet, const uint8_t *wetflags, int64_t n_wet,
                                              int64_t nx, int64_t ny, int64_t nz, int32_t topology, const otmb_ff_slab *slab, void *tables) {
    if (!slab) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    return count_tables_impl(ctx, lwet3d, lwet, wetflags, n_wet, nx, ny, nz, topology, slab->k_own0, slab->nz_ext, slab->wet_base, tables);
}

// facefluxes + the tile counts of the transportmatrix that will be built from its fluxes (see FfCountArgs).  Grids whose row-mates can
// coincide (nx < 3) and contexts with OTMB_COUNT_IN_FF=0 take the plain kernel and write the whole push mask, as
// otmb_facefluxes_flags_dev does: the caller need not know which happened.
extern "C" int32_t otmb_facefluxes_counts_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32,
                                              const uint8_t *wetflags, double fill, int64_t nx, int64_t ny, int64_t nz,
                                              int32_t topology, double *const phi[6], uint16_t *push_mask,
                                              const otmb_ff_counts *counts) {
    if (!ctx || !counts) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    const bool can = ctx->count_in_ff != 0 && nx >= 3 && push_mask && counts->tables && counts->lwet3d && counts->mlotst && counts->zt &&
                     counts->n_wet > 0 && nx * ny * nz < (1ll << 32) && nz <= FFC_MAX_NZ;
    return facefluxes_impl(ctx, umo, vmo, src_is_f32, wetflags, fill, nx, ny, nz, topology, phi, nullptr, push_mask, false, true,
                           can ? counts : nullptr);
}

// The same for a depth slab, optionally one row band at a time (the chain over depth slabs, otmb_facefluxes_rows_dev).  A row band other
// than the whole plane needs the four-row wave geometry (its segments are whole rows); otherwise, and wherever
// otmb_facefluxes_counts_dev would fall back, the call is otmb_facefluxes_flags_dev's / _rows_dev's: mask written, no counts.
extern "C" int32_t otmb_facefluxes_slab_counts_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32, const uint8_t *wetflags,
                                                   double fill, int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *const phi[6],
                                                   const double *top_below, uint16_t *push_mask, const otmb_ff_counts *counts,
                                                   const otmb_ff_slab *slab, int64_t j0, int64_t j1, int32_t first) {
    if (!ctx || !counts || !slab || !wetflags) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    if (slab->k_own0 < 0 || slab->k_own0 + nz > slab->nz_ext || slab->nz_ext > nz + 2 || slab->wet_base < 0)
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "depth slab");
    if (j0 < 0 || j1 > ny || j0 >= j1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "row range");
    const bool whole = j0 == 0 && j1 == ny;
    bool can = ctx->count_in_ff != 0 && nx >= 3 && push_mask && counts->tables && counts->lwet3d && counts->mlotst && counts->zt &&
               counts->n_wet > 0 && nx * ny * slab->nz_ext < (1ll << 32) && slab->nz_ext <= FFC_MAX_NZ &&
               (whole || ff_rows_for(ctx, nx, ny) == 4);
    // (a piece of a call whose first piece could not count must not start counting: the key of the call is the first piece's)
    if (!first && !ctx->ffc.pieces_open) can = false;
    if (!can)
        return facefluxes_impl(ctx, umo, vmo, src_is_f32, wetflags + slab->k_own0 * nx * ny, fill, nx, ny, nz, topology, phi, top_below, push_mask,
                               false, true, nullptr, j0, j1, first == 0);
    return facefluxes_impl(ctx, umo, vmo, src_is_f32, wetflags, fill, nx, ny, nz, topology, phi, top_below, push_mask, false, true, counts, j0, j1,
                           first == 0, false, slab);
}

extern "C" int32_t otmb_facefluxes_counts_pending(const otmb_ctx *ctx) { return (ctx && ctx->ffc.valid) ? 1 : 0; }

// facefluxes of the fused step (otmb_step_dev, otmb_transportmatrix.hip): counts + ϕtop only.  `token`: the pointer the counts are keyed to in
// place of a push mask (never read or written).
int32_t otmb_facefluxes_top_counts(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32, const uint8_t *wetflags, double fill,
                                   int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *const phi[6], uint16_t *token,
                                   const otmb_ff_counts *counts) {
    return facefluxes_impl(ctx, umo, vmo, src_is_f32, wetflags, fill, nx, ny, nz, topology, phi, nullptr, token, false, true, counts, 0, -1, false, true);
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
    // (an array that held the token of a counting facefluxes call and now receives a mask for cells of its own is an ordinary push mask
    // again -- a partial one if only halo planes are written, but then the counts are still pending and take precedence)
    if (push_mask == ctx->ffc_partial_mask && !ctx->ffc.valid) ctx->ffc_partial_mask = nullptr;
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
