// otmb_tm_dense.h -- DENSE-TILE MARCH formulation of the counting and fill passes of transportmatrix
// (src/matrixbuilding.jl:128-150; same columns, same arithmetic, same CSC output as the gather kernels of
// otmb_transportmatrix.hip -- only WHERE a stencil value comes from differs).
//
// The gather kernels give every lane one wet cell and fetch its 58 stencil values from global memory: each value of
// v3D / ρ / Lwet3D crosses L2 -> L1 five times (as a cell's own and as its south / north / above / below
// neighbours') and, once a level of the grid no longer fits the caches (0.25 degree: 124 MB), the levels above and
// below come from HBM again (measured: 15.4 GB fetched for 9.3 GB of inputs).  Here a wave is 62 consecutive cells of
// ONE grid row (lanes 1..62; lanes 0 and 63 hold the west / east halo cells, wrapped periodically,
// src/gridtopology.jl:57-58) and MARCHES down the levels:
//   * east / west neighbours are the neighbouring LANES (DPP wave_shl:1 / wave_shr:1);
//   * the level above is what the wave held one step ago, the level below what it loaded one step ahead (registers);
//   * south / north rows are the only neighbour loads (same lines as the sibling waves' own loads: the four waves of
//     a workgroup are four consecutive rows);
//   * per level 18 coalesced 512-byte loads serve ~45 wet cells (the gather form: 58 loads per 64), every input value
//     is read from HBM once, and the 2-D metrics are loaded once per column instead of once per cell.
// Land lanes idle (27 % of the segment-levels of the 0.25 degree grid are all land and skip everything but their
// loads; the others hold 72 % wet lanes).  The wet lanes of a wave are consecutive columns of all five matrices, so
// the output side is the gather kernel's: packed wave scan, per-wave LDS staging, 16-byte stores from scalar bases.
// Offsets: the counting pass leaves six sums per (level, row, segment) -- in that order, which is the order of the wet
// rank (src/matrixbuilding.jl:14-15) -- and the tile scan turns them into the absolute position of every wave's run.
#pragma once
#include "otmb_tm_column.h"

#define DM_W 62     // columns a wave produces (lanes 1..62)
#define DM_ROWS 4   // rows (= waves) per workgroup
#define DM_NF 6     // scan fields per segment-level: T (union), Tadv, TκH, TκVML, TκVdeep, wet cells
#ifndef DM_WAVES_PER_SIMD
#define DM_WAVES_PER_SIMD 2
#endif
#ifndef DM_PREFETCH
#define DM_PREFETCH 1     // 1: the next level's loads are issued before this level's arithmetic (36 more live VGPRs)
#endif
#ifndef DM_LDS_METRICS
#define DM_LDS_METRICS 0  // 1: the column's fourteen 2-D metrics wait in LDS instead of 28 VGPRs
#endif
#define DM_NMET 12  // (area and mlotst stay in registers: 12 x 2 KB + the staging leaves room for three workgroups per CU)
#define DM_WSTAGE (DM_W * TM_MAXROWS)  // a wave's staging entries
#define DM_STAGE (DM_ROWS * DM_WSTAGE)

// Tables written by EARLIER launches (the counting pass, the scan) are read through the constant address space: a uniform
// load from it is a scalar load (s_load, counted by lgkmcnt).  As an ordinary global load it would be a vector load
// whose s_waitcnt vmcnt(0) also waits for every store the wave has in flight -- one full drain of the write pipeline per level.
typedef const __attribute__((address_space(4))) uint32_t dm_cu32;
typedef const __attribute__((address_space(4))) i64 dm_ci64;
typedef const __attribute__((address_space(4))) int dm_cint;
typedef const __attribute__((address_space(4))) double dm_cf64;

struct DmGeom {
    int nseg;        // segments of DM_W cells per row
    int nrowgrp;     // groups of DM_ROWS rows
    int kparts;      // the owned levels are cut into kparts pieces per (row, segment)
    const int *kown; // [2] first and last level that holds owned wet cells (device; slabs own whole levels)
};

__device__ __forceinline__ int dpp_next_i(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, false); }  // lane l <- lane l+1
__device__ __forceinline__ int dpp_prev_i(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false); }  // lane l <- lane l-1
__device__ __forceinline__ double dpp_next(double x) {
    return __hiloint2double(dpp_next_i(__double2hiint(x)), dpp_next_i(__double2loint(x)));
}
__device__ __forceinline__ double dpp_prev(double x) {
    return __hiloint2double(dpp_prev_i(__double2hiint(x)), dpp_prev_i(__double2loint(x)));
}
__device__ __forceinline__ i64 dpp_next(i64 x) {
    return (i64)(((u64)(unsigned)dpp_next_i((int)((u64)x >> 32)) << 32) | (unsigned)dpp_next_i((int)(unsigned)(u64)x));
}
__device__ __forceinline__ i64 dpp_prev(i64 x) {
    return (i64)(((u64)(unsigned)dpp_prev_i((int)((u64)x >> 32)) << 32) | (unsigned)dpp_prev_i((int)(unsigned)(u64)x));
}

// (Stencil and column_compute -- the regular-cell arithmetic, ONE copy shared with the gather kernel -- live in otmb_tm_column.h)

// lane geometry of a wave: cells i0-1 .. i0+62 of row j, wrapped periodically
struct DmLane {
    int i;         // the lane's cell
    bool active;   // lanes 1..62 inside the row produce columns
};
__device__ __forceinline__ DmLane dm_lane(int seg, int lane, int nx) {
    DmLane d;
    int il = seg * DM_W - 1 + lane;
    il %= nx;
    d.i = il < 0 ? il + nx : il;
    d.active = (lane >= 1) & (lane <= DM_W) & (seg * DM_W + lane - 1 < nx);
    return d;
}

// first / last level with owned wet cells (the slab's own levels; the whole grid otherwise)
__global__ void dm_kown_kernel(const i64 *__restrict__ lwet, i64 n_own, i64 P, int nz, int *kown) {
    if (threadIdx.x == 0) {
        i64 a = n_own > 0 ? (lwet[0] - 1) / P : 0, b = n_own > 0 ? (lwet[n_own - 1] - 1) / P : -1;
        a = a < 0 ? 0 : (a >= nz ? nz - 1 : a);
        b = b >= nz ? nz - 1 : b;
        kown[0] = (int)a;
        kown[1] = (int)b;
    }
}

// ---- counting pass: one wave per (level, row, segment) ------------------------------------------------------------
// Presence only, from the push mask (2 bytes per cell, see tm_count_kernel): own row + south / north rows + the levels
// above / below = five 128-byte loads per wave, east / west from the neighbouring lanes.
__global__ __launch_bounds__(256) void dm_count_kernel(const TmParams p, const DmGeom g) {
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const i64 nsl = (i64)nz * ny * g.nseg;
    const i64 id = (i64)blockIdx.x * 4 + wid;  // (k * ny + j) * nseg + seg
    if (id >= nsl) return;
    const int seg = (int)(id % g.nseg);
    const i64 r = id / g.nseg;
    const int j = (int)(r % ny), k = (int)(r / ny);
    uint32_t *out = p.tilesums + id * DM_NF;
    if (k < g.kown[0] || k > g.kown[1]) {  // a halo level of a depth slab: neighbours only
        if (lane < DM_NF) out[lane] = 0;
        return;
    }
    const DmLane dl = dm_lane(seg, lane, nx);
    const int i = dl.i;
    const bool hS = j > 0, hN = j + 1 < ny, hA = k > 0, hB = k + 1 < nz;
    const i64 L = (i64)k * p.P + (i64)j * nx + i;
    const unsigned sh = p.upwind ? 0u : 8u;
    const uint16_t *mk = p.mask;
    unsigned padv = 0, phh = 0, pml = 0, pdp = 0;
    const unsigned mC = (unsigned)mk[L] >> sh;
    const bool wet = dl.active && (mC & PM_WET);
    const bool regular = !(p.topo == OTMB_TRIPOLAR && j == ny - 1);
    if (regular) {
        const unsigned mS = (unsigned)mk[hS ? L - nx : L] >> sh, mN = (unsigned)mk[hN ? L + nx : L] >> sh;
        const unsigned mA = (unsigned)mk[hA ? L - p.P : L] >> sh, mB = (unsigned)mk[hB ? L + p.P : L] >> sh;
        const unsigned mE = (unsigned)dpp_next_i((int)mC), mW = (unsigned)dpp_prev_i((int)mC);
        const double mld = p.ml[(i64)j * nx + i];
        const double ztk = p.zt[k], zta = p.zt[hA ? k - 1 : k], ztb = p.zt[hB ? k + 1 : k];
        const bool wE = mE & PM_WET, wW = mW & PM_WET, wS = hS && (mS & PM_WET), wN = hN && (mN & PM_WET), wA = hA && (mA & PM_WET),
                   wB = hB && (mB & PM_WET);
        if (wet) {
            // own pushes land in wet cells (the reference indexes Lwet3D[C𝑗] unconditionally, :247 etc.); ρ (:233)
            const bool bad = ((mC & PM_W) && !wW) | ((mC & PM_E) && !wE) | ((mC & PM_S) && !wS) | ((mC & PM_N) && !wN) |
                             ((mC & PM_B) && !wB) | (hA && (mC & PM_T) && !wA);
            if (bad) raise_flag(p.flags, FLAG_FLUX_INTO_LAND);
            if (!p.rho_in_fill && p.rho && isnan(p.rho[L])) raise_flag(p.flags, FLAG_RHO_NAN);
            const bool aE = wE && (mE & PM_W), aW = wW && (mW & PM_E), aS = wS && (mS & PM_N), aN = wN && (mN & PM_S);
            const bool aA = wA && (mA & PM_B), aB = wB && (mB & PM_T);
            padv = ((unsigned)aA << S_A) | ((unsigned)aS << S_S) | ((unsigned)aW << S_WC) | ((unsigned)aE << S_EC) |
                   ((unsigned)aN << S_N) | ((unsigned)aB << S_B) | ((unsigned)(aA | aS | aW | aE | aN | aB) << S_SELF);
            phh = ((unsigned)wW << S_WC) | ((unsigned)wE << S_EC) | ((unsigned)wS << S_S) | ((unsigned)wN << S_N) |
                  ((unsigned)(wW | wE | wS | wN) << S_SELF);
            pdp = ((unsigned)wB << S_B) | ((unsigned)wA << S_A) | ((unsigned)(wA | wB) << S_SELF);
            const bool omC = ztk < mld;
            const bool mlB = wB & omC & (ztb < mld), mlA = wA & omC & (zta < mld);
            pml = ((unsigned)mlB << S_B) | ((unsigned)mlA << S_A) | ((unsigned)(mlA | mlB) << S_SELF);
        }
    } else if (wet) {  // the tripolar seam row: the generic slot logic on the mask (general_presence)
        const Cell cell = cell_of(L, nx, ny, p.P);
        general_presence(p, cell, padv, phh, pml, pdp);
    }
    u64 x = 0;
    if (wet) {
        x = p.only_t ? (u64)__popc(padv | phh | pml | pdp)
                     : ((u64)__popc(padv | phh | pml | pdp) | ((u64)__popc(padv) << 11) | ((u64)__popc(phh) << 22) |
                        ((u64)__popc(pml) << 33) | ((u64)__popc(pdp) << 43));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    const unsigned nwet = (unsigned)__popcll(__builtin_amdgcn_ballot_w64(wet));
    if (lane < DM_NF) {
        const unsigned f[DM_NF] = {(unsigned)(x & 0x7ff), (unsigned)((x >> 11) & 0x7ff), (unsigned)((x >> 22) & 0x7ff),
                                   (unsigned)((x >> 33) & 0x3ff), (unsigned)((x >> 43) & 0x3ff), nwet};
        unsigned v = 0;
#pragma unroll
        for (int q = 0; q < DM_NF; ++q)
            if (q == lane) v = f[q];
        out[lane] = v;
    }
}

// ---- fill pass: one wave per (row, segment, depth part), marching down the levels --------------------------------------
struct DmOwn {  // a level's own values that its neighbours above / below need too
    i64 lw;
    double v, rho, pt, pb;
};
struct DmRest {  // the rest of a level's loads
    double thk, pe, pw;
    i64 lwS, lwN;
    double vS, vN, tS, tN, rS, rN, pnS, psN;
};

// SEAM: the launch that takes the rows whose north neighbours lie in the row itself (the tripolar seam row, :94) through
// the generic column builder; the main launch (SEAM = false) skips them and carries none of that code or its registers.
template <bool SEAM>
__global__ __launch_bounds__(256, DM_WAVES_PER_SIMD) void dm_fill_kernel(const TmParams p, const DmGeom g, const i64 *totals) {
    __shared__ __attribute__((aligned(16))) i64 s_stage[2 * DM_STAGE];  // per wave: rows, then value bits (as tm_kernel)
#if DM_LDS_METRICS
    __shared__ double s_met[DM_NMET][256];
#endif
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (p.next_state && blockIdx.x == 0 && tid < (int)(OTMB_TM_STATE_BYTES / sizeof(int))) p.next_state[tid] = 0;
    const int nx = p.nx, ny = p.ny, nz = p.nz, up = p.upwind;
    (void)up;
    // units are dealt to the XCDs in eighths (see tm_kernel): a unit's sibling rows and segments share an L2
    i64 unit;
    {
        const i64 nt = gridDim.x, q = nt / 8, r = nt % 8, x = blockIdx.x % 8, y = blockIdx.x / 8;
        unit = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int seg = __builtin_amdgcn_readfirstlane((int)(unit % g.nseg));  // (uniform by construction; said so explicitly, so that
    const i64 ur = unit / g.nseg;                                           //  bases and table addresses live in scalar registers)
    const int rowgrp = __builtin_amdgcn_readfirstlane((int)(ur % g.nrowgrp)), kp = __builtin_amdgcn_readfirstlane((int)(ur / g.nrowgrp));
    const int j = SEAM ? ny - 1 : rowgrp * DM_ROWS + wid;  // (SEAM: the four waves of a workgroup are four depth parts of the seam row)
    if (!SEAM && unit == 0 && tid == 0 && ((dm_ci64 *)totals)[5] != p.n_own) raise_flag(p.flags, FLAG_NONCANONICAL);  // Lwet is not the list of wet cells
    if (j >= ny) return;
    const bool regular = !(p.topo == OTMB_TRIPOLAR && j == ny - 1);
    if (regular == SEAM) return;
    const int ko0 = ((dm_cint *)g.kown)[0], ko1 = ((dm_cint *)g.kown)[1] + 1;  // owned levels [ko0, ko1)
    const int nparts = SEAM ? g.kparts * DM_ROWS : g.kparts, part = SEAM ? kp * DM_ROWS + wid : kp;
    const int k_lo = __builtin_amdgcn_readfirstlane(ko0 + (int)((i64)(ko1 - ko0) * part / nparts)),
              k_hi = __builtin_amdgcn_readfirstlane(ko0 + (int)((i64)(ko1 - ko0) * (part + 1) / nparts));
    if (k_lo >= k_hi) return;

    const DmLane dl = dm_lane(seg, lane, nx);
    const int i = dl.i;
    const bool hS = j > 0, hN = j + 1 < ny;
    const unsigned nx8 = (unsigned)nx * 8u;
    const unsigned o = ((unsigned)j * (unsigned)nx + (unsigned)i) * 8u;  // byte offset inside a level (P < 2^27)
    const unsigned oS = hS ? o - nx8 : o, oN = hN ? o + nx8 : o;
    const i64 P8 = p.P * 8;

    // ---- per column: the 2-D metrics (:366-411), once for the whole march ----
    Stencil st;
    {
        const char *eWp = (const char *)p.edge[OTMB_DIR_WEST], *eEp = (const char *)p.edge[OTMB_DIR_EAST],
                   *eSp = (const char *)p.edge[OTMB_DIR_SOUTH], *eNp = (const char *)p.edge[OTMB_DIR_NORTH];
        const char *dWp = (const char *)p.dist[OTMB_DIR_WEST], *dEp = (const char *)p.dist[OTMB_DIR_EAST],
                   *dSp = (const char *)p.dist[OTMB_DIR_SOUTH], *dNp = (const char *)p.dist[OTMB_DIR_NORTH];
        st.eW_c = ldd(eWp, o); st.eE_c = ldd(eEp, o); st.eS_c = ldd(eSp, o); st.eN_c = ldd(eNp, o);
        st.dW_c = ldd(dWp, o); st.dE_c = ldd(dEp, o); st.dS_c = ldd(dSp, o); st.dN_c = ldd(dNp, o);
        st.eN_s = ldd(eNp, oS); st.dN_s = ldd(dNp, oS);
        st.eS_n = ldd(eSp, oN); st.dS_n = ldd(dSp, oN);  // oppdir = south away from the seam row (:407)
        st.ar = ldd((const char *)p.area, o); st.mld = ldd((const char *)p.ml, o);
#if DM_LDS_METRICS
        const double mv[DM_NMET] = {st.eW_c, st.eE_c, st.eS_c, st.eN_c, st.dW_c, st.dE_c, st.dS_c, st.dN_c, st.eN_s, st.dN_s, st.eS_n, st.dS_n};
#pragma unroll
        for (int q = 0; q < DM_NMET; ++q) s_met[q][tid] = mv[q];  // (each lane reads back only what it wrote: no barrier)
#else
        st.eE_w = dpp_prev(st.eE_c); st.dE_w = dpp_prev(st.dE_c);  // west cell's east edge / distance to its east neighbour
        st.eW_e = dpp_next(st.eW_c); st.dW_e = dpp_next(st.dW_c);
#endif
    }

    const char *lwp = (const char *)p.lw, *vp = (const char *)p.v, *tp = (const char *)p.thk, *rp = (const char *)p.rho;
    const char *pep = (const char *)p.phi[OTMB_EAST], *pwp = (const char *)p.phi[OTMB_WEST], *pnp = (const char *)p.phi[OTMB_NORTH],
               *psp = (const char *)p.phi[OTMB_SOUTH], *ptp = (const char *)p.phi[OTMB_TOP], *pbp = (const char *)p.phi[OTMB_BOTTOM];
    const double rho_s = p.rho_s;
    // A level's base pointers are scalars made opaque to the optimiser: left alone it hoists `array + lane offset` out of the
    // march as eighteen per-lane 64-bit pointers (36 VGPRs, a 64-bit add per load); this way every load takes the
    // `global_load v, v_offset32, s[base]` form with three lane offsets for the whole kernel.
    // (global address space kept through the asm: a generic pointer would turn the loads into flat_load)
    typedef const __attribute__((address_space(1))) char gchar;
    typedef const __attribute__((address_space(1))) double gf64;
    typedef const __attribute__((address_space(1))) i64 gi64;
    auto lvl = [&](const char *a, i64 b) { gchar *q = (gchar *)(a + b); asm volatile("" : "+s"(q)); return q; };
    // (the lane offset is made opaque at every load as well: its zero-extension hoisted into another block hides the
    //  scalar-base + 32-bit-offset addressing mode from instruction selection)
    auto ldd = [&](gchar *b, unsigned byteoff) { asm volatile("" : "+v"(byteoff)); return *(gf64 *)(b + byteoff); };
    auto ldi = [&](gchar *b, unsigned byteoff) { asm volatile("" : "+v"(byteoff)); return *(gi64 *)(b + byteoff); };
    auto load_own = [&](int kk) {
        const int kc = kk < 0 ? 0 : (kk >= nz ? nz - 1 : kk);  // (outside the grid: masked by hA / hB)
        const i64 b = (i64)kc * P8;
        DmOwn w;
        w.lw = ldi(lvl(lwp, b), o);
        w.v = ldd(lvl(vp, b), o);
        w.rho = rp ? ldd(lvl(rp, b), o) : rho_s;
        w.pt = ldd(lvl(ptp, b), o);
        w.pb = ldd(lvl(pbp, b), o);
        return w;
    };
    auto load_rest = [&](int kk) {
        const int kc = kk >= nz ? nz - 1 : kk;
        const i64 b = (i64)kc * P8;
        gchar *lwl = lvl(lwp, b), *vl = lvl(vp, b), *tl = lvl(tp, b), *rl = rp ? lvl(rp, b) : nullptr;
        DmRest w;
        w.thk = ldd(tl, o); w.pe = ldd(lvl(pep, b), o); w.pw = ldd(lvl(pwp, b), o);
        w.lwS = ldi(lwl, oS); w.lwN = ldi(lwl, oN);
        w.vS = ldd(vl, oS); w.vN = ldd(vl, oN);
        w.tS = ldd(tl, oS); w.tN = ldd(tl, oN);
        w.rS = rp ? ldd(rl, oS) : rho_s; w.rN = rp ? ldd(rl, oN) : rho_s;
        w.pnS = ldd(lvl(pnp, b), oS); w.psN = ldd(lvl(psp, b), oN);
        return w;
    };

    // rows and value bits of the wave's entries are staged here (see tm_kernel)
    i64 *my_row = s_stage + wid * DM_WSTAGE;
    i64 *my_val = my_row + DM_STAGE;
    typedef i64 i64x2 __attribute__((ext_vector_type(2)));
    typedef i64x2 i64x2g __attribute__((aligned(8)));

#if DM_PREFETCH
    DmOwn A = load_own(k_lo - 1), C = load_own(k_lo), B = load_own(k_lo + 1);
    DmRest R = load_rest(k_lo);
#else
    DmOwn A = load_own(k_lo - 1), C = load_own(k_lo), B;
    DmRest R;
#endif
    for (int k = k_lo; k < k_hi; ++k) {
#if DM_PREFETCH
        // the next step's loads are in flight while this level is turned into columns and stored
        const DmOwn B2 = load_own(k + 2);
        const DmRest R2 = load_rest(k + 1);
#else
        B = load_own(k + 1);
        R = load_rest(k);
#endif
        const i64 id = wave_uniform(((i64)k * ny + j) * g.nseg + seg);
        const bool wet = dl.active && C.lw != 0;
        const u64 wetmask = __builtin_amdgcn_ballot_w64(wet);
        const i64 c = C.lw;  // the column's (global) wet rank
        Column col;
        col.padv = col.phh = col.pml = col.pdp = 0;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) { col.idx[s] = 0; col.bef[s] = 0; col.adv[s] = 0; col.hh[s] = 0; col.ml[s] = 0; col.dp[s] = 0; col.tv[s] = 0; }
        unsigned pT = 0, nU = 0, nA = 0, nH = 0, nM = 0, nD = 0;
        if (wetmask != 0) {  // (a level that is all land in this segment skips the arithmetic, not the store schedule below)
            // east / west neighbours sit in the neighbouring lanes
            st.lE = dpp_next(C.lw); st.lW = dpp_prev(C.lw);
            st.vE = dpp_next(C.v); st.vW = dpp_prev(C.v);
            st.rE = dpp_next(C.rho); st.rW = dpp_prev(C.rho);
            st.tE = dpp_next(R.thk); st.tW = dpp_prev(R.thk);
            st.gE = dpp_next(R.pw);  // the east cell pushes with its west flux (:244)
            st.gW = dpp_prev(R.pe);  // the west cell with its east flux (:253)
#if DM_LDS_METRICS
            st.eE_w = dpp_prev(s_met[1][tid]); st.dE_w = dpp_prev(s_met[5][tid]);
            st.eW_e = dpp_next(s_met[0][tid]); st.dW_e = dpp_next(s_met[4][tid]);
#endif
            if (wet) {
                if (!SEAM) {
#if DM_LDS_METRICS
                    st.eW_c = s_met[0][tid]; st.eE_c = s_met[1][tid]; st.eS_c = s_met[2][tid]; st.eN_c = s_met[3][tid];
                    st.dW_c = s_met[4][tid]; st.dE_c = s_met[5][tid]; st.dS_c = s_met[6][tid]; st.dN_c = s_met[7][tid];
                    st.eN_s = s_met[8][tid]; st.dN_s = s_met[9][tid]; st.eS_n = s_met[10][tid]; st.dS_n = s_met[11][tid];
#endif
                    st.lS = R.lwS; st.lN = R.lwN; st.lA = A.lw; st.lB = B.lw;
                    st.gS = R.pnS; st.gN = R.psN; st.gA = A.pb; st.gB = B.pt;
                    st.vC = C.v; st.vS = R.vS; st.vN = R.vN; st.vA = A.v; st.vB = B.v;
                    st.rC = C.rho; st.rS = R.rS; st.rN = R.rN; st.rA = A.rho; st.rB = B.rho;
                    st.tC = R.thk; st.tS = R.tS; st.tN = R.tN;
                    dm_cf64 *zt = (dm_cf64 *)p.zt;
                    st.ztk = zt[k]; st.zta = zt[k > 0 ? k - 1 : k]; st.ztb = zt[k + 1 < nz ? k + 1 : k];
                    column_compute<true>(p, st, i, j, k, c, col);
                } else {  // the tripolar seam row (:94): the generic column builder on global memory
                    const Cell cell = cell_of((i64)k * p.P + (i64)j * nx + i, nx, ny, p.P);
                    build_column(p, cell, c, col);
                }
                const unsigned uni = col.padv | col.phh | col.pml | col.pdp;
                nU = __popc(uni);
                if (!p.only_t) { nA = __popc(col.padv); nH = __popc(col.phh); nM = __popc(col.pml); nD = __popc(col.pdp); }
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    col.tv[s] = t_value(col, s);
                    if (((uni >> s) & 1u) && col.tv[s] != 0.0) pT |= 1u << s;
                }
                if (pT != uni) raise_flag(p.flags, FLAG_T_CANCEL);
            }
        }
        // ---- the write phase: THE SAME INSTRUCTION STREAM FOR EVERY LEVEL ------------------------------------------------
        // s_waitcnt vmcnt(N) lets the wave's N youngest vector-memory operations stay in flight, loads and stores counted
        // together in issue order.  The next level's loads were issued BEFORE this level's stores, so using them only needs
        // the operations older than they are -- but the compiler must pick N at compile time, as the SMALLEST number of
        // younger operations over all paths.  With a store loop of run-time length (or a path that skips the stores) that
        // minimum is the handful of loads alone, and every level would wait for all of its predecessor's stores to be
        // acknowledged by the memory system (measured: half of every wave's life, 9.3 ms at 0.25 degree).  Hence: a fixed
        // schedule of predicated stores -- ceil(62 * rows / 128) iterations per matrix -- executed by every level, all-land
        // levels included (their stores are masked off and cost an issue slot each).
        // packed scan T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10
        const u64 mine = (u64)nU | ((u64)nA << 11) | ((u64)nH << 22) | ((u64)nM << 33) | ((u64)nD << 43);
        u64 incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        const u64 excl = incl - mine;
        const u64 wtot = ((u64)__builtin_amdgcn_readlane((unsigned)(incl >> 32), 63) << 32) | __builtin_amdgcn_readlane((unsigned)incl, 63);
        const unsigned ex[5] = {(unsigned)(excl & 0x7ff), (unsigned)((excl >> 11) & 0x7ff), (unsigned)((excl >> 22) & 0x7ff),
                                (unsigned)((excl >> 33) & 0x3ff), (unsigned)((excl >> 43) & 0x3ff)};
        const unsigned wc[5] = {(unsigned)(wtot & 0x7ff), (unsigned)((wtot >> 11) & 0x7ff), (unsigned)((wtot >> 22) & 0x7ff),
                                (unsigned)((wtot >> 33) & 0x3ff), (unsigned)((wtot >> 43) & 0x3ff)};
        const unsigned cn[5] = {nU, nA, nH, nM, nD};
        // what the counting pass reserved for this wave (six uniform words) and where its runs start
        dm_cu32 *sums = (dm_cu32 *)(p.tilesums + id * DM_NF);
        dm_ci64 *offs = (dm_ci64 *)(p.tileoffs + id * DM_NF);
        bool ok = true;
#pragma unroll
        for (int m = 0; m < TM_NF; ++m) ok &= sums[m] == wc[m];
        ok &= sums[5] == (unsigned)__popcll(wetmask);
        // Lwet3D is the wet rank in linear-index order (makeindices, :14-20): the wave's wet cells are consecutive ranks
        // and the first follows the cells of all earlier segment-levels
        const i64 wprefix = offs[5];
        const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(wetmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wetmask, 0u));
        const i64 w = wprefix + before;  // the column's index in this launch
        const bool canonical = !wet || (c == p.wet_base + w + 1 && w < p.n_own);
        if (!canonical) raise_flag(p.flags, FLAG_NONCANONICAL);
        if (!ok && lane == 0) raise_flag(p.flags, FLAG_COUNT_MISMATCH);
        const bool good = ok && __builtin_amdgcn_ballot_w64(!canonical) == 0;  // (uniform) nothing is written otherwise
        const bool wetw = wet && good;
        i64 g0[5];
#pragma unroll
        for (int m = 0; m < TM_NF; ++m) g0[m] = offs[m];
#pragma unroll
        for (int m = 0; m < TM_NF; ++m) {
            const i64 cp = p.nnz_base[m] + g0[m] + ex[m] + 1;
            const bool on = wetw && (m == 0 || !p.only_t);
            if (on) p.colptr[m][w] = cp;
            if (on && w + 1 == p.n_own) p.colptr[m][p.n_own] = cp + cn[m];  // the closing entry
        }
        const unsigned vslots = (1u << S_A) | (1u << S_SELF) | (1u << S_B);
        const unsigned ops = p.only_t ? 0u : ~0u;
        const unsigned pm[5] = {pT, col.padv & ops, col.phh & ops, col.pml & vslots & ops, col.pdp & vslots & ops};
#pragma unroll
        for (int m = 0; m < TM_NF; ++m) {
            const i64 run0 = g0[m];
            char *rvb = (char *)(p.rowval[m] + run0);
            char *nzb = (char *)(p.nzval[m] + run0);
            if (wetw) {
                const unsigned q0 = ex[m];
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    if ((pm[m] >> s) & 1u) {
                        const unsigned q = q0 + __popc(pm[m] & col.bef[s]);
                        const double v = (m == 0) ? col.tv[s] : (m == 1) ? col.adv[s] : (m == 2) ? col.hh[s] : (m == 3) ? col.ml[s] : col.dp[s];
                        my_row[q] = col.idx[s];
                        my_val[q] = __double_as_longlong(v);
                    }
                }
                if (m == 0 && (unsigned)__popc(pT) != nU) {  // exact cancellation: row 0 marks the unused reserved slots (tfix_*)
                    for (unsigned e = __popc(pT); e < nU; ++e) { my_row[q0 + e] = 0; my_val[q0 + e] = 0; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            bool room = true;
            if (p.cap[m] > 0) {
                room = run0 + wc[m] <= p.cap[m];
                if (!room && good && lane == 0) raise_flag(p.flags, FLAG_CAPACITY);
            }
            const unsigned cnt = (good && room) ? wc[m] : 0u;
            const int nit = (m < 2) ? 4 : (m == 2) ? 3 : 2;  // 62 columns of at most 7, 7, 5, 3, 3 rows, 128 entries per iteration (m unrolled: constant)
#pragma unroll
            for (int it = 0; it < 4; ++it) {  // pairs of entries: 16-byte stores at 8-byte alignment
                const unsigned u = (unsigned)it * 128u + 2u * (unsigned)lane;
                if (it < nit && u + 1 < cnt) {
                    *(i64x2g *)(rvb + u * 8u) = *(const i64x2 *)(my_row + u);
                    *(i64x2g *)(nzb + u * 8u) = *(const i64x2 *)(my_val + u);
                }
            }
            if ((lane < 2) & ((cnt & 1u) == 1u)) {  // an odd run's last entry: lane 0 the row, lane 1 the value
                const unsigned e = cnt - 1;
                i64 *dst = (lane == 0) ? (i64 *)(rvb + e * 8u) : (i64 *)(nzb + e * 8u);
                *dst = (lane == 0) ? my_row[e] : my_val[e];
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // march on: this level becomes the level above
        A.lw = C.lw; A.v = C.v; A.rho = C.rho; A.pb = C.pb;
        C = B;
#if DM_PREFETCH
        B = B2;
        R = R2;
#endif
    }
}
