// otmb_transportmatrix.hip -- fused assembly of (T, Tadv, TκH, TκVML, TκVdeep) in CSC.
//
// Replaces, on the device, the whole of `transportmatrix` (src/matrixbuilding.jl:128-150): the three
// COO generators, the four sparse() calls and the three sparse adds.  No COO is materialised.
//
// Work decomposition: one thread per WET cell (driven by Lwet, so every lane works), one workgroup
// (tile) per 256 consecutive wet cells = 256 consecutive columns of all five matrices.  Per tile:
//   1. every thread builds its column in registers (otmb_tm_column.h);
//   2. a packed 64-bit block scan gives each column's offset inside the tile for the five matrices;
//   3. the tile's global offsets come from the tile counts (counted by facefluxes for the fluxes it writes, by tm_count_kernel
//      otherwise) + the tile scan: inputs are read once and outputs written once -- the algorithmic HBM traffic;
//   4. entries are staged through LDS, per wave, and streamed out with 16-byte-per-lane non-temporal stores from scalar run
//      bases (a column's entries are contiguous, a wave's 64 columns are one contiguous run in each matrix).
#include <cstdlib>

#include "otmb_tm_column.h"

#ifndef TM_THREADS
#define TM_THREADS 256  // measured: one-wave (64-thread) tiles are no faster in fill
#endif
#define TM_NF 5
#ifndef TM_WAVES_PER_SIMD
#define TM_WAVES_PER_SIMD 3  // measured: capping at 128 VGPRs (4 waves/SIMD) spills and is 8 % slower
#endif
#define TM_MAXROWS 7  // rows per column: A, S, W, SELF, E, N|fold, B
#ifndef OTMB_MARCH_AUTO_ROWS
#define OTMB_MARCH_AUTO_ROWS 8  // tile order when the caller does not choose: march order, bands of 8 rows (with the matrices written by
                                // non-temporal stores: -8 % against wet-rank order at 1 and at 0.25 degree, R = 2 ... 32 within 1 %)
#endif
#ifndef OTMB_MARCH_AUTO_COLS
#define OTMB_MARCH_AUTO_COLS 1536  // ... and, on grids with longer rows, blocks of at most this many columns (i) of a band: what an XCD's L2 (4 MB) has to keep
                                   // from one level of a block to the next is rows x columns cells of Lwet3D / v3D / rho; with whole rows of 3600 cells
                                   // (0.1 degree) every line above / below came from HBM again (fetch 70 GB for 49 GB of touched inputs, profiles/r04 section 10)
#endif
#define TM_INFILL_GROUPS 64  // up to this many scan groups the fill pass adds the group bases itself
#define TM_WSTAGE (64 * TM_MAXROWS + 2)  // per-wave staging entries (+2: parity shift for 16-byte stores)
#define TM_STAGE ((TM_THREADS / 64) * TM_WSTAGE)

struct TmPlan {
    otmb_tm_args args;  // device pointers
    i64 ntiles = 0;
    i64 nnz[5] = {0, 0, 0, 0, 0};
    bool valid = false;
    bool onepass_pending = false;
    i64 wet_base = 0;
    i64 nnz_base[5] = {0, 0, 0, 0, 0};
    bool rho_in_fill = false;  // the plan took its counts from facefluxes: no pass has looked at ρ yet, the fill pass does (:233)
    // otmb_tm_args.given: operators the caller passes (bit m).  derived: bit for bit what the fill pass computes -- re-derived in registers, not
    // materialised; foreign: any other matrix -- T is then the device sparse add of the four operands (two-phase protocol only)
    unsigned given = 0, derived = 0, foreign = 0;
    unsigned read = 0;         // (subset of derived) the derived ROWS with other values: not materialised either, but the fill pass reads the values
    unsigned skip = 0;         // matrices the kernels neither count nor write (TmParams.skip)
    bool want_t = true;        // the caller wants T (otmb_tm_args.skip_ops bit 0 clear)
    i64 built_nnz[5] = {0, 0, 0, 0, 0};  // (foreign) the counts of the matrices the kernel writes; nnz[0] is then the sparse adds' bound
};

// fields of the packed count word (T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10) that belong to the matrices NOT in `skip`
static u64 keep_mask(unsigned skip) {
    static const u64 field[5] = {0x7ffull, 0x7ffull << 11, 0x7ffull << 22, 0x3ffull << 33, 0x3ffull << 43};
    u64 k = 0;
    for (int m = 0; m < 5; ++m)
        if (!((skip >> m) & 1u)) k |= field[m];
    return k;
}
static unsigned given_mask(const otmb_tm_args &a) {
    unsigned g = 0;
    for (int m = 1; m < 5; ++m)
        if (a.given[m].colptr) g |= 1u << m;
    return g;
}

// A value every lane of the wave holds identically, moved to scalar registers.
__device__ __forceinline__ i64 wave_uniform(i64 x) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u64)x);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((u64)x >> 32));
    return (i64)(((u64)hi << 32) | lo);
}

// ---- COUNT pass as its own kernel: presence only, TPB tiles per workgroup -------------------------------
// The pass is a chain of dependent loads (Lwet -> neighbours' Lwet3D / fluxes) with almost no arithmetic, i.e.
// latency bound; giving every thread one cell of each of TPB tiles puts TPB independent chains in flight.
__device__ __forceinline__ u64 count_cell(const TmParams &p, i64 tile, int tid) {
    const i64 w0 = tile * TM_THREADS, w = w0 + tid;
    if (w0 >= p.n_own) return 0;
    // every index load is issued before anything is tested: one memory round trip in front of the mask loads
    const i64 wlast = (w0 + TM_THREADS - 1 < p.n_own) ? w0 + TM_THREADS - 1 : p.n_own - 1;
    const i64 wc = (w < p.n_own) ? w : wlast;
    const i64 Lmin = p.lwet[w0] - 1, Lmax = p.lwet[wlast] - 1;
    const i64 L = p.lwet[wc] - 1;
    const i64 Lnext_ld = p.lwet[(wc + 1 < p.n_own) ? wc + 1 : wc] - 1;
    const i64 Lnext = (wc + 1 < p.n_own) ? Lnext_ld : p.G;
    const i64 base_elem = (Lmin > p.P) ? Lmin - p.P : 0;
    const bool span_ok = (Lmax + p.P - base_elem) < (1ll << 28) && Lmin >= 0 && Lmax < p.G && Lmin <= Lmax;
    if (!span_ok) {
        if (tid == 0) raise_flag(p.flags, FLAG_NONCANONICAL);
        return 0;
    }
    if (w >= p.n_own) return 0;
    if (L < Lmin || L > Lmax || Lnext <= L) {
        raise_flag(p.flags, FLAG_NONCANONICAL);
        return 0;
    }
    TileBase tb;
    tb.lw = (const char *)(p.lw + base_elem);
    tb.rho = p.rho ? (const char *)(p.rho + base_elem) : nullptr;
    tb.mk = (const char *)(p.mask + base_elem);
    tb.v = tb.thk = tb.pe = tb.pw = tb.pn = tb.ps = tb.pt = tb.pb = tb.pu = tb.pv = nullptr;  // not read by the presence pass
    const Cell cell = cell_of(L, p.nx, p.ny, p.P);
    const unsigned oC = (unsigned)(L - base_elem) * 8u;
    // (Lwet3D[Lwet[w]] == w + 1 is verified by the fill pass, which loads Lwet3D anyway)
    unsigned padv, phh, pml, pdp;
    const bool regular = (p.nx >= 3) && !(p.topo == OTMB_TRIPOLAR && cell.j == p.ny - 1);
    if (regular) {
        fast_presence(p, tb, oC, cell.i, cell.j, cell.k, padv, phh, pml, pdp);
    } else {
        general_presence(p, cell, padv, phh, pml, pdp);
    }
    return ((u64)__popc(padv | phh | pml | pdp) | ((u64)__popc(padv) << 11) | ((u64)__popc(phh) << 22) | ((u64)__popc(pml) << 33) |
            ((u64)__popc(pdp) << 43)) & p.keep;  // (matrices that are not materialised count nothing: TmParams.skip)
}

#ifndef TM_COUNT_TPB
#define TM_COUNT_TPB 1  // measured at 1 deg: 1 -> 0.099 ms, 2 -> 0.109 ms, 4 -> 0.115 ms (count as a mode of tm_kernel: 0.116 ms)
#endif
template <int TPB>
__global__ __launch_bounds__(TM_THREADS) void tm_count_kernel(const TmParams p, i64 ntiles) {
    __shared__ u64 wave_tot[TPB][TM_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // Tile(s) of this workgroup.  Like the fill pass, XCD x (workgroups are dealt round-robin over the 8 XCDs) takes the x-th
    // contiguous eighth of the tile sequence, so that the mask lines of the rows south / north and of the levels above / below --
    // read again by later tiles -- are found in the same L2 (count_order 1: wet-rank sequence, 2: the fill pass's march sequence).
    i64 blk = blockIdx.x;
    if (p.count_order) {
        unsigned pos;
        if (!xcd_position(blockIdx.x, gridDim.x, 0u, pos)) return;
        blk = pos;
        if (TPB == 1 && p.count_order == 2 && p.order) blk = p.order[pos];
    }
    u64 mine[TPB];
#pragma unroll
    for (int q = 0; q < TPB; ++q) mine[q] = count_cell(p, blk * TPB + q, tid);
#pragma unroll
    for (int q = 0; q < TPB; ++q) {  // wave totals (the in-tile offsets are recomputed by the fill pass)
        u64 x = mine[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0) wave_tot[q][wid] = x;
    }
    __syncthreads();
    if (tid < TPB * TM_NF) {
        const int q = tid / TM_NF, m = tid - q * TM_NF;
        const i64 tile = blk * TPB + q;
        if (tile < ntiles) {
            u64 all = 0;
#pragma unroll
            for (int w = 0; w < TM_THREADS / 64; ++w) all += wave_tot[q][w];
            const unsigned sh = (m == 0) ? 0 : (m == 1) ? 11 : (m == 2) ? 22 : (m == 3) ? 33 : 43;
            p.tilesums[tile * TM_NF + m] = (unsigned)((all >> sh) & ((m < 3) ? 0x7ffu : 0x3ffu));
        }
    }
}

static_assert(TM_THREADS == (1 << FFC_TILE_SHIFT), "the counts in facefluxes are per tile of TM_THREADS columns");

// FUSED: the fused step's fill pass (otmb_step_dev): five of the six fluxes re-derived from umo / vmo / ϕtop (1 Float64, 2 Float32).
// GIVEN (otmb_tm_args.given), bit 0 -- HREAD: a given TκH with the derived rows is READ where it lies instead of re-derived (TmParams.hcp / hx):
// fewer L1 requests per column when its values are the derived ones, the only way when they are not (another κH); bit 1 -- DREAD: a given
// TκVdeep with the derived rows and OTHER values (another κVdeep) is read likewise (TmParams.dcp / dx).  Instantiations, not branches: a uniform
// branch on dx in the default kernel measured +1 ... 2 % (profiles/r06/call17_dx_*.jsonl).
template <int FUSED = 0, int GIVEN = 0>
__global__ __launch_bounds__(TM_THREADS, TM_WAVES_PER_SIMD) void tm_kernel(const TmParams p) {
    constexpr bool HREAD = (GIVEN & 1) != 0, DREAD = (GIVEN & 2) != 0;
    __shared__ u64 wave_tot[TM_THREADS / 64];
    __shared__ i64 s_prefix[TM_NF];
    __shared__ unsigned s_presum[TM_NF];
    __shared__ __attribute__((aligned(16))) i64 s_stage[2 * TM_STAGE];  // rows, then value bits
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    Stamps st;
#ifdef OTMB_DBG_STAMPS
    for (int q = 0; q < OTMB_NSTAMP; ++q) st.t[q] = 0;
#endif
    STAMP(st, 0, 0);
#ifdef OTMB_DBG_STAMPS
    u64 rt_entry;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_entry)::"memory");
#endif
    if (p.next_state && blockIdx.x == 0 && tid < (int)(OTMB_TM_STATE_BYTES / sizeof(int))) p.next_state[tid] = 0;

    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Give XCD x the x-th
    // contiguous eighth of the tiles, so that a tile's south/north rows and the levels above/below, which
    // the same XCD touched a little earlier, are L2 hits instead of fabric re-reads.  Speed only: any
    // bijection is correct.
    i64 tile;
    {
        unsigned pos;
        if (!xcd_position(blockIdx.x, p.nt_order, p.nheavy, pos)) return;  // (the whole workgroup: no barrier has been reached)
        tile = pos;
        if (p.order) tile = p.order[pos];  // march order: heavy tiles first, then the XCD's eighth is a run of (row band, level) buckets
    }
#ifdef OTMB_DBG_STAMPS_ORDER  // diagnostic (tools/stamps.py): when is the tile id known (kernel arguments + tile order)
    STAMP(st, 6, 1);
#endif
    // The tile's reserved offsets (counting pass + scan) do not depend on anything this workgroup computes: their loads
    // are issued at the top (right after the index loads below), so that this memory round trip runs beside the Lwet and
    // stencil round trips instead of after the arithmetic (tools/stamps.py: the late fetch held every wave for ~15 % of its life).
    const i64 w0 = tile * TM_THREADS;
    const i64 w = w0 + tid;
    const bool valid = w < p.n_own;

    // tile-uniform base pointers: all neighbours of all cells of the tile sit at non-negative 32-bit
    // byte offsets from them.  (The lanes' own index loads are issued together with the tile's two: one round trip.)
    const i64 wlast = (w0 + TM_THREADS - 1 < p.n_own) ? w0 + TM_THREADS - 1 : p.n_own - 1;
    const i64 wcl = valid ? w : wlast;
    const i64 L_own = p.lwet[wcl] - 1;
    const i64 Lnext_ld = p.lwet[(wcl + 1 < p.n_own) ? wcl + 1 : wcl] - 1;  // (unconditional: a branch here would wait for the load above)
    const i64 Lnext_own = (wcl + 1 < p.n_own) ? Lnext_ld : p.G;
    const i64 Lmin = p.lwet[w0] - 1;
    const i64 Lmax = p.lwet[wlast] - 1;
    i64 hq = 0, dq = 0;
    // (with the index loads: the column's first entry in the given matrix.  Never negative for the arrays the comparing pass saw; a device caller
    // who rewrote them in place without otmb_ctx_forget_given gets wrong values, not a fault: the reads below are clamped into the arrays)
    if (HREAD) { hq = p.hcp[wcl] - p.hcp[0]; hq = hq > 0 ? hq : 0; }
    if (DREAD) { dq = p.dcp[wcl] - p.dcp[0]; dq = dq > 0 ? dq : 0; }
    unsigned pre_sum = 0;
    i64 pre_off = 0;
    if (tid < TM_NF) {
        pre_sum = p.tilesums[tile * TM_NF + tid];
        pre_off = p.tileoffs[tile * TM_NF + tid];
        if (p.gsum) {  // offsets are relative to the tile's scan group: add the totals of the groups before it, eight loads in flight
            const i64 g = tile / OTMB_SCAN_GROUP;
            for (i64 q0 = 0; q0 < g; q0 += 8) {
                i64 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = p.gsum[((q0 + u < g) ? q0 + u : 0) * TM_NF + tid];
#pragma unroll
                for (int u = 0; u < 8; ++u) pre_off += (q0 + u < g) ? t[u] : 0;
            }
        }
    }
    const i64 base_elem = (Lmin > p.P) ? Lmin - p.P : 0;
    const bool span_ok = (Lmax + p.P - base_elem) < (1ll << 28) && Lmin >= 0 && Lmax < p.G && Lmin <= Lmax;
    if (!span_ok && tid == 0) raise_flag(p.flags, FLAG_NONCANONICAL);
    TileBase tb;
    tb.lw = (const char *)(p.lw + base_elem);
    tb.v = (const char *)(p.v + base_elem);
    tb.thk = (const char *)(p.thk + base_elem);
    tb.rho = p.rho ? (const char *)(p.rho + base_elem) : nullptr;
    tb.pt = (const char *)(p.phi[OTMB_TOP] + base_elem);
    if (FUSED == 0) {
        tb.pe = (const char *)(p.phi[OTMB_EAST] + base_elem);
        tb.pw = (const char *)(p.phi[OTMB_WEST] + base_elem);
        tb.pn = (const char *)(p.phi[OTMB_NORTH] + base_elem);
        tb.ps = (const char *)(p.phi[OTMB_SOUTH] + base_elem);
        tb.pb = (const char *)(p.phi[OTMB_BOTTOM] + base_elem);
        tb.pu = tb.pv = nullptr;
    } else {
        tb.pe = tb.pw = tb.pn = tb.ps = tb.pb = nullptr;
        tb.pu = (const char *)p.umo + base_elem * (FUSED == 2 ? 4 : 8);
        tb.pv = (const char *)p.vmo + base_elem * (FUSED == 2 ? 4 : 8);
    }
    tb.mk = nullptr;  // the push mask is read by the counting pass only

    // ---- 1. the column ----
    // T's rows are RESERVED as the union of the four operators' rows (known without arithmetic); the rows
    // actually stored are those whose sum is non-zero (:147).  Exact cancellation is rare: the column is
    // written left-aligned in its reserved slots, the shortfall is flagged and the host compacts T.
    Column col;
    unsigned pT = 0, nU = 0, nA = 0, nH = 0, nM = 0, nD = 0;
    bool live = false;
    if (valid && span_ok) {
        const i64 L = L_own, Lnext = Lnext_own;
        const i64 c = p.wet_base + w + 1;  // this column's (global) wet rank
        STAMP(st, 1, 1);  // Lwet is back
        // Lwet ascending inside [Lmin, Lmax] and Lwet3D[Lwet[w]] == w + 1: together they make the wet
        // rank monotone in the linear index, which is what orders the rows of a column
        if (L < Lmin || L > Lmax || Lnext <= L) {
            raise_flag(p.flags, FLAG_NONCANONICAL);
        } else {
            const Cell cell = cell_of(L, p.nx, p.ny, p.P);
            const unsigned oC = (unsigned)(L - base_elem) * 8u;
            // Lwet3D[Lwet[w]] == w + 1 is verified with a load that travels WITH the stencil loads (every stencil address
            // follows from L and the tile's base, none from Lwet3D's contents, so nothing is read out of bounds if the
            // check fails): a separate round trip in front of them cost every wave ~12 % of its life.
            bool canonical;
            const bool regular = (p.nx >= 3) && !(p.topo == OTMB_TRIPOLAR && cell.j == p.ny - 1);
            {
                if (regular) canonical = fast_column<FUSED, HREAD>(p, tb, oC, cell.i, cell.j, cell.k, c, col, st, hq);  // (the value-free input checks ran with the counts)
                else {
                    canonical = ldi(tb.lw, oC) == c;
                    if (canonical) {
                        build_column(p, cell, c, col);
                        // (seam row, nx < 3: the generic builder derived TκH's values; the given ones take their place)
                        if (HREAD) given_values<(1u << S_S) | (1u << S_SELF) | (1u << S_EC) | (1u << S_WC) | (1u << S_FQ) | (1u << S_N)>(col.hh, col.phh, col.bef, p.hx, hq, p.hnnz);
                    }
                }
            }
            if (!canonical) {
                raise_flag(p.flags, FLAG_NONCANONICAL);
            } else {
                live = true;
                if (DREAD) given_values<(1u << S_A) | (1u << S_SELF) | (1u << S_B)>(col.dp, col.pdp, col.bef, p.dx, dq, p.dnnz);
                const unsigned uni = col.padv | col.phh | col.pml | col.pdp;
                nU = __popc(uni);
                nA = __popc(col.padv); nH = __popc(col.phh); nM = __popc(col.pml); nD = __popc(col.pdp);
                {
#pragma unroll
                    for (int s = 0; s < NSLOT; ++s)
                    {   // T's values are kept: recomputing them in the write phase measured 4 % slower
                        col.tv[s] = t_value(col, s);
                        if (((uni >> s) & 1u) && col.tv[s] != 0.0) pT |= 1u << s;
                    }
                    if (pT != uni && !(p.skip & 1u)) raise_flag(p.flags, FLAG_T_CANCEL);
                }
            }
        }
    }

    STAMP(st, 3, 0);  // the column's arithmetic is done
    // ---- 2. packed block scan: T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10 bits ----
    const u64 mine = ((u64)nU | ((u64)nA << 11) | ((u64)nH << 22) | ((u64)nM << 33) | ((u64)nD << 43)) & p.keep;  // (matrices that are not materialised: TmParams.skip)
    u64 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) wave_tot[wid] = incl;
    if (tid < TM_NF) {  // the offsets fetched at the top travel through the scan's barrier
        s_prefix[tid] = pre_off;
        s_presum[tid] = pre_sum;
    }
    __syncthreads();
    u64 before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < TM_THREADS / 64; ++q) {
        const u64 v = wave_tot[q];
        if (q < wid) before += v;
        all += v;
    }
    const u64 excl = before + incl - mine;
    const unsigned ex[5] = {(unsigned)(excl & 0x7ff), (unsigned)((excl >> 11) & 0x7ff), (unsigned)((excl >> 22) & 0x7ff),
                            (unsigned)((excl >> 33) & 0x3ff), (unsigned)((excl >> 43) & 0x3ff)};
    const unsigned agg[5] = {(unsigned)(all & 0x7ff), (unsigned)((all >> 11) & 0x7ff), (unsigned)((all >> 22) & 0x7ff),
                             (unsigned)((all >> 33) & 0x3ff), (unsigned)((all >> 43) & 0x3ff)};

    // ---- 3. the tile's global offsets ----
    i64 g0[5];
#pragma unroll
    for (int m = 0; m < TM_NF; ++m) g0[m] = s_prefix[m];  // entries of matrix m before this tile (this launch)
    {
        // The space of this tile was reserved by the counting pass from the push mask.  A mask that does not
        // describe these ϕ / Lwet3D (stale, or not a makeindices result) would make the two passes disagree:
        // compare the tile's counts and write nothing on a mismatch.
        bool same = true;
#pragma unroll
        for (int m = 0; m < TM_NF; ++m) same &= s_presum[m] == agg[m];
        if (!same) {
            if (tid == 0) raise_flag(p.flags, FLAG_COUNT_MISMATCH);
            return;
        }
    }

    if (w0 + TM_THREADS >= p.n_own) {  // last tile: the closing colptr entry (and the totals, if no scan produced them)
        if (tid < TM_NF) {
            i64 tot = 0;
#pragma unroll
            for (int m = 0; m < TM_NF; ++m)
                if (m == tid) tot = g0[m] + agg[m];
            if (p.gsum) p.totals[tid] = tot;
            i64 *cp = (tid == 0) ? p.colptr[0] : (tid == 1) ? p.colptr[1] : (tid == 2) ? p.colptr[2] : (tid == 3) ? p.colptr[3] : p.colptr[4];
            const i64 nb = (tid == 0) ? p.nnz_base[0] : (tid == 1) ? p.nnz_base[1] : (tid == 2) ? p.nnz_base[2] : (tid == 3) ? p.nnz_base[3] : p.nnz_base[4];
            if (!((p.skip >> tid) & 1u)) cp[p.n_own] = nb + tot + 1;
        }
    }

    STAMP(st, 4, 0);  // the tile's offsets are known
    // ---- 4. write: colptr, then LDS-staged entries ----
    // Each wave stages the entries of ITS 64 columns in its own LDS region and streams them out itself:
    // a wave's columns are contiguous, so its entries of matrix m are one contiguous run starting at
    // g0[m] + (entries of the tile's earlier waves).  No workgroup barrier is needed in this phase.
    if (live) {
#pragma unroll
        for (int m = 0; m < TM_NF; ++m)
            if (!((p.skip >> m) & 1u)) p.colptr[m][w] = p.nnz_base[m] + g0[m] + ex[m] + 1;  // (non-temporal here: no gain)
    }
    // the vertical operators only ever hold the rows above, self and below (:438-479): lets the compiler drop
    // the other five slot tests of their staging loops
    const unsigned vslots = (1u << S_A) | (1u << S_SELF) | (1u << S_B);
    // matrices that are not materialised (T alone; a given operator): nothing of them is staged or stored
    const unsigned on0 = (p.skip & 1u) ? 0u : ~0u, on1 = (p.skip & 2u) ? 0u : ~0u, on2 = (p.skip & 4u) ? 0u : ~0u, on3 = (p.skip & 8u) ? 0u : ~0u,
                   on4 = (p.skip & 16u) ? 0u : ~0u;
    const unsigned pm[5] = {pT & on0, col.padv & on1, col.phh & on2, col.pml & vslots & on3, col.pdp & vslots & on4};
    // wave-uniform quantities go to scalar registers: the run's base pointers are then SGPR pairs, the stores
    // take the `global_store vaddr32, vdata, sbase` form and the copy loop is a scalar loop
    const u64 ubefore = (u64)wave_uniform((i64)before);
    const u64 wtot = ((u64)__builtin_amdgcn_readlane((unsigned)(incl >> 32), 63) << 32) | __builtin_amdgcn_readlane((unsigned)incl, 63);
    const unsigned wb[5] = {(unsigned)(ubefore & 0x7ff), (unsigned)((ubefore >> 11) & 0x7ff), (unsigned)((ubefore >> 22) & 0x7ff),
                            (unsigned)((ubefore >> 33) & 0x3ff), (unsigned)((ubefore >> 43) & 0x3ff)};
    const unsigned wc[5] = {(unsigned)(wtot & 0x7ff), (unsigned)((wtot >> 11) & 0x7ff), (unsigned)((wtot >> 22) & 0x7ff),
                            (unsigned)((wtot >> 33) & 0x3ff), (unsigned)((wtot >> 43) & 0x3ff)};
    typedef i64 i64x2 __attribute__((ext_vector_type(2)));
    // The matrices are written once and read by nobody on the device: NON-TEMPORAL stores, so that 1 GB of output per 0.44 GB of input
    // does not push the stencil's lines (south / north rows, levels above / below: re-read by later tiles) out of the L2.  Together with
    // the march order: HBM reads back to the algorithmic bytes (7.5 GB fetched at 0.25 degree instead of 15.4 GB), -8 % time.
#define TM_STORE(val, ptr) __builtin_nontemporal_store((val), (ptr))
    typedef i64x2 i64x2g __attribute__((aligned(8)));
    // rows and value bits are staged in two arrays: an entry is two 8-byte LDS writes straight from the registers
    // that hold it, a pair of entries one 16-byte LDS read per array
    i64 *my_row = s_stage + wid * TM_WSTAGE;
    i64 *my_val = my_row + TM_STAGE;  // a constant distance: one address register, the LDS offset field does the rest
#pragma unroll
    for (int m = 0; m < TM_NF; ++m) {
        // The run is streamed out with 16-byte stores (two entries per lane): 8-byte-per-lane stores are store-issue
        // bound per CU (measured: the write phase cost as much as loads + arithmetic).  The run starts at an arbitrary
        // 8-byte position; global_store_dwordx4 does not need more alignment than that, so pairs are simply counted from
        // the run's first entry (shifting the staging by the run's parity to keep the stores 16-byte aligned costs two more
        // 8-byte store instructions per matrix and array for the unpaired ends: +6 % time, tools/experiments/).
        const i64 run0 = wave_uniform(g0[m]) + wb[m];
        i64 *rv = p.rowval[m] + run0;
        double *nz = p.nzval[m] + run0;
        if (live) {
            const unsigned q0 = ex[m] - wb[m];
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                if ((pm[m] >> s) & 1u) {
                    const unsigned q = q0 + __popc(pm[m] & col.bef[s]);  // position inside the wave's run
                    const double v = (m == 0) ? col.tv[s] : (m == 1) ? col.adv[s] : (m == 2) ? col.hh[s] : (m == 3) ? col.ml[s] : col.dp[s];
                    my_row[q] = col.idx[s];
                    my_val[q] = __double_as_longlong(v);
                }
            }
            // T after an exact cancellation (rare): the column keeps its reserved (union) width, its entries are left-aligned
            // and the unused slots carry row 0 -- which is how the compaction (tfix_*) finds a column's real length, for
            // any step of an asynchronous pipeline, from the step's own output arrays
            if (m == 0 && on0 && (unsigned)__popc(pT) != nU) {
                for (unsigned e = __popc(pT); e < nU; ++e) { my_row[q0 + e] = 0; my_val[q0 + e] = 0; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const unsigned cnt = wc[m];
        bool room = true;
        if (p.cap[m] > 0) {  // callers that preallocate at an upper bound (0 = sized exactly by a plan)
            room = run0 + cnt <= p.cap[m];
            if (!room && lane == 0) raise_flag(p.flags, FLAG_CAPACITY);
        }
        if (room) {
            const unsigned end = cnt;  // staged entries occupy LDS indices [0, end)
            char *rvb = (char *)rv;
            char *nzb = (char *)nz;
            for (unsigned base = 0; base < end; base += 128) {  // full pairs
                const unsigned u = base + 2 * lane;
                if (u + 1 < end) {
                    TM_STORE(*(const i64x2 *)(my_row + u), (i64x2g *)(rvb + u * 8u));
                    TM_STORE(*(const i64x2 *)(my_val + u), (i64x2g *)(nzb + u * 8u));
                }
            }
            // an odd run's last entry: ONE 8-byte store instruction, lane 0 writes the row, lane 1 the value
            if ((lane < 2) & ((end & 1u) == 1u)) {
                const unsigned e = end - 1;
                i64 *dst = (lane == 0) ? (i64 *)(rvb + e * 8u) : (i64 *)(nzb + e * 8u);
                TM_STORE((lane == 0) ? my_row[e] : my_val[e], dst);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#ifdef OTMB_DBG_STAMPS
    STAMP(st, 5, 0);  // every store is issued
#ifndef OTMB_DBG_STAMPS_ORDER
    STAMP(st, 6, 1);  // ... and acknowledged
#endif
    if (p.status && lane == 0) {
        u64 *o = p.status + ((u64)tile * (TM_THREADS / 64) + wid) * OTMB_NSTAMP;
        for (int q = 0; q < 7; ++q) o[q] = st.t[q];
        // HW_REG_HW_ID (4): wave slot / SIMD / CU / SH / SE;  HW_REG_XCC_ID (20): which XCD (each XCD has its own s_memtime base)
        o[7] = (u64)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
               ((u64)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
        o[8] = rt_entry;
        u64 rt_end;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
        o[9] = rt_end;
    }
#endif
}

// ---- otmb_tm_args.given: the COMPARING pass ------------------------------------------------------------------------------------
// Is a given operator bit for bit what the fill pass would write?  One thread per column builds the column exactly as tm_kernel does
// (fast_column / build_column: the one copy of the arithmetic) and, for every operator m in g.check, reads the given matrix's column:
// same length, same rows in the same order (else: bit m of the verdict), same value BITS (-0.0 is not +0.0, a NaN equals itself; else: bit 8 + m --
// the derived PATTERN with other values, e.g. built with another κ: the fill pass can still read it).  The given arrays may be a depth
// slab's slice: column w holds entries [colptr[w] - colptr[0], colptr[w + 1] - colptr[0]) of rowval / nzval.  Nothing is stored but
// the verdict in flags[FLAG_GIVEN_MISMATCH].  Once per grid and κ (the verdict is cached), so plain wet-rank order, no staging.
struct GivenCmp {
    const i64 *cp[5], *ri[5], *vx[5];  // the given matrices' colptr / rowval / nzval (value bits)
    i64 nnz[5];
    unsigned check;
};
__global__ __launch_bounds__(TM_THREADS, TM_WAVES_PER_SIMD) void tm_given_kernel(const TmParams p, const GivenCmp g) {
    const int tid = threadIdx.x;
    const i64 w0 = (i64)blockIdx.x * TM_THREADS, w = w0 + tid;
    if (w0 >= p.n_own) return;
    const bool valid = w < p.n_own;
    const i64 wlast = (w0 + TM_THREADS - 1 < p.n_own) ? w0 + TM_THREADS - 1 : p.n_own - 1;
    const i64 wcl = valid ? w : wlast;
    const i64 L = p.lwet[wcl] - 1;
    const i64 Lnext = (wcl + 1 < p.n_own) ? p.lwet[wcl + 1] - 1 : p.G;
    const i64 Lmin = p.lwet[w0] - 1, Lmax = p.lwet[wlast] - 1;
    const i64 base_elem = (Lmin > p.P) ? Lmin - p.P : 0;
    const bool span_ok = (Lmax + p.P - base_elem) < (1ll << 28) && Lmin >= 0 && Lmax < p.G && Lmin <= Lmax;
    if (!valid) return;
    if (!span_ok || L < Lmin || L > Lmax || Lnext <= L) {  // not a makeindices result: nothing can be derived from it
        atomicOr(&p.flags[FLAG_GIVEN_MISMATCH], (int)g.check);
        return;
    }
    TileBase tb;
    tb.lw = (const char *)(p.lw + base_elem);
    tb.v = (const char *)(p.v + base_elem);
    tb.thk = (const char *)(p.thk + base_elem);
    tb.rho = p.rho ? (const char *)(p.rho + base_elem) : nullptr;
    tb.pt = (const char *)(p.phi[OTMB_TOP] + base_elem);
    tb.pe = (const char *)(p.phi[OTMB_EAST] + base_elem);
    tb.pw = (const char *)(p.phi[OTMB_WEST] + base_elem);
    tb.pn = (const char *)(p.phi[OTMB_NORTH] + base_elem);
    tb.ps = (const char *)(p.phi[OTMB_SOUTH] + base_elem);
    tb.pb = (const char *)(p.phi[OTMB_BOTTOM] + base_elem);
    tb.pu = tb.pv = tb.mk = nullptr;
    Column col;
    Stamps st;
    const i64 c = p.wet_base + w + 1;
    const Cell cell = cell_of(L, p.nx, p.ny, p.P);
    const unsigned oC = (unsigned)(L - base_elem) * 8u;
    const bool regular = (p.nx >= 3) && !(p.topo == OTMB_TRIPOLAR && cell.j == p.ny - 1);
    bool canonical;
    if (regular) canonical = fast_column<0>(p, tb, oC, cell.i, cell.j, cell.k, c, col, st);
    else {
        canonical = ldi(tb.lw, oC) == c;
        if (canonical) build_column(p, cell, c, col);
    }
    if (!canonical) {
        atomicOr(&p.flags[FLAG_GIVEN_MISMATCH], (int)g.check);
        return;
    }
    const unsigned vslots = (1u << S_A) | (1u << S_SELF) | (1u << S_B);
    const unsigned pm[5] = {0u, col.padv, col.phh, col.pml & vslots, col.pdp & vslots};
    unsigned bad = 0;  // bit m: the column's length or rows differ; bit 8 + m: only values do
#pragma unroll
    for (int m = 1; m < TM_NF; ++m) {
        if (!((g.check >> m) & 1u)) continue;
        const i64 c0 = g.cp[m][0];
        const i64 lo = g.cp[m][w] - c0, hi = g.cp[m][w + 1] - c0;
        bool ok = lo >= 0 && hi <= g.nnz[m] && hi - lo == (i64)__popc(pm[m]), same = true;
        if (w == p.n_own - 1) ok &= hi == g.nnz[m];
        if (ok) {
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) {
                if ((pm[m] >> sl) & 1u) {
                    const i64 q = lo + (i64)__popc(pm[m] & col.bef[sl]);
                    const double v = (m == 1) ? col.adv[sl] : (m == 2) ? col.hh[sl] : (m == 3) ? col.ml[sl] : col.dp[sl];
                    ok &= g.ri[m][q] == col.idx[sl];
                    same &= g.vx[m][q] == __double_as_longlong(v);
                }
            }
        }
        if (!ok) bad |= 1u << m;
        else if (!same) bad |= 0x100u << m;
    }
    if (bad) atomicOr(&p.flags[FLAG_GIVEN_MISMATCH], (int)bad);
}

// closing colptr entry of each matrix: nnz_base + nnz + 1 (values known on the host since the plan)
__global__ void tm_finish_colptr(i64 *c0, i64 *c1, i64 *c2, i64 *c3, i64 *c4, i64 N, i64 t0, i64 t1, i64 t2, i64 t3, i64 t4) {
    if (threadIdx.x == 0) {
        if (c0) c0[N] = t0 + 1;
        if (c1) c1[N] = t1 + 1;
        if (c2) c2[N] = t2 + 1;
        if (c3) c3[N] = t3 + 1;
        if (c4) c4[N] = t4 + 1;
    }
}

// ---- rare path: T had exact-zero sums, so its columns were written left-aligned in slots reserved for the
// union pattern.  Compact: per-column actual counts (tcount) -> scan -> move.  One thread per column.
#define TFIX_THREADS 256
#define TFIX_PER 4
__global__ __launch_bounds__(TFIX_THREADS) void tfix_derive(const i64 *__restrict__ colptr, const i64 *__restrict__ rowval, i64 n, i64 nnz_base,
                                                            uint8_t *__restrict__ tcount) {
    const i64 c = (i64)blockIdx.x * TFIX_THREADS + threadIdx.x;
    if (c >= n) return;
    const i64 lo = colptr[c] - 1 - nnz_base, hi = colptr[c + 1] - 1 - nnz_base;
    unsigned cnt = 0;
    for (i64 e = lo; e < hi && e < lo + TM_MAXROWS; ++e) cnt += rowval[e] != 0;  // (rows are 1-based: 0 marks an unused slot)
    tcount[c] = (uint8_t)cnt;
}
__global__ __launch_bounds__(TFIX_THREADS) void tfix_count(const uint8_t *__restrict__ tcount, i64 n, uint32_t *tilesums) {
    __shared__ unsigned part[TFIX_THREADS / 64];
    unsigned x = 0;
    for (int q = 0; q < TFIX_PER; ++q) {
        const i64 c = ((i64)blockIdx.x * TFIX_PER + q) * TFIX_THREADS + threadIdx.x;
        if (c < n) x += tcount[c];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) tilesums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(TFIX_THREADS) void tfix_move(const uint8_t *__restrict__ tcount, i64 n, const i64 *__restrict__ tileoffs,
                                                          const i64 *__restrict__ old_colptr, const i64 *__restrict__ old_row,
                                                          const double *__restrict__ old_val, i64 nnz_base, i64 *new_colptr,
                                                          i64 *new_row, double *new_val) {
    __shared__ unsigned wave_tot[TFIX_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    i64 run = tileoffs[blockIdx.x];
    for (int q = 0; q < TFIX_PER; ++q) {
        const i64 c = ((i64)blockIdx.x * TFIX_PER + q) * TFIX_THREADS + tid;
        const unsigned mine = (c < n) ? tcount[c] : 0;
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < TFIX_THREADS / 64; ++w) {
            const unsigned v = wave_tot[w];
            if (w < wid) before += v;
            all += v;
        }
        __syncthreads();
        if (c < n) {
            const i64 dst = run + before + incl - mine;  // entries before this column (this launch)
            const i64 src = old_colptr[c] - 1 - nnz_base;
            new_colptr[c] = nnz_base + dst + 1;
            for (unsigned e = 0; e < mine; ++e) {
                new_row[dst + e] = old_row[src + e];
                new_val[dst + e] = old_val[src + e];
            }
        }
        run += all;
    }
}

// ---- march order of the fill pass's tiles (otmb_ctx_set_tile_order) -----------------------------------------------
// Tiles are 256 consecutive wet columns, i.e. pieces of one level's rows.  In wet-rank order a tile's vertical
// neighbours (levels k-1 and k+1 of Lwet3D, v3D, ρ) were touched one whole LEVEL of traffic earlier -- 124 MB of inputs
// plus 300 MB of outputs on a 1440x1080 grid, past every cache -- so they come from HBM three times.  In march order the
// tiles of a band of R rows are taken level after level: the same lines are needed again a few tiles later and are
// served by the L2 / Infinity Cache.  Bucket = (band, block of columns, level) -- one block per band unless the rows are longer than
// OTMB_MARCH_AUTO_COLS cells; a tile belongs to the block its first cell lies in -- ; a counting sort of the tiles by bucket.  Speed only.
__device__ __forceinline__ unsigned order_key(const i64 *__restrict__ lwet, i64 t, i64 n, int nx, int ny, i64 P, int rows, int nz, int topo, int cols) {
    const i64 L = lwet[t * TM_THREADS] - 1;  // (whatever Lwet holds, the key stays inside the bucket table)
    i64 k = L / P, j = (L - k * P) / nx;
    i64 ic = (L - k * P - j * nx) / cols;
    const i64 nblk = (nx + cols - 1) / cols;
    ic = ic < 0 ? 0 : (ic >= nblk ? nblk - 1 : ic);
    k = k < 0 ? 0 : (k >= nz ? nz - 1 : k);
    j = j < 0 ? 0 : (j >= ny ? ny - 1 : j);
    // HEAVY tiles -- bucket 0, the front of the sequence, dealt over the XCDs by xcd_position: tiles with cells on the tripolar
    // seam row (generic column builder, waves live about twice as long).  Lwet ascends, so the tile's cells lie between its
    // first and its last entry in (level, row) order: it touches row ny - 1 iff it starts there, ends there or runs into the next level.
    if (topo == OTMB_TRIPOLAR && nx >= 3) {
        const i64 wl = (t * TM_THREADS + TM_THREADS - 1 < n) ? t * TM_THREADS + TM_THREADS - 1 : n - 1;
        const i64 L1 = lwet[wl] - 1;
        i64 k1 = L1 / P, j1 = (L1 - k1 * P) / nx;
        if (j == ny - 1 || j1 == ny - 1 || k1 > k) return 0u;
    }
    // bands from north to south (the seam row's neighbours at the START of an XCD's eighth: -5 % at 1 degree against south first)
    return 1u + ((unsigned)((ny - 1 - j) / rows) * (unsigned)nblk + (unsigned)ic) * (unsigned)nz + (unsigned)k;
}
__global__ void order_hist(const i64 *__restrict__ lwet, i64 ntiles, i64 n, int nx, int ny, i64 P, int rows, int nz, int topo, int cols, unsigned *hist) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntiles) atomicAdd(&hist[order_key(lwet, t, n, nx, ny, P, rows, nz, topo, cols)], 1u);
}
__global__ __launch_bounds__(1024) void order_scan(unsigned *hist, i64 nbuckets) {  // in place: exclusive prefix
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (i64 b0 = 0; b0 < nbuckets; b0 += 1024) {
        const i64 b = b0 + tid;
        const unsigned mine = (b < nbuckets) ? hist[b] : 0u;
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        unsigned before = carry;
        for (int q = 0; q < wid; ++q) before += wave_tot[q];
        if (b < nbuckets) hist[b] = before + incl - mine;
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
}
// every tile takes the next free position of its bucket: a bijection whatever the keys are (the order inside a bucket
// -- a few dozen neighbouring tiles -- is left to the atomics)
__global__ void order_scatter(const i64 *__restrict__ lwet, i64 ntiles, i64 n, int nx, int ny, i64 P, int rows, int nz, int topo, int cols,
                              unsigned *cursor, unsigned *order) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntiles) order[atomicAdd(&cursor[order_key(lwet, t, n, nx, ny, P, rows, nz, topo, cols)], 1u)] = (unsigned)t;
}

// Decide and build the tile order of a fill launch.  Returns the device pointer (or NULL: wet-rank order).
static int32_t build_tile_order(otmb_ctx *ctx, const otmb_tm_args &a, i64 ntiles, TmParams &p) {
    p.order = nullptr;
    p.nt_order = (unsigned)ntiles;
    p.nheavy = 0;
    int rows = ctx->march_rows;
    if (rows < 0) rows = OTMB_MARCH_AUTO_ROWS;
    if (rows <= 0 || ntiles < 64 || ntiles >= (1ll << 31) - 16) return OTMB_OK;
    if (rows > a.ny) rows = (int)a.ny;
    // blocks of columns: equal pieces of a row, none longer than the limit (whole rows when they are short enough)
    int cols = ctx->march_cols;
    if (cols < 0) cols = OTMB_MARCH_AUTO_COLS;
    if (cols <= 0 || cols >= a.nx) cols = (int)a.nx;
    else { const i64 nb_ = (a.nx + cols - 1) / cols; cols = (int)((a.nx + nb_ - 1) / nb_); }
    const i64 nblk = (a.nx + cols - 1) / cols;
    const i64 nbands = (a.ny + rows - 1) / rows, nbuckets = nbands * nblk * a.nz + 1;
    if (nbuckets >= (1ll << 31)) return OTMB_OK;
    const size_t ob = ((size_t)ntiles * sizeof(unsigned) + 255) / 256 * 256, bb = ((size_t)nbuckets * sizeof(unsigned) + 255) / 256 * 256;
    // the order is a function of the grid alone: computed once per (Lwet array, shape, band height) and kept.  (Any permutation of
    // the tiles is correct, so an Lwet array rewritten in place can only cost speed.)
    otmb_ctx::OrderKey key;
    key.lwet = a.lwet; key.n = a.n_wet; key.nx = a.nx; key.ny = a.ny; key.nz = a.nz; key.rows = rows; key.topo = a.topology; key.cols = cols;
    if (ctx->order.p && ctx->order.cap >= ob + bb && ctx->order_key == key) {
        p.order = (const unsigned *)ctx->order.p;
        p.nheavy = ctx->deal_heavy ? ctx->order_nheavy : 0u;
        return OTMB_OK;
    }
    int32_t rc;
    // (a failed or half-enqueued build must not be trusted by the next call: the key is recorded only once all three kernels are
    // enqueued without an error, on the stream they were enqueued on -- otmb_ctx_set_stream forgets the key, so a fill on another
    // stream can never read a permutation that is still being built)
    ctx->order_key = otmb_ctx::OrderKey();
    if ((rc = otmb_reserve(ctx, ctx->order, ob + bb))) return rc;
    unsigned *order = (unsigned *)ctx->order.p, *hist = (unsigned *)((char *)ctx->order.p + ob);
    {
        KernelTimer kt(ctx, K_TM_ORDER);
        HIP_TRY(ctx, hipMemsetAsync(hist, 0, bb, ctx->stream));
        const unsigned nb = (unsigned)((ntiles + 255) / 256);
        hipLaunchKernelGGL(order_hist, dim3(nb), dim3(256), 0, ctx->stream, (const i64 *)a.lwet, ntiles, (i64)a.n_wet, (int)a.nx, (int)a.ny, a.nx * a.ny,
                           rows, (int)a.nz, (int)a.topology, cols, hist);
        hipLaunchKernelGGL(order_scan, dim3(1), dim3(1024), 0, ctx->stream, hist, nbuckets);
        // the number of heavy tiles = the exclusive prefix at bucket 1: the host needs it (grid size, kernel argument), once per grid
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot + 14, hist + 1, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        hipLaunchKernelGGL(order_scatter, dim3(nb), dim3(256), 0, ctx->stream, (const i64 *)a.lwet, ntiles, (i64)a.n_wet, (int)a.nx, (int)a.ny, a.nx * a.ny,
                           rows, (int)a.nz, (int)a.topology, cols, hist, order);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned nh = *(const unsigned *)(ctx->h_tot + 14);
    if (nh > (unsigned)ntiles) nh = 0;  // (cannot happen; any value <= ntiles is a correct mapping)
    ctx->order_nheavy = nh;
    ctx->order_key = key;
    p.order = order;
    p.nheavy = ctx->deal_heavy ? nh : 0u;
    return OTMB_OK;
}


// ---- host side ------------------------------------------------------------------------------
// One launch site for the fill pass's instantiations: FUSED (the fused step's flux re-derivation) x GIVEN (a given TκH / TκVdeep read where it lies).
template <int GIVEN> static void launch_fill_given(otmb_ctx *ctx, const TmParams &p, int fused, dim3 grid, dim3 block) {
    if (fused == 1) hipLaunchKernelGGL((tm_kernel<1, GIVEN>), grid, block, 0, ctx->stream, p);
    else if (fused == 2) hipLaunchKernelGGL((tm_kernel<2, GIVEN>), grid, block, 0, ctx->stream, p);
    else hipLaunchKernelGGL((tm_kernel<0, GIVEN>), grid, block, 0, ctx->stream, p);
}
static void launch_fill(otmb_ctx *ctx, const TmParams &p, int fused) {
    static const bool env_read = [] { const char *e = getenv("OTMB_GIVEN_READ"); return !(e && e[0] == '0'); }();
    // (a derived TκH: reading is a choice -- regular cells only, OTMB_GIVEN_READ=0 re-derives; the derived rows with other values: it is the only way)
    const bool hread = p.hcp != nullptr && (p.hmust || (env_read && p.nx >= 3)), dread = p.dcp != nullptr;
    const dim3 grid(xcd_grid(p.nt_order, p.nheavy)), block(TM_THREADS);
    if (hread && dread) launch_fill_given<3>(ctx, p, fused, grid, block);
    else if (dread) launch_fill_given<2>(ctx, p, fused, grid, block);
    else if (hread) launch_fill_given<1>(ctx, p, fused, grid, block);
    else launch_fill_given<0>(ctx, p, fused, grid, block);
}
static void fill_params(TmParams &p, const otmb_tm_args &a, otmb_ctx *ctx, const TmPlan *pl) {
    memset(&p, 0, sizeof p);
    for (int f = 0; f < 6; ++f) p.phi[f] = a.phi[f];
    p.v = a.v3d; p.thk = a.thkcello; p.rho = a.rho; p.rho_s = a.rho_scalar;
    p.lw = (const i64 *)a.lwet3d; p.lwet = (const i64 *)a.lwet;
    for (int d = 0; d < 4; ++d) { p.edge[d] = a.edge_length[d]; p.dist[d] = a.dist_nbr[d]; }
    p.area = a.area2d; p.zt = a.zt; p.ml = a.mlotst;
    p.kH = a.kappa_h; p.kML = a.kappa_vml; p.kDeep = a.kappa_vdeep;
    p.nx = (int)a.nx; p.ny = (int)a.ny; p.nz = (int)a.nz; p.topo = a.topology; p.upwind = a.upwind;
    p.skip = pl ? pl->skip : ((a.only_t ? 0x1eu : 0u) | ((unsigned)a.skip_ops & 0x1fu));
    p.hcp = nullptr; p.hx = nullptr; p.hnnz = 0;
    p.dcp = nullptr; p.dx = nullptr; p.dnnz = 0;
    if (pl && ((pl->read >> OTMB_TKVDEEP) & 1u) && a.given[OTMB_TKVDEEP].nnz > 0) {
        p.dcp = (const i64 *)a.given[OTMB_TKVDEEP].colptr; p.dx = a.given[OTMB_TKVDEEP].nzval; p.dnnz = a.given[OTMB_TKVDEEP].nnz;
    }
    if (pl && ((pl->derived >> OTMB_TKH) & 1u) && a.given[OTMB_TKH].nnz > 0) {  // (read by the HREAD fill kernels only)
        p.hcp = (const i64 *)a.given[OTMB_TKH].colptr; p.hx = a.given[OTMB_TKH].nzval; p.hnnz = a.given[OTMB_TKH].nnz;
        p.hmust = (int)((pl->read >> OTMB_TKH) & 1u);
    }
    p.keep = keep_mask(p.skip);
    p.P = a.nx * a.ny; p.G = p.P * a.nz;
    p.n_own = a.n_wet;
    if (pl) {
        p.wet_base = pl->wet_base;
        for (int m = 0; m < 5; ++m) p.nnz_base[m] = pl->nnz_base[m];
    }
    p.tilesums = (uint32_t *)ctx->tm_sums.p;
    p.tileoffs = (const i64 *)ctx->tm_offs.p;
    p.flags = (int *)ctx->flags.p;
    p.count_order = ctx->count_order;
    p.nt_order = (unsigned)((a.n_wet + TM_THREADS - 1) / TM_THREADS);
    p.nheavy = 0;
}

// The counting pass reads the push mask: the caller's (written by facefluxes for exactly these ϕ), or one derived
// here from ϕ and Lwet3D.
static int32_t ensure_push_mask(otmb_ctx *ctx, const otmb_tm_args &a, TmParams &p) {
    // (the mask argument of a counting facefluxes call was not written by it: never a counting pass's input)
    if (a.push_mask && a.push_mask != ctx->ffc_partial_mask) {
        p.mask = a.push_mask;
        return OTMB_OK;
    }
    int32_t rc;
    if ((rc = otmb_reserve(ctx, ctx->mask, (size_t)p.G * sizeof(uint16_t) + 16))) return rc;
    p.mask = (const uint16_t *)ctx->mask.p;
    return otmb_launch_push_mask(ctx, a.phi, a.lwet3d, 0, p.G, (uint16_t *)ctx->mask.p);
}

// Counts in facefluxes: do the tile counts that the last facefluxes call on this context accumulated describe exactly this
// transportmatrix?  (Same ϕ arrays, the mask pointer that call was given, mixed-layer inputs, indices, weighting; whole grid.)
// Returns the buffer that holds them, or -1.
static int ffc_match(const otmb_ctx *ctx, const otmb_tm_args &a, const TmPlan &pl) {
    const otmb_ctx::FfCountsKey &k = ctx->ffc;
    // (depth slab: the transportmatrix names the extended local grid's arrays, facefluxes wrote their owned levels, k_own0 levels in)
    const i64 off = k.k_own0 * a.nx * a.ny;
    if (!k.valid || k.gen != ctx->ff_gen || !a.push_mask || a.push_mask + off != k.mask || pl.wet_base != k.wet_base) return -1;
    for (int f = 0; f < 6; ++f)
        if (a.phi[f] + off != k.phi[f]) return -1;
    if (a.mlotst != k.mlotst || a.zt != k.zt || a.lwet3d != k.lwet3d || a.nx != k.nx || a.ny != k.ny || a.nz != k.nz ||
        a.n_wet != k.n_wet || a.topology != k.topo || (a.upwind != 0) != (k.upwind != 0) || (a.only_t != 0) != (k.only_t != 0))
        return -1;
    return k.buf;
}
// ... then the scan takes them (and leaves the buffer zeroed for the facefluxes call after next)
static void ffc_consume(otmb_ctx *ctx, int buf, const TmParams &p, i64 *offs, i64 *dtot, i64 *gsum, i64 ntiles, bool all_levels) {
    KernelTimer kt(ctx, K_TILESCAN);
    otmb_launch_tilescan_packed(ctx->stream, (unsigned long long *)ctx->ffc_sums[buf].p, (const unsigned long long *)ctx->ffc.stat, p.tilesums, offs, dtot, gsum, ntiles, p.flags,
                                p.keep, all_levels);
    ctx->ffc.valid = false;
    ctx->ffc_dirty[buf] = false;
}
// ρ on the wet cells (:233), for the one case where no pass has read ρ before an error must be ranked (see otmb_transportmatrix_plan_dev)
__global__ __launch_bounds__(256) void rho_nan_kernel(const double *__restrict__ rho, const i64 *__restrict__ lwet, i64 n, int *flags) {
    const i64 w = (i64)blockIdx.x * 256 + threadIdx.x;
    if (w < n && isnan(rho[lwet[w] - 1])) raise_flag(flags, FLAG_RHO_NAN);
}

// ---- otmb_tm_args.given (host side) ---------------------------------------------------------------------------------------------
static bool verdict_matches(const otmb_ctx::GivenVerdict &v, const otmb_ctx *ctx, const otmb_tm_args &a, const TmPlan &pl, int m) {
    if (!v.valid || v.epoch != ctx->given_epoch) return false;
    const otmb_csc &g = a.given[m];
    if (v.g.colptr != g.colptr || v.g.rowval != g.rowval || v.g.nzval != g.nzval || v.g.nnz != g.nnz) return false;
    if (v.lwet3d != a.lwet3d || v.lwet != a.lwet || v.v3d != a.v3d || v.nx != a.nx || v.ny != a.ny || v.nz != a.nz || v.n_wet != a.n_wet ||
        v.wet_base != pl.wet_base || v.topo != a.topology)
        return false;
    if (m == OTMB_TKH) {
        if (v.thk != a.thkcello || v.kappa != a.kappa_h) return false;
        for (int d = 0; d < 4; ++d)
            if (v.edge[d] != a.edge_length[d] || v.dist[d] != a.dist_nbr[d]) return false;
    } else {
        if (v.area != a.area2d || v.zt != a.zt || v.kappa != a.kappa_vdeep) return false;
    }
    return true;
}
static void verdict_store(otmb_ctx *ctx, const otmb_tm_args &a, const TmPlan &pl, int m, bool derived, bool pattern) {
    otmb_ctx::GivenVerdict &v = ctx->given_verdict[m];
    v.valid = true; v.derived = derived; v.pattern = pattern; v.epoch = ctx->given_epoch; v.g = a.given[m];
    v.lwet3d = a.lwet3d; v.lwet = a.lwet; v.v3d = a.v3d; v.thk = a.thkcello; v.area = a.area2d; v.zt = a.zt;
    for (int d = 0; d < 4; ++d) { v.edge[d] = a.edge_length[d]; v.dist[d] = a.dist_nbr[d]; }
    v.nx = a.nx; v.ny = a.ny; v.nz = a.nz; v.n_wet = a.n_wet; v.wet_base = pl.wet_base; v.topo = a.topology;
    v.kappa = (m == OTMB_TKH) ? a.kappa_h : a.kappa_vdeep;
}
// the comparing pass over the operators in `check`; *derived: those that are bit for bit what the fill pass writes; *pattern: those with
// exactly its rows and other values.  Synchronises.
static int32_t verify_given(otmb_ctx *ctx, const otmb_tm_args &a, const TmPlan &pl, unsigned check, unsigned *derived, unsigned *pattern) {
    *derived = *pattern = 0;
    if (a.n_wet == 0) {  // a 0 x 0 matrix: derived iff it is empty
        for (int m = 1; m < 5; ++m)
            if (((check >> m) & 1u) && a.given[m].nnz == 0) *derived |= 1u << m;
        return OTMB_OK;
    }
    TmPlan tmp;
    tmp.wet_base = pl.wet_base;
    tmp.skip = 0;
    TmParams p;
    fill_params(p, a, ctx, &tmp);
    // TκH / TκVdeep do not look at the fluxes: the six ϕ pointers name v3D (G readable Float64), so that this pass can run for callers
    // whose ϕ arrays do not exist (otmb_step_dev) or are about to be overwritten
    for (int f = 0; f < 6; ++f) p.phi[f] = a.v3d;
    int *dflags = (int *)ctx->flags.p;
    p.flags = dflags;
    GivenCmp g;
    memset(&g, 0, sizeof g);
    g.check = check;
    for (int m = 1; m < 5; ++m) {
        g.cp[m] = (const i64 *)a.given[m].colptr; g.ri[m] = (const i64 *)a.given[m].rowval; g.vx[m] = (const i64 *)a.given[m].nzval;
        g.nnz[m] = a.given[m].nnz;
    }
    HIP_TRY(ctx, hipMemsetAsync(dflags, 0, OTMB_TM_STATE_BYTES, ctx->stream));
    const i64 ntiles = (a.n_wet + TM_THREADS - 1) / TM_THREADS;
    hipLaunchKernelGGL(tm_given_kernel, dim3((unsigned)ntiles), dim3(TM_THREADS), 0, ctx->stream, p, g);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_NFLAGS_TM * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(dflags, 0, OTMB_TM_STATE_BYTES, ctx->stream));  // (whatever the columns' arithmetic flagged is the real pass's to report)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned differs = (unsigned)ctx->h_flags[FLAG_GIVEN_MISMATCH], rows = differs & 0xffu, values = (differs >> 8) & 0xffu;
    *derived = check & ~rows & ~values;
    *pattern = check & ~rows & values;
    ctx->given_checks += 1;
    return OTMB_OK;
}
// Which operators does the caller pass, and how is each treated?  Sets pl.given / derived / read / foreign / skip and ctx->given_state.
static int32_t classify_given(otmb_ctx *ctx, const otmb_tm_args &a, TmPlan &pl) {
    // (OTMB_GIVEN_PATTERN=0: an operator with the derived rows and other values is treated as any foreign matrix -- A/B, tests of the sparse-add path)
    static const bool env_pattern = [] { const char *e = getenv("OTMB_GIVEN_PATTERN"); return !(e && e[0] == '0'); }();
    pl.given = given_mask(a);
    pl.derived = pl.foreign = pl.read = 0;
    for (int m = 0; m < 5; ++m) { ctx->given_state[m] = 0; pl.built_nnz[m] = 0; }
    pl.skip = (a.only_t ? 0x1eu : 0u) | ((unsigned)a.skip_ops & 0x1fu);
    pl.want_t = !(pl.skip & 1u);
    if (a.given[OTMB_T].colptr) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "given[OTMB_T]: T is never passed in (src/matrixbuilding.jl:133-138)");
    if (!pl.given) return OTMB_OK;
    unsigned check = 0;
    for (int m = 1; m < 5; ++m) {
        if (!((pl.given >> m) & 1u)) continue;
        const otmb_csc &g = a.given[m];
        if (g.nnz < 0 || (g.nnz > 0 && (!g.rowval || !g.nzval))) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "given operator: arrays / nnz");
        if (m == OTMB_TKH || m == OTMB_TKVDEEP) {  // functions of the grid and κ alone: worth a verdict that is kept
            if (verdict_matches(ctx->given_verdict[m], ctx, a, pl, m)) {
                if (ctx->given_verdict[m].derived) pl.derived |= 1u << m;
                if (ctx->given_verdict[m].pattern) pl.read |= 1u << m;
            } else {
                check |= 1u << m;
            }
        }
    }
    if (check) {
        unsigned d = 0, pt = 0;
        int32_t rc;
        if ((rc = verify_given(ctx, a, pl, check, &d, &pt))) return rc;
        for (int m = 1; m < 5; ++m)
            if ((check >> m) & 1u) verdict_store(ctx, a, pl, m, (d >> m) & 1u, (pt >> m) & 1u);
        pl.derived |= d;
        pl.read |= pt;
    }
    // the derived rows with other values: not materialised either -- the fill pass reads the values where they lie
    if (!env_pattern) pl.read = 0;
    pl.derived |= pl.read;
    pl.foreign = pl.given & ~pl.derived;
    if (pl.foreign && pl.want_t && (pl.skip & 0x1eu & ~pl.given))
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "only_t / skip_ops with a foreign given operator: T is then a sum of materialised matrices");
    // nothing given is built; with a foreign operand T is not the kernel's business either (it is the device sparse add of the four)
    pl.skip |= pl.given | (pl.foreign ? 1u : 0u);
    for (int m = 1; m < 5; ++m)
        if ((pl.given >> m) & 1u) ctx->given_state[m] = ((pl.read >> m) & 1u) ? 3 : (((pl.derived >> m) & 1u) ? 1 : 2);
    return OTMB_OK;
}

// ignore: otmb_tm_args.ignore_ops -- errors that only an operator the caller already has would have raised
static int32_t check_flags(otmb_ctx *ctx, const int *f = nullptr, int ignore = 0) {
    if (!f) f = ctx->h_flags;
    const bool iA = (ignore >> OTMB_TADV) & 1, iH = (ignore >> OTMB_TKH) & 1, iM = (ignore >> OTMB_TKVML) & 1, iD = (ignore >> OTMB_TKVDEEP) & 1;
    if (f[FLAG_NONCANONICAL]) return otmb_fail(ctx, OTMB_ERR_NONCANONICAL_INDICES);
    if (f[FLAG_COUNT_MISMATCH]) return otmb_fail(ctx, OTMB_ERR_PUSH_MASK);
    if (f[FLAG_RHO_NAN] && !iA) return otmb_fail(ctx, OTMB_ERR_RHO_NAN);  // reference order: :233, loop, :39, :61, :90, :114
    if (f[FLAG_FLUX_INTO_LAND] && !iA) return otmb_fail(ctx, OTMB_ERR_FLUX_INTO_LAND);
    if (f[FLAG_TADV_NAN] && !iA) return otmb_fail(ctx, OTMB_ERR_TADV_NAN);
    if (f[FLAG_TKH_NAN] && !iH) return otmb_fail(ctx, OTMB_ERR_TKH_NAN);
    if (f[FLAG_TKVML_NAN] && !iM) return otmb_fail(ctx, OTMB_ERR_TKVML_NAN);
    if (f[FLAG_TKVDEEP_NAN] && !iD) return otmb_fail(ctx, OTMB_ERR_TKVDEEP_NAN);
    if (f[FLAG_CAPACITY]) return otmb_fail(ctx, OTMB_ERR_CAPACITY);
    return OTMB_OK;
}

static int32_t validate_args(otmb_ctx *ctx, const otmb_tm_args *a, bool top_only = false) {
    if (a->nx < 1 || a->ny < 1 || a->nz < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    const i64 G = a->nx * a->ny * a->nz;
    if (a->nx * a->ny >= (1ll << 27) || G >= (1ll << 32)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid too large");
    if (a->topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);
    if (a->topology != OTMB_BIPOLAR && a->topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "topology");
    for (int f = 0; f < 6; ++f)
        if (!a->phi[f] && !(top_only && f != OTMB_TOP)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "phi");
    for (int d = 0; d < 4; ++d)
        if (!a->edge_length[d] || !a->dist_nbr[d]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "metrics");
    if (!a->v3d || !a->thkcello || !a->lwet3d || !a->area2d || !a->zt || !a->mlotst)
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null input array");
    if (a->n_wet < 0 || a->n_wet > G || (a->n_wet > 0 && !a->lwet)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "lwet / n_wet");
    if (!a->rho && a->rho_scalar != a->rho_scalar && a->n_wet > 0 && !((a->ignore_ops >> OTMB_TADV) & 1) && !a->given[OTMB_TADV].colptr)
        return otmb_fail(ctx, OTMB_ERR_RHO_NAN);  // :233 (buildTadv's check: not reached when Tadv is passed in, :140)
    return OTMB_OK;
}


// After a fill launch has completed and flagged FLAG_T_CANCEL: compact T in place (through temporaries).  n columns whose
// reserved (union-pattern) entries number `reserved`; *actual receives the final nnz.  The stream is idle on entry.
static int32_t t_fixup(otmb_ctx *ctx, i64 n, i64 nnz_base, i64 reserved, i64 *colptrT, i64 *rowvalT, double *nzvalT, i64 *actual_out) {
    *actual_out = reserved;
    if (n == 0) return OTMB_OK;
    const i64 per = (i64)TFIX_THREADS * TFIX_PER;
    const i64 nt = (n + per - 1) / per;
    int32_t rc;
    if ((rc = otmb_reserve(ctx, ctx->tcount, (size_t)n + 16))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->blocksums, (size_t)(nt + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->blockoffs, (size_t)(nt + 1) * sizeof(i64) + otmb_scan_scratch(nt, 1)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->tfix[0], (size_t)(n + 1) * sizeof(i64)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->tfix[1], (size_t)(reserved + 1) * sizeof(i64)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->tfix[2], (size_t)(reserved + 1) * sizeof(double)))) return rc;
    int *dflags = (int *)ctx->flags.p;
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS) + 8;
    uint8_t *tc = (uint8_t *)ctx->tcount.p;
    hipLaunchKernelGGL(tfix_derive, dim3((unsigned)((n + TFIX_THREADS - 1) / TFIX_THREADS)), dim3(TFIX_THREADS), 0, ctx->stream,
                       (const i64 *)colptrT, (const i64 *)rowvalT, n, nnz_base, tc);
    hipLaunchKernelGGL(tfix_count, dim3((unsigned)nt), dim3(TFIX_THREADS), 0, ctx->stream, (const uint8_t *)tc, n, (uint32_t *)ctx->blocksums.p);
    otmb_launch_tilescan(ctx->stream, (const uint32_t *)ctx->blocksums.p, (i64 *)ctx->blockoffs.p, dtot, nt, 1,
                         (i64 *)ctx->blockoffs.p + (nt + 1));
    hipLaunchKernelGGL(tfix_move, dim3((unsigned)nt), dim3(TFIX_THREADS), 0, ctx->stream, (const uint8_t *)tc, n, (const i64 *)ctx->blockoffs.p,
                       (const i64 *)colptrT, (const i64 *)rowvalT, (const double *)nzvalT, nnz_base,
                       (i64 *)ctx->tfix[0].p, (i64 *)ctx->tfix[1].p, (double *)ctx->tfix[2].p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot + 8, dtot, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const i64 actual = ctx->h_tot[8];
    HIP_TRY(ctx, hipMemcpyAsync(colptrT, ctx->tfix[0].p, (size_t)n * sizeof(i64), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(rowvalT, ctx->tfix[1].p, (size_t)actual * sizeof(i64), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(nzvalT, ctx->tfix[2].p, (size_t)actual * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    const i64 last = nnz_base + actual + 1;
    HIP_TRY(ctx, hipMemcpyAsync(colptrT + n, &last, sizeof(i64), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *actual_out = actual;
    return OTMB_OK;
}

// The state blocks of the asynchronous steps, device ring -> pinned host mirror, once the stream has drained.
static int32_t fetch_ring(otmb_ctx *ctx) {
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_ring, ctx->ring.p, (size_t)OTMB_RING * OTMB_TM_STATE_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return OTMB_OK;
}

// Fold the completed pending steps [tm_first, tm_next) of the asynchronous protocol: every step's verdict and nnz go to
// ctx->tm_hist (otmb_transportmatrix_result_step), the first failing step into the sticky (status, step) pair, and a step
// whose T had exact cancellations is compacted in ITS OWN output arrays (every otmb_transportmatrix_dev call recorded
// them) -- unless a later pending step was given the same arrays, which then hold that later step's matrix.
// The stream is idle on entry (the steps' state blocks have landed in h_ring).
static int32_t fold_pending(otmb_ctx *ctx) {
    int32_t ret = OTMB_OK;
    for (i64 s = ctx->tm_first; s < ctx->tm_next; ++s) {
        const int *f = otmb_ring_tm(ctx->h_ring, s);
        const i64 *tot = (const i64 *)(f + OTMB_NFLAGS);
        otmb_ctx::TmStepResult r;
        const size_t q = (size_t)(s - ctx->tm_first);
        r.status = check_flags(ctx, f, q < ctx->tm_rec.size() ? ctx->tm_rec[q].ignore_ops : 0);  // sets ctx->err
        for (int m = 0; m < 5; ++m) r.nnz[m] = tot[m];
        if (r.status && !ctx->tm_sticky) { ctx->tm_sticky = r.status; ctx->tm_sticky_step = s; ctx->tm_sticky_msg = ctx->err; }
        if (!r.status && f[FLAG_T_CANCEL] && q < ctx->tm_rec.size()) {
            const otmb_ctx::TmStepRec &rec = ctx->tm_rec[q];
            bool superseded = false;
            for (size_t l = q + 1; l < ctx->tm_rec.size(); ++l)
                superseded |= ctx->tm_rec[l].colptrT == rec.colptrT || ctx->tm_rec[l].rowvalT == rec.rowvalT || ctx->tm_rec[l].nzvalT == rec.nzvalT;
            if (!superseded && !ret) {
                i64 actual = r.nnz[0];
                ret = t_fixup(ctx, rec.n_wet, rec.nnz_base0, r.nnz[0], (i64 *)rec.colptrT, (i64 *)rec.rowvalT, (double *)rec.nzvalT, &actual);
                r.nnz[0] = actual;
            }
        }
        ctx->tm_hist.push_back(r);
    }
    if (ctx->tm_sticky) ctx->err = ctx->tm_sticky_msg;
    ctx->tm_rec.clear();
    ctx->tm_first = ctx->tm_next;
    return ret;
}

// The foreign path of otmb_tm_args.given: T = ((Tadv + TκH) + TκVML) + TκVdeep (:147) from four materialised operands -- the ones the fill
// pass has just written into the caller's arrays and the GIVEN ones where they lie -- by SparseArrays' `+` on the device (otmb_spadd.hip:
// per column a sorted merge, a missing operand is +0.0, exact-zero results are dropped), left to right, through two temporaries.
static int32_t foreign_sum(otmb_ctx *ctx, TmPlan &pl, const TmParams &p) {
    const otmb_tm_args &a = pl.args;
    const i64 n = a.n_wet;
    if (pl.wet_base != 0 || pl.nnz_base[0] != 0) return otmb_fail(ctx, OTMB_ERR_GIVEN_FOREIGN, "depth slab");
    otmb_csc op[5];
    for (int m = 1; m < 5; ++m) {
        if ((pl.given >> m) & 1u) op[m] = a.given[m];
        else { op[m].colptr = p.colptr[m]; op[m].rowval = p.rowval[m]; op[m].nzval = p.nzval[m]; op[m].nnz = pl.built_nnz[m]; }
    }
    int32_t rc;
    otmb_csc acc = op[1];
    for (int step = 2; step < 5; ++step) {
        int64_t k = 0;
        if ((rc = otmb_spadd_plan_dev(ctx, n, acc.colptr, acc.rowval, acc.nzval, op[step].colptr, op[step].rowval, op[step].nzval, &k))) return rc;
        i64 *Cp, *Ci;
        double *Cx;
        if (step == 4) {  // the last add lands in the caller's T arrays (planned at the sum of the operands' counts: k cannot exceed it)
            if (k > pl.nnz[0]) return otmb_fail(ctx, OTMB_ERR_CAPACITY, "T");
            Cp = p.colptr[0]; Ci = p.rowval[0]; Cx = p.nzval[0];
        } else {
            DevBuf *t = &ctx->given_tmp[(step - 2) * 3];
            if ((rc = otmb_reserve(ctx, t[0], (size_t)(n + 1) * 8)) || (rc = otmb_reserve(ctx, t[1], (size_t)(k > 0 ? k : 1) * 8)) ||
                (rc = otmb_reserve(ctx, t[2], (size_t)(k > 0 ? k : 1) * 8)))
                return rc;
            Cp = (i64 *)t[0].p; Ci = (i64 *)t[1].p; Cx = (double *)t[2].p;
        }
        if ((rc = otmb_spadd_fill_dev(ctx, n, acc.colptr, acc.rowval, acc.nzval, op[step].colptr, op[step].rowval, op[step].nzval, Cp, Ci, Cx))) return rc;
        acc.colptr = Cp; acc.rowval = Ci; acc.nzval = Cx; acc.nnz = k;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    pl.nnz[0] = acc.nnz;
    return OTMB_OK;
}

void otmb_tm_plan_free(otmb_ctx *ctx) {
    delete ctx->plan;
    ctx->plan = nullptr;
}

void otmb_tm_plan_invalidate(otmb_ctx *ctx) {
    // only the two-phase plan points into the host entry points' staging slots; pending asynchronous steps keep their verdicts
    if (ctx->plan) ctx->plan->valid = false;
}

bool otmb_tm_plan_foreign(otmb_ctx *ctx) { return ctx->plan && ctx->plan->foreign && ctx->plan->want_t; }
// matrices (bit m) the pending plan does not hand out: neither counted nor written -- T alone, given operators
unsigned otmb_tm_plan_skip(otmb_ctx *ctx) {
    if (!ctx->plan) return 0u;
    const TmPlan &pl = *ctx->plan;
    return (pl.foreign && pl.want_t) ? (pl.skip & ~1u) : pl.skip;  // (a foreign build's T is written by the sparse adds)
}

int32_t otmb_tm_plan_query(otmb_ctx *ctx, int64_t *nnz, int64_t *N) {
    if (!ctx->plan || !ctx->plan->valid) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    for (int m = 0; m < 5; ++m) nnz[m] = ctx->plan->nnz[m];
    *N = ctx->plan->args.n_wet;
    return OTMB_OK;
}

extern "C" {

int32_t otmb_transportmatrix_plan_dev(otmb_ctx *ctx, const otmb_tm_args *a, int64_t nnz[5]) {
    if (!ctx || !a || !nnz) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (ctx->plan) ctx->plan->valid = false;
    int32_t rc;
    if ((rc = validate_args(ctx, a))) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 ntiles = (a->n_wet + TM_THREADS - 1) / TM_THREADS;
    if ((rc = otmb_reserve(ctx, ctx->tm_sums, (size_t)(ntiles + 1) * TM_NF * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->tm_offs, (size_t)(ntiles + 1) * TM_NF * sizeof(i64) + otmb_scan_scratch(ntiles, TM_NF)))) return rc;
    if (!ctx->plan) ctx->plan = new TmPlan();
    TmPlan &pl = *ctx->plan;
    pl.args = *a;
    pl.ntiles = ntiles;
    if ((rc = classify_given(ctx, *a, pl))) return rc;  // (may run the comparing pass: before anything of this plan is on the stream)
    TmParams p;
    fill_params(p, *a, ctx, &pl);
    int *dflags = (int *)ctx->flags.p;
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS);
    HIP_TRY(ctx, hipMemsetAsync(dflags, 0, OTMB_TM_STATE_BYTES, ctx->stream));  // flag words and totals: one block
    pl.rho_in_fill = false;
    int fbuf = -1;
    if (ntiles > 0 && (fbuf = ffc_match(ctx, *a, pl)) >= 0) {
        // the counts came with the fluxes (otmb_facefluxes_counts_dev): no counting pass
        pl.rho_in_fill = true;
        ffc_consume(ctx, fbuf, p, (i64 *)ctx->tm_offs.p, dtot, (i64 *)ctx->tm_offs.p + (ntiles + 1) * TM_NF, ntiles, true);
    } else if (ntiles > 0) {
        if ((rc = ensure_push_mask(ctx, *a, p))) return rc;
        if (p.count_order == 2 && (rc = build_tile_order(ctx, *a, ntiles, p))) return rc;
        {
            KernelTimer kt(ctx, K_TM_COUNT);
            hipLaunchKernelGGL(tm_count_kernel<TM_COUNT_TPB>, dim3((unsigned)((ntiles + TM_COUNT_TPB - 1) / TM_COUNT_TPB)),
                               dim3(TM_THREADS), 0, ctx->stream, p, (i64)ntiles);
        }
        {
            KernelTimer kt(ctx, K_TILESCAN);
            otmb_launch_tilescan(ctx->stream, p.tilesums, (i64 *)ctx->tm_offs.p, dtot, ntiles, TM_NF,
                                 (i64 *)ctx->tm_offs.p + (ntiles + 1) * TM_NF);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_TM_STATE_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (fbuf >= 0 && ctx->h_flags[FLAG_FLUX_INTO_LAND] && a->rho && a->n_wet > 0) {
        // the reference tests ρ (:233) before its loop can run into land: rank the two errors as it does (nothing has read ρ yet)
        hipLaunchKernelGGL(rho_nan_kernel, dim3((unsigned)((a->n_wet + 255) / 256)), dim3(256), 0, ctx->stream, a->rho, (const i64 *)a->lwet,
                           (i64)a->n_wet, dflags);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_TM_STATE_BYTES, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if ((rc = check_flags(ctx, nullptr, a->ignore_ops | (int)pl.given))) return rc;
    for (int m = 0; m < 5; ++m) nnz[m] = pl.nnz[m] = pl.built_nnz[m] = ctx->h_tot[m];  // (0 for what is not materialised)
    if (pl.foreign && pl.want_t) {
        // T = ((Tadv + TκH) + TκVML) + TκVdeep by the device sparse add (:147): its pattern is the union of the four operands', at most the
        // sum of their counts -- what the caller's T arrays must hold until otmb_transportmatrix_nnz gives the final count
        i64 bound = 0;
        for (int m = 1; m < 5; ++m) bound += ((pl.given >> m) & 1u) ? a->given[m].nnz : pl.built_nnz[m];
        nnz[0] = pl.nnz[0] = bound;
    }
    pl.valid = true;
    return OTMB_OK;
}

int32_t otmb_transportmatrix_set_slab(otmb_ctx *ctx, int64_t wet_base) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    if (!ctx->plan) ctx->plan = new TmPlan();
    ctx->plan->valid = false;
    ctx->plan->wet_base = wet_base;
    for (int m = 0; m < 5; ++m) ctx->plan->nnz_base[m] = 0;
    return OTMB_OK;
}

int32_t otmb_transportmatrix_set_nnz_base(otmb_ctx *ctx, const int64_t nnz_base[5]) {
    if (!ctx || !nnz_base) return OTMB_ERR_INVALID_ARG;
    if (!ctx->plan) ctx->plan = new TmPlan();
    for (int m = 0; m < 5; ++m) ctx->plan->nnz_base[m] = nnz_base[m];
    return OTMB_OK;
}

int32_t otmb_transportmatrix_fill_dev(otmb_ctx *ctx, int64_t *const colptr[5], int64_t *const rowval[5],
                                      double *const nzval[5]) {
    if (!ctx || !colptr || !rowval || !nzval) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (!ctx->plan || !ctx->plan->valid) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    TmPlan &pl = *ctx->plan;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    TmParams p;
    fill_params(p, pl.args, ctx, &pl);
    p.rho_in_fill = pl.rho_in_fill ? 1 : 0;
    for (int m = 0; m < 5; ++m) {
        const bool wanted = !((pl.skip >> m) & 1u) || (m == 0 && pl.foreign && pl.want_t);  // (T of a foreign build: written by the sparse adds below)
        if (wanted && (!colptr[m] || (pl.nnz[m] > 0 && (!rowval[m] || !nzval[m])))) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
        p.colptr[m] = wanted ? (i64 *)colptr[m] : nullptr; p.rowval[m] = wanted ? (i64 *)rowval[m] : nullptr;
        p.nzval[m] = wanted ? nzval[m] : nullptr;
    }
    int32_t rc;
    int *dflags = (int *)ctx->flags.p;
    if (pl.ntiles > 0) {
        if ((rc = build_tile_order(ctx, pl.args, pl.ntiles, p))) return rc;
        KernelTimer kt(ctx, K_TM_FILL);
        launch_fill(ctx, p, 0);
    }
    if (pl.ntiles == 0) {  // (otherwise the fill kernel's last tile writes the closing colptr entries)
        KernelTimer kt(ctx, K_TM_FINISH);
        hipLaunchKernelGGL(tm_finish_colptr, dim3(1), dim3(64), 0, ctx->stream, p.colptr[0], p.colptr[1], p.colptr[2],
                           p.colptr[3], p.colptr[4], (i64)pl.args.n_wet, p.nnz_base[0] + pl.nnz[0], p.nnz_base[1] + pl.nnz[1],
                           p.nnz_base[2] + pl.nnz[2], p.nnz_base[3] + pl.nnz[3], p.nnz_base[4] + pl.nnz[4]);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_NFLAGS_TM * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    // the values exist only now: raise the reference's errors, and repair T if entries cancelled
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // a plan is consumed by its fill: T's final count may be smaller than the reserved (union) one, so a second fill into
    // buffers sized from otmb_transportmatrix_nnz would overflow them -- plan again instead
    pl.valid = false;
    if ((rc = check_flags(ctx, nullptr, pl.args.ignore_ops | (int)pl.given))) return rc;
    if (pl.foreign && pl.want_t) return foreign_sum(ctx, pl, p);
    if (pl.skip & 1u) return OTMB_OK;  // (no T: nothing to compact)
    if (ctx->h_flags[FLAG_T_CANCEL]) {
        i64 actual = pl.nnz[0];
        if ((rc = t_fixup(ctx, pl.args.n_wet, pl.nnz_base[0], pl.nnz[0], p.colptr[0], p.rowval[0], p.nzval[0], &actual))) return rc;
        pl.nnz[0] = actual;
    }
    return OTMB_OK;
}

#ifdef OTMB_DBG_STAMPS
// diagnostic build only: copy the stamp buffer of the last asynchronous fill pass to the host (tools/stamps.py)
int32_t otmb_debug_stamps(otmb_ctx *ctx, uint64_t *host, int64_t n_words) {
    if (!ctx || !host || !ctx->stamps.p || (size_t)n_words * 8 > ctx->stamps.cap) return OTMB_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(host, ctx->stamps.p, (size_t)n_words * 8, hipMemcpyDeviceToHost));
    return OTMB_OK;
}
#endif

int32_t otmb_transportmatrix_failed_step(otmb_ctx *ctx, int64_t *step) {
    if (!ctx || !step) return OTMB_ERR_INVALID_ARG;
    *step = ctx->tm_failed_step;
    return OTMB_OK;
}

int32_t otmb_transportmatrix_nnz(otmb_ctx *ctx, int64_t nnz[5]) {
    if (!ctx || !nnz) return OTMB_ERR_INVALID_ARG;
    if (!ctx->plan) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    for (int m = 0; m < 5; ++m) nnz[m] = ctx->plan->nnz[m];
    return OTMB_OK;
}

// One pass, asynchronous: the caller provides output buffers of known capacity (nnz per column is at
// most 7, 7, 5, 3, 3 for T, Tadv, TκH, TκVML, TκVdeep).  Errors and the nnz are collected afterwards by
// otmb_transportmatrix_result (which synchronises).
struct TmFused { const void *umo = nullptr, *vmo = nullptr; double fill = 0.0; int kind = 0; };  // kind: 0 none, 1 Float64, 2 Float32
static int32_t transportmatrix_dev_impl(otmb_ctx *ctx, const otmb_tm_args *a, int64_t *const colptr[5], int64_t *const rowval[5],
                                        double *const nzval[5], const int64_t capacity[5], const TmFused &fu);

int32_t otmb_transportmatrix_dev(otmb_ctx *ctx, const otmb_tm_args *a, int64_t *const colptr[5],
                                 int64_t *const rowval[5], double *const nzval[5], const int64_t capacity[5]) {
    return transportmatrix_dev_impl(ctx, a, colptr, rowval, nzval, capacity, TmFused());
}

// The fused device-resident step (include/otmb.h): facefluxes that stores ϕtop only (+ the tile counts), then scan + fill with the other
// five fluxes re-derived from umo / vmo where they are used.  Same five matrices bit for bit as otmb_facefluxes_counts_dev +
// otmb_transportmatrix_dev; 64 bytes per cell less HBM traffic.
int32_t otmb_step_dev(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32, double fill, const uint8_t *wetflags,
                      const void *count_tables, double *phi_top, const otmb_tm_args *a, int64_t *const colptr[5], int64_t *const rowval[5],
                      double *const nzval[5], const int64_t capacity[5]) {
    if (!ctx || !umo || !vmo || !wetflags || !count_tables || !phi_top || !a) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (a->nx < 3 || a->nz > 128 || ctx->count_in_ff == 0 || a->n_wet <= 0)
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "otmb_step_dev needs nx >= 3, nz <= 128, wet cells and the counts in facefluxes (OTMB_COUNT_IN_FF)");
    otmb_tm_args b = *a;
    for (int f = 0; f < 6; ++f) b.phi[f] = nullptr;
    b.phi[OTMB_TOP] = phi_top;
    b.push_mask = (const uint16_t *)phi_top;  // the token the counts are keyed to (never read as a mask)
    int32_t rc;
    if ((rc = validate_args(ctx, &b, true))) return rc;
    otmb_ff_counts cnt;
    cnt.tables = count_tables; cnt.lwet3d = b.lwet3d; cnt.mlotst = b.mlotst; cnt.zt = b.zt; cnt.n_wet = b.n_wet;
    cnt.upwind = b.upwind; cnt.only_t = b.only_t;
    double *phi[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    phi[OTMB_TOP] = phi_top;
    if ((rc = otmb_facefluxes_top_counts(ctx, umo, vmo, src_is_f32, wetflags, fill, b.nx, b.ny, b.nz, b.topology, phi, (uint16_t *)phi_top, &cnt))) return rc;
    TmFused fu;
    fu.umo = umo; fu.vmo = vmo; fu.fill = fill; fu.kind = src_is_f32 ? 2 : 1;
    return transportmatrix_dev_impl(ctx, &b, colptr, rowval, nzval, capacity, fu);
}

static int32_t transportmatrix_dev_impl(otmb_ctx *ctx, const otmb_tm_args *a, int64_t *const colptr[5], int64_t *const rowval[5],
                                        double *const nzval[5], const int64_t capacity[5], const TmFused &fu) {
    if (!ctx || !a || !colptr || !rowval || !nzval || !capacity) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    int32_t rc;
    if ((rc = validate_args(ctx, a, fu.kind != 0))) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 ntiles = (a->n_wet + TM_THREADS - 1) / TM_THREADS;
    // COUNT (or the counts that came with the fluxes) -> tile scan -> FILL enqueued back to back with no host round trip (the totals
    // stay on the device).
    if ((rc = otmb_reserve(ctx, ctx->tm_sums, (size_t)(ntiles + 1) * TM_NF * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->tm_offs, (size_t)(ntiles + 1) * TM_NF * sizeof(i64) + otmb_scan_scratch(ntiles, TM_NF)))) return rc;
    if (!ctx->plan) ctx->plan = new TmPlan();
    TmPlan &pl = *ctx->plan;
    pl.valid = false;
    pl.args = *a;
    pl.ntiles = ntiles;
    // operators the caller passes (otmb_tm_args.given): a derived one is re-derived in registers (the FIRST call for a grid and κ runs the
    // comparing pass and waits for its verdict: one stream synchronisation, like the tile order); a foreign one needs the two-phase protocol
    if ((rc = classify_given(ctx, *a, pl))) return rc;
    if (pl.foreign && pl.want_t) return otmb_fail(ctx, OTMB_ERR_GIVEN_FOREIGN);
    TmParams p;
    fill_params(p, *a, ctx, &pl);
    p.umo = fu.umo; p.vmo = fu.vmo; p.fillv = fu.fill; p.fused = fu.kind;
    for (int m = 0; m < 5; ++m) {
        const bool wanted = !((pl.skip >> m) & 1u);
        if (wanted && (!colptr[m] || !rowval[m] || !nzval[m])) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
        if (wanted && capacity[m] <= 0) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "capacity");
        p.colptr[m] = wanted ? (i64 *)colptr[m] : nullptr; p.rowval[m] = wanted ? (i64 *)rowval[m] : nullptr;
        p.nzval[m] = wanted ? nzval[m] : nullptr;
        p.cap[m] = wanted ? capacity[m] : 0;
    }
    if (ctx->tm_hist_final) { ctx->tm_hist.clear(); ctx->tm_hist_final = false; }  // a new pipeline starts
    // this step's own state block (flag words + totals): a ring slot, so that the verdict on every step of a pipeline
    // of asynchronous calls is still there when otmb_transportmatrix_result finally looks.  A full ring is folded
    // into the sticky (status, step) pair first -- one host synchronisation per OTMB_RING steps.
    // (one slot short of the ring: every fill zeroes the slot of the step after it, which must not be a pending one)
    if (ctx->tm_next - ctx->tm_first >= OTMB_RING - 1) {
        int32_t frc;
        if ((frc = fetch_ring(ctx))) return frc;
        if ((frc = fold_pending(ctx))) return frc;
    }
    int *dflags = otmb_ring_tm((int *)ctx->ring.p, ctx->tm_next);
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS);
    p.flags = dflags;
    p.totals = dtot;
    // The state blocks stay on the device until somebody asks (otmb_transportmatrix_result, or a full ring): no copy per
    // step, and no memset either when the previous step's fill has already zeroed this block (two 4-5 us blit kernels
    // per step on the stream otherwise, 1.6 % of a 1-degree step).
    const int slot = (int)(ctx->tm_next % OTMB_RING), slot_after = (int)((ctx->tm_next + 1) % OTMB_RING);
    if ((ctx->ring_clean >> slot) & 1ull) {
        ctx->ring_clean &= ~(1ull << slot);
    } else {
        HIP_TRY(ctx, hipMemsetAsync(dflags, 0, OTMB_TM_STATE_BYTES, ctx->stream));  // flag words and totals: one block
    }
    ctx->ring_clean &= ~(1ull << slot_after);
    p.next_state = (ntiles > 0) ? otmb_ring_tm((int *)ctx->ring.p, ctx->tm_next + 1) : nullptr;
    if (ntiles == 0) {
        KernelTimer kt(ctx, K_TM_FINISH);
        hipLaunchKernelGGL(tm_finish_colptr, dim3(1), dim3(64), 0, ctx->stream, p.colptr[0], p.colptr[1], p.colptr[2],
                           p.colptr[3], p.colptr[4], (i64)0, p.nnz_base[0], p.nnz_base[1], p.nnz_base[2], p.nnz_base[3],
                           p.nnz_base[4]);
    } else {
        p.rho_in_fill = 1;  // count and fill both run before the flags are read: check ρ where it is loaded anyway
        i64 *gsum = (i64 *)ctx->tm_offs.p + (ntiles + 1) * TM_NF;
        const bool infill = ntiles <= TM_INFILL_GROUPS * OTMB_SCAN_GROUP;  // first scan level only; the fill pass adds the group bases
        const int fbuf = ffc_match(ctx, *a, pl);
        if (fbuf >= 0) {
            // the counts came with the fluxes (otmb_facefluxes_counts_dev): no counting pass, the scan unpacks them
            ffc_consume(ctx, fbuf, p, (i64 *)ctx->tm_offs.p, dtot, gsum, ntiles, !infill);
            if (infill) p.gsum = gsum;
        } else if (fu.kind != 0) {
            return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "otmb_step_dev: the counts of its own facefluxes are not there (depth slab?)");
        } else {
            if ((rc = ensure_push_mask(ctx, *a, p))) return rc;
            if (p.count_order == 2 && (rc = build_tile_order(ctx, *a, ntiles, p))) return rc;
            {
                KernelTimer kt(ctx, K_TM_COUNT);
                hipLaunchKernelGGL(tm_count_kernel<TM_COUNT_TPB>, dim3((unsigned)((ntiles + TM_COUNT_TPB - 1) / TM_COUNT_TPB)),
                                   dim3(TM_THREADS), 0, ctx->stream, p, (i64)ntiles);
            }
            KernelTimer kt(ctx, K_TILESCAN);
            if (infill) {
                otmb_launch_tilescan_groups(ctx->stream, p.tilesums, (i64 *)ctx->tm_offs.p, gsum, ntiles, TM_NF);
                p.gsum = gsum;
            } else {
                otmb_launch_tilescan(ctx->stream, p.tilesums, (i64 *)ctx->tm_offs.p, dtot, ntiles, TM_NF, gsum);
            }
        }
        {
#ifdef OTMB_DBG_STAMPS
            if ((rc = otmb_reserve(ctx, ctx->stamps, (size_t)ntiles * (TM_THREADS / 64) * OTMB_NSTAMP * sizeof(u64)))) return rc;
            p.status = (u64 *)ctx->stamps.p;
#endif
            if ((rc = build_tile_order(ctx, *a, ntiles, p))) return rc;
            KernelTimer kt(ctx, K_TM_FILL);
            launch_fill(ctx, p, fu.kind);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    if (p.next_state) ctx->ring_clean |= 1ull << slot_after;
    ctx->tm_rec.push_back({p.colptr[0], p.rowval[0], p.nzval[0], (i64)a->n_wet, p.nnz_base[0], (int)a->ignore_ops | (int)pl.given});
    ctx->tm_next += 1;
    pl.onepass_pending = true;
    return OTMB_OK;
}

int32_t otmb_transportmatrix_result(otmb_ctx *ctx, int64_t nnz[5]) {
    if (!ctx || !nnz) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (!ctx->plan || !ctx->plan->onepass_pending) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    int32_t frc;
    if ((frc = fetch_ring(ctx))) return frc;
    ctx->plan->onepass_pending = false;
    // every step enqueued since the previous result: the FIRST one that failed is reported (the reference would have
    // thrown there, src/matrixbuilding.jl:39,61,90,114,233), with its position in the error text; every step that did
    // not fail has its own nnz (and its T compacted if entries cancelled): otmb_transportmatrix_result_step
    const i64 n_steps = ctx->tm_next;
    frc = fold_pending(ctx);
    const int32_t st = ctx->tm_sticky;
    ctx->tm_failed_step = ctx->tm_sticky_step;
    ctx->tm_sticky = 0; ctx->tm_sticky_step = -1;
    ctx->tm_first = ctx->tm_next = 0;
    ctx->tm_hist_final = true;
    if (st) {
        if (n_steps > 1) {
            char where[96];
            snprintf(where, sizeof where, " (asynchronous step %lld of %lld)", (long long)ctx->tm_failed_step + 1, (long long)n_steps);
            ctx->err += where;
        }
        return st;
    }
    if (frc) return frc;
    if (ctx->tm_hist.empty()) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    for (int m = 0; m < 5; ++m) nnz[m] = ctx->plan->nnz[m] = ctx->tm_hist.back().nnz[m];
    return OTMB_OK;
}

int32_t otmb_transportmatrix_result_step(otmb_ctx *ctx, int64_t step, int64_t nnz[5]) {
    if (!ctx || !nnz) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (!ctx->tm_hist_final || step < 0 || (size_t)step >= ctx->tm_hist.size()) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "no such asynchronous step");
    for (int m = 0; m < 5; ++m) nnz[m] = ctx->tm_hist[(size_t)step].nnz[m];
    return ctx->tm_hist[(size_t)step].status;
}

}  // extern "C"
