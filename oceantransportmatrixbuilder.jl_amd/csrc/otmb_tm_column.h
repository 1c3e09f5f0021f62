// otmb_tm_column.h -- one CSC column of (T, Tadv, TκH, TκVML, TκVdeep) from the local stencil.
//
// Gather formulation of src/matrixbuilding.jl:221-299 (advection), :337-418 (horizontal diffusion),
// :438-479 (vertical diffusion), sparse()'s duplicate summation (:41,63,92,116) and the sparse adds
// (:147).  The reference scatters: wet cell 𝑖 pushes triplets into its own and its neighbours' columns,
// then sparse() sorts and sums.  Every triplet of column c comes from c itself or from one of the <= 7
// cells whose neighbour (in some direction) is c, so the thread that owns c rebuilds the column directly:
//   * which triplets land in column c, and the order the reference emits them in (ascending emitting wet
//     index, then W,E,S,N,B,T, then first/second push), is a pure function of the local stencil;
//     duplicates are summed left-to-right in that order, first touch copies (explicit zeros, -0.0 kept);
//   * T[r,c] = ((Tadv + TκH) + TκVML) + TκVdeep with absent operands +0.0, stored iff != 0;
//   * rows ascend in wet index == linear index (makeindices is monotone; verified by the kernel).
// Float64 throughout, compiled with -ffp-contract=off: every value is the reference's own expression.
#pragma once
#include "otmb_common.h"
#include "otmb_topology.h"

struct TmParams {
    const double *phi[6];
    // The fused step (otmb_step_dev): only ϕtop exists in memory (phi[OTMB_TOP]); the other five fluxes are what facefluxes WOULD have stored,
    // re-derived where they are used: ϕeast / ϕwest / ϕnorth / ϕsouth are masked copies of umo / vmo (nofluxboundaries! + replace + shift,
    // src/velocities.jl:161-175,203-224) and ϕbottom is ϕtop of the level below (:238-240).  fused: 0 = six arrays, 1 = umo / vmo are Float64, 2 = Float32.
    const void *umo, *vmo;
    double fillv;
    int fused;
    const double *v, *thk, *rho;
    double rho_s;
    const i64 *lw;    // Lwet3D (global wet ranks in a slab run), 0 = missing
    const i64 *lwet;  // Lwet: 1-based linear indices (in this grid) of the wet cells this launch owns
    const uint16_t *mask;  // push mask of every cell of the grid (counting pass; otmb_push_bits)
    const double *edge[4], *dist[4];
    const double *area, *zt, *ml;
    double kH, kML, kDeep;
    int nx, ny, nz, topo, upwind;
    // (round 6) a given TκH in which the comparing pass found the DERIVED ROWS: its values, where they lie, instead of re-deriving them -- the given
    // matrix holds a column's <= 5 values contiguously (one coalesced run per wave) where re-deriving them takes 5 thkcello and 18 metric loads per column,
    // a third of the pass's L1 requests; and when its values are NOT the derived ones (another κH: hmust) they are what T must carry.
    // hcp: the matrix's colptr for this launch's columns (a slab's slice: entries count from hcp[0]), hx: its nzval, hnnz: entries.
    const i64 *hcp;
    const double *hx;
    i64 hnnz;
    // ... and a given TκVdeep whose ROWS are the derived ones but whose values are not (built with another κVdeep): its values are read (the DREAD
    // instantiations of the fill kernel) and enter T in place of the derived ones.  (A TκH of that kind is read by the HREAD kernels, seam row included.)
    const i64 *dcp;
    const double *dx;
    i64 dnnz;
    int hmust;         // (host side) the given TκH's values are NOT the derived ones: reading them is not a choice (launch_fill)
    unsigned skip;     // bit m: matrix m is evaluated (T is the sum of all four) but neither counted nor written -- otmb_tm_args.only_t (bits 1-4),
                       // a given operator with the derived rows (otmb_tm_args.given: values re-derived or read), T itself when a foreign given operator makes it a sparse add
    u64 keep;          // the packed count word's fields of the matrices that ARE counted (T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10)
    int rho_in_fill;   // the ρ-NaN check (:233) is done by the fill pass (which loads ρ anyway) instead of the counting pass:
                       // set when both passes run before the flags are read (otmb_transportmatrix_dev)
    i64 P, G;
    i64 n_own;         // number of entries of lwet = columns produced
    i64 wet_base;      // wet rank of column 0 is wet_base + 1 (depth slabs; 0 otherwise)
    i64 nnz_base[5];   // entries owned by lower-ranked slabs
    i64 cap[5];        // capacity of rowval/nzval (one-pass mode)
    // outputs
    i64 *colptr[5], *rowval[5];
    double *nzval[5];
    i64 *totals;           // [5] nnz of this launch (one-pass mode)
    // scan state
    uint32_t *tilesums;    // [ntiles][5]  (COUNT writes)
    const i64 *tileoffs;   // [ntiles][5]  (FILL reads)
    const i64 *gsum;       // [groups][5] totals of the scan groups, when tileoffs are group-relative (else NULL)
    u64 *status;           // diagnostic builds only (OTMB_DBG_STAMPS): the stamp buffer
    int *flags;
    int *next_state;       // the NEXT asynchronous step's state block, zeroed by this fill (or NULL): no memset between steps
    const unsigned *order; // fill pass: tile taken by the q-th workgroup slot (march order), or NULL = wet-rank order
    int count_order;       // counting pass: 0 blockIdx order, 1 XCD-contiguous eighths, 2 XCD-contiguous eighths of `order`
    unsigned nt_order;     // number of tiles (the fill pass's grid may be a few workgroups larger: xcd_position)
    unsigned nheavy;       // `order` starts with this many HEAVY tiles (tripolar seam row: generic column builder), dealt over the XCDs
};

// Which position of the tile sequence does workgroup b take?  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2):
// XCD x = b % 8 takes (1) every eighth of the sequence's first `nh` positions -- the HEAVY tiles, whose waves live about twice as
// long (tripolar seam row: generic column builder); left in one XCD's share they made that XCD finish 10-20 us after the other
// seven (profiles/r04: dispatch timeline), 5 % of a 1 degree fill pass with seven eighths of the chip idle -- and then (2) the x-th
// contiguous eighth of the remaining positions, so that a tile's south / north rows and the levels above / below, which the same
// XCD touched a little earlier, are L2 hits instead of fabric re-reads.  Returns false for the (at most 15) workgroups of the
// rounded-up grid that have nothing to do.  Speed only: every position is taken exactly once whatever nh is.
__device__ __forceinline__ bool xcd_position(unsigned b, unsigned nt, unsigned nh, unsigned &pos) {
    const unsigned x = b % 8u, y = b / 8u;
    const unsigned hx = (nh + 7u - x) / 8u;  // heavy positions x, x + 8, ... below nh
    const unsigned R = nt - nh, q = R / 8u, r = R % 8u;
    const unsigned rx = q + (x < r ? 1u : 0u), rstart = (x < r) ? x * (q + 1u) : r * (q + 1u) + (x - r) * q;
    if (y >= hx + rx) return false;
    pos = (y < hx) ? x + 8u * y : nh + rstart + (y - hx);
    return true;
}
// grid size of a launch that maps its workgroups with xcd_position
static inline unsigned xcd_grid(unsigned nt, unsigned nh) {
    unsigned m = 0;
    for (unsigned x = 0; x < 8; ++x) {
        const unsigned hx = (nh + 7u - x) / 8u, R = nt - nh, c = hx + R / 8u + (x < R % 8u ? 1u : 0u);
        m = c > m ? c : m;
    }
    return 8u * m;
}

// Diagnostic build only (-DOTMB_DBG_STAMPS, tools/stamps.py): s_memtime stamps of the phases of a wave of the fill
// pass, kept in SGPR pairs and written by lane 0 at the end to a buffer nothing else reads (p.status).  STAMP(n, WAITVM)
// with WAITVM drains the wave's vector-memory operations first, so the stamp says when the loads were back.
#ifdef OTMB_DBG_STAMPS
#define OTMB_NSTAMP 10  // 0-6 s_memtime stamps (shader clock of the wave's XCD), 7 HW_ID | XCC_ID << 32, 8 / 9 s_memrealtime (100 MHz, chip-wide) at entry / end
struct Stamps { unsigned long long t[OTMB_NSTAMP]; };
#define STAMP(st, n, WAITVM)                                                                              \
    do {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (WAITVM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"((st).t[n])::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)
#else
struct Stamps {};
#define STAMP(st, n, WAITVM) do { } while (0)
#endif

// slots of a column: the cells that can hold a row of column c
enum { S_A = 0, S_S = 1, S_SELF = 2, S_EC = 3, S_WC = 4, S_FQ = 5, S_N = 6, S_B = 7, NSLOT = 8 };

struct Column {
    i64 idx[NSLOT];      // wet rank of the slot's cell (row index), 0 = no such wet cell
    double adv[NSLOT];   // Tadv values
    double hh[NSLOT];    // TκH values   (slots SELF, EC, WC, FQ, S, N)
    double ml[NSLOT];    // TκVML values (slots SELF, A, B)
    double dp[NSLOT];    // TκVdeep values
    double tv[NSLOT];    // T values (filled by the kernel once the four operators are known)
    unsigned padv, phh, pml, pdp;  // presence masks (bit = slot)
    unsigned bef[NSLOT];  // bef[X]: slots ordered before X in the column
};

__device__ __forceinline__ void acc(double &val, unsigned &pres, int slot, double x) {
    // sparse(): first touch copies, later ones combine acc = acc + x in emission order
    val = ((pres >> slot) & 1u) ? val + x : x;
    pres |= 1u << slot;
}
// accumulate into one of the four row-mate slots chosen at run time.  Written with value selects
// only: an if-chain over val[slot] is turned by the optimiser into a run-time indexed access,
// which drags the whole Column into scratch memory.
__device__ __forceinline__ void acc_rowmate(double (&val)[NSLOT], unsigned &pres, int slot, double x) {
    const double vS = val[S_SELF], vE = val[S_EC], vW = val[S_WC], vF = val[S_FQ];  // unconditional loads
    const bool tS = slot == S_SELF, tE = slot == S_EC, tW = slot == S_WC, tF = slot == S_FQ;
    const double cur = tS ? vS : (tE ? vE : (tW ? vW : vF));
    const double nv = ((pres >> slot) & 1u) ? cur + x : x;
    val[S_SELF] = tS ? nv : vS;
    val[S_EC] = tE ? nv : vE;
    val[S_WC] = tW ? nv : vW;
    val[S_FQ] = tF ? nv : vF;
    pres |= 1u << slot;
}
__device__ __forceinline__ double sel_pos(double x, int upwind) {  // max(ϕ,0) or ϕ/2  (:244,262,280)
    return upwind ? ((x > 0.0) ? x : 0.0) : x / 2;
}
__device__ __forceinline__ double sel_neg(double x, int upwind) {  // min(ϕ,0) or ϕ/2  (:253,271,289)
    return upwind ? ((x < 0.0) ? x : 0.0) : x / 2;
}
__device__ __forceinline__ bool nonzero(double f) { return (f > 0.0) || (f < 0.0); }
__device__ __forceinline__ double jl_min(double a, double b) {
    return (isnan(a) || isnan(b)) ? __builtin_nan("") : ((a < b) ? a : b);
}
__device__ __forceinline__ void raise_flag(int *flags, int f) {
    if (flags[f] == 0) atomicExch(&flags[f], 1);
}

// replace(x, NaN => 0.0, FillValue => 0.0) with isequal semantics (src/velocities.jl:203,215): facefluxes' ff_replace
__device__ __forceinline__ double tm_replace(double x, double fill) {
    return (isnan(x) || __double_as_longlong(x) == __double_as_longlong(fill)) ? 0.0 : x;
}
// ---- THE value expressions of the three generators: one copy, used by the generic column builder (build_column) and by the
// regular-cell arithmetic (column_compute) of the gather kernel (fast_column). -------------
#define FDIV(a, b) ((a) / (b))
// pushTadvectionvalues! (src/matrixbuilding.jl:193-204): ρ̄ = (ρx + ρc) / 2; row x: -ϕ / (ρ̄ vx); diagonal: ϕ / (ρ̄ vc)
__device__ __forceinline__ void adv_pair(double phi, double rx, double rc, double vx, double vc, double &off, double &dg) {
    const double rb = (rx + rc) / 2;
    off = FDIV(-phi, (rb * vx));
    dg = FDIV(phi, (rb * vc));
}
// pushTmixingvalues! for a horizontal neighbour (:348-415, :426-435): a = min(thk_c e_c, thk_x e_x) shared by the cell's own push
// (+Tval on the diagonal) and the neighbour's push towards the cell (-Tval' on row x)
__device__ __forceinline__ void h_pair(double kH, double tc, double e_c, double tx, double e_x, double d_c, double vc, double d_x, double vx,
                                       double &own, double &in) {
    const double a = jl_min(tc * e_c, tx * e_x);
    own = FDIV((kH * a), (d_c * vc));
    in = FDIV((kH * a), (d_x * vx));
}
// the same for a vertical neighbour (:450-477): a = area2D, d = |zt[k] - zt[k']|
__device__ __forceinline__ void v_pair(double kar, double d, double vc, double vx, double &own, double &in) {
    own = FDIV(kar, (d * vc));
    in = FDIV(kar, (d * vx));
}

// Build the column of wet cell `cell` (c = own wet rank > 0): generic path, any topology corner case.
__device__ __forceinline__ void build_column(const TmParams &p, const Cell &cell, i64 c, Column &col) {
    const int nx = p.nx, ny = p.ny, nz = p.nz, up = p.upwind;
    const i64 L = cell.L;
    const int i = cell.i, j = cell.j, k = cell.k;
    const int ie = (i + 1 < nx) ? i + 1 : 0, iw = (i > 0) ? i - 1 : nx - 1;
    const i64 LEc = cell.row0 + ie, LWc = cell.row0 + iw;
    const i64 LS = nb_jm1(cell, nx), LNq = nb_jp1(cell, nx, ny, p.topo);
    const i64 LA = nb_km1(cell, p.P), LB = nb_kp1(cell, nz, p.P);
    const bool fold = (j == ny - 1) && (LNq >= 0);  // north neighbour through the tripolar seam
    const int ifd = nx - 1 - i;

    // Every load of the column is issued up front, unconditionally, with neighbours that do not exist clamped to the cell itself
    // (their values are never used: the same tests as before decide): one memory round trip instead of the four dependent ones
    // of a load-test-load chain.  These cells are 1 row in ny, but their waves were the slowest of the pass (3.4 % of it at 1 degree).
    const i64 cS = (LS >= 0) ? LS : L, cN = (LNq >= 0) ? LNq : L, cA = (LA >= 0) ? LA : L, cB = (LB >= 0) ? LB : L;
    const i64 s2 = (i64)j * nx + i;
    const i64 s2W = (i64)j * nx + iw, s2E = (i64)j * nx + ie, s2S = (LS >= 0) ? s2 - nx : s2;
    const i64 s2N = (LNq >= 0) ? (fold ? (i64)j * nx + ifd : s2 + nx) : s2;
    const i64 lEc = p.lw[LEc], lWc = p.lw[LWc], lS_ = p.lw[cS], lN_ = p.lw[cN], lA_ = p.lw[cA], lB_ = p.lw[cB];
    double gEc, gWc, gNq, gS, gA, gB, qW, qE, qS, qN, qB, qT;
    if (!p.fused) {
        const double *phiN_in = fold ? p.phi[OTMB_NORTH] : p.phi[OTMB_SOUTH];  // through the seam the north neighbour pushes with its NORTH flux
        gEc = p.phi[OTMB_WEST][LEc]; gWc = p.phi[OTMB_EAST][LWc]; gNq = phiN_in[cN]; gS = p.phi[OTMB_NORTH][cS];
        gA = p.phi[OTMB_BOTTOM][cA]; gB = p.phi[OTMB_TOP][cB];
        qW = p.phi[OTMB_WEST][L]; qE = p.phi[OTMB_EAST][L]; qS = p.phi[OTMB_SOUTH][L]; qN = p.phi[OTMB_NORTH][L];
        qB = p.phi[OTMB_BOTTOM][L]; qT = p.phi[OTMB_TOP][L];
    } else {
        // (fused step, nx >= 3) the same twelve values as facefluxes would have stored them: ϕeast[x] = x's transport if x and its east
        // neighbour are wet, ϕnorth[x] likewise with its north (or fold) neighbour, NaN / fill -> 0; ϕwest / ϕsouth are those of the west /
        // south cell; ϕbottom[x] = ϕtop of the cell below (0 at the sea floor level), ϕtop from memory.  c is wet.
        auto U = [&](i64 x) { return p.fused == 2 ? (double)((const float *)p.umo)[x] : ((const double *)p.umo)[x]; };
        auto V = [&](i64 x) { return p.fused == 2 ? (double)((const float *)p.vmo)[x] : ((const double *)p.vmo)[x]; };
        const bool wEc = lEc != 0, wWc = lWc != 0, wS_ = (LS >= 0) && lS_ != 0, wN_ = (LNq >= 0) && lN_ != 0;
        const double eC = wEc ? tm_replace(U(L), p.fillv) : 0.0;           // ϕeast[c]
        const double eW = wWc ? tm_replace(U(LWc), p.fillv) : 0.0;         // ϕeast[W] (W's east neighbour is c)
        const double nC = wN_ ? tm_replace(V(L), p.fillv) : 0.0;           // ϕnorth[c] (north or fold neighbour wet)
        const double nS = wS_ ? tm_replace(V(cS), p.fillv) : 0.0;          // ϕnorth[S] (S's north neighbour is c: S is never on the seam row)
        const double nF = (fold && wN_) ? tm_replace(V(cN), p.fillv) : 0.0;  // ϕnorth[fold cell] (its fold neighbour is c)
        gEc = eC;                     // ϕwest[E] = ϕeast[i₋₁(E)] = ϕeast[c]
        gWc = eW;
        gNq = fold ? nF : nC;         // ϕnorth[fold cell] through the seam, else ϕsouth[N] = ϕnorth[c]
        gS = nS;
        const double topC = p.phi[OTMB_TOP][L];
        gA = topC;                    // ϕbottom[A] = ϕtop[c]
        gB = p.phi[OTMB_TOP][cB];     // ϕtop[B]
        qW = eW; qE = eC; qS = (LS >= 0) ? nS : 0.0; qN = nC;
        qB = (LB >= 0) ? p.phi[OTMB_TOP][LB] : 0.0;  // ϕbottom[c]
        qT = topC;
    }
    const double vc = p.v[L], vEc = p.v[LEc], vWc = p.v[LWc], vS_ = p.v[cS], vN_ = p.v[cN], vA_ = p.v[cA], vB_ = p.v[cB];
    const double rc = p.rho ? p.rho[L] : p.rho_s;
    const double rEc = p.rho ? p.rho[LEc] : p.rho_s, rWc = p.rho ? p.rho[LWc] : p.rho_s, rS_ = p.rho ? p.rho[cS] : p.rho_s,
                 rN_ = p.rho ? p.rho[cN] : p.rho_s, rA_ = p.rho ? p.rho[cA] : p.rho_s, rB_ = p.rho ? p.rho[cB] : p.rho_s;
    const double thc = p.thk[L], tEc = p.thk[LEc], tWc = p.thk[LWc], tS_ = p.thk[cS], tN_ = p.thk[cN];
    const double *edgeNc = fold ? p.edge[OTMB_DIR_NORTH] : p.edge[OTMB_DIR_SOUTH];  // oppdir (:407): the neighbour's facing edge through the seam is its NORTH edge
    const double *distNc = fold ? p.dist[OTMB_DIR_NORTH] : p.dist[OTMB_DIR_SOUTH];
    const double eW_c = p.edge[OTMB_DIR_WEST][s2], eE_c = p.edge[OTMB_DIR_EAST][s2], eS_c = p.edge[OTMB_DIR_SOUTH][s2], eN_c = p.edge[OTMB_DIR_NORTH][s2];
    const double dW_c = p.dist[OTMB_DIR_WEST][s2], dE_c = p.dist[OTMB_DIR_EAST][s2], dS_c = p.dist[OTMB_DIR_SOUTH][s2], dN_c = p.dist[OTMB_DIR_NORTH][s2];
    const double eE_w = p.edge[OTMB_DIR_EAST][s2W], dE_w = p.dist[OTMB_DIR_EAST][s2W], eW_e = p.edge[OTMB_DIR_WEST][s2E], dW_e = p.dist[OTMB_DIR_WEST][s2E];
    const double eN_s = p.edge[OTMB_DIR_NORTH][s2S], dN_s = p.dist[OTMB_DIR_NORTH][s2S], eX_n = edgeNc[s2N], dX_n = distNc[s2N];
    const double ar = p.area[s2], mld = p.ml[s2];
    const double ztk = p.zt[k], zta_ = p.zt[k > 0 ? k - 1 : k], ztb_ = p.zt[k + 1 < nz ? k + 1 : k];

    const i64 xEc = lEc, xWc = lWc;
    const i64 xS = (LS >= 0) ? lS_ : 0, xNq = (LNq >= 0) ? lN_ : 0;
    const i64 xA = (LA >= 0) ? lA_ : 0, xB = (LB >= 0) ? lB_ : 0;

    // ---- advective fluxes pushed towards this cell by its neighbours (:244-296) -------------
    // emitter EC pushes its west flux, WC its east flux, N-side its south flux, the fold and
    // S-side cells their north flux, the cell above its bottom flux, the cell below its top flux.
    const double fEc = xEc ? sel_pos(gEc, up) : 0.0;
    const double fWc = xWc ? sel_neg(gWc, up) : 0.0;
    const double fNq = xNq ? (fold ? sel_neg(gNq, up) : sel_pos(gNq, up)) : 0.0;
    const double fS = xS ? sel_neg(gS, up) : 0.0;
    const double fA = xA ? sel_pos(gA, up) : 0.0;
    const double fB = xB ? sel_neg(gB, up) : 0.0;  // emitter has k+1 > 1 (:290)
    const bool aEc = nonzero(fEc), aWc = nonzero(fWc), aNq = nonzero(fNq), aS = nonzero(fS), aA = nonzero(fA),
               aB = nonzero(fB);

    // own pushes (:244-296) must land in a wet cell: the reference indexes Lwet3D[C𝑗] / pushes 𝑗 without
    // testing it, so a non-zero selected flux towards land or `nothing` throws there
    {
        const double ow = sel_pos(qW, up), oe = sel_neg(qE, up);
        const double os = sel_pos(qS, up), on = sel_neg(qN, up);
        const double ob = sel_pos(qB, up), ot = (k > 0) ? sel_neg(qT, up) : 0.0;
        const bool bad = (nonzero(ow) && xWc == 0) || (nonzero(oe) && xEc == 0) || (nonzero(os) && xS == 0) ||
                         (nonzero(on) && xNq == 0) || (nonzero(ob) && xB == 0) || (nonzero(ot) && xA == 0);
        if (bad) raise_flag(p.flags, FLAG_FLUX_INTO_LAND);
    }

#pragma unroll
    for (int s = 0; s < NSLOT; ++s) { col.idx[s] = 0; col.adv[s] = 0; col.hh[s] = 0; col.ml[s] = 0; col.dp[s] = 0; }
    col.padv = col.phh = col.pml = col.pdp = 0;

    // canonical slot of each row-mate (cells of the same (j,k) row can coincide when nx <= 2 or
    // on the fold: north neighbour of (nx/2) is (nx/2+1), of the centre of an odd row itself)
    const int cEC = (ie == i) ? S_SELF : S_EC;
    const int cWC = (iw == i) ? S_SELF : ((iw == ie) ? S_EC : S_WC);
    const int cFQ = (ifd == i) ? S_SELF : ((ifd == ie) ? S_EC : ((ifd == iw) ? cWC : S_FQ));

    col.idx[S_A] = xA; col.idx[S_S] = xS; col.idx[S_SELF] = c; col.idx[S_B] = xB;
    col.idx[S_EC] = (cEC == S_EC) ? xEc : 0;
    col.idx[S_WC] = (cWC == S_WC) ? xWc : 0;
    col.idx[S_FQ] = (fold && cFQ == S_FQ) ? xNq : 0;
    col.idx[S_N] = fold ? 0 : xNq;

    // order of the rows inside the column: A, S, row-mates by i, N, B
    {
        const unsigned lo = (1u << S_A) | (1u << S_S);
        const unsigned mates = (1u << S_SELF) | (1u << S_EC) | (1u << S_WC) | (1u << S_FQ);
        col.bef[S_A] = 0;
        col.bef[S_S] = 1u << S_A;
        col.bef[S_SELF] = lo | ((ie < i) ? 1u << S_EC : 0) | ((iw < i) ? 1u << S_WC : 0) | ((ifd < i) ? 1u << S_FQ : 0);
        col.bef[S_EC] = lo | ((i < ie) ? 1u << S_SELF : 0) | ((iw < ie) ? 1u << S_WC : 0) | ((ifd < ie) ? 1u << S_FQ : 0);
        col.bef[S_WC] = lo | ((i < iw) ? 1u << S_SELF : 0) | ((ie < iw) ? 1u << S_EC : 0) | ((ifd < iw) ? 1u << S_FQ : 0);
        col.bef[S_FQ] = lo | ((i < ifd) ? 1u << S_SELF : 0) | ((ie < ifd) ? 1u << S_EC : 0) | ((iw < ifd) ? 1u << S_WC : 0);
        col.bef[S_N] = lo | mates;
        col.bef[S_B] = lo | mates | (1u << S_N);
    }

    if (isnan(rc)) raise_flag(p.flags, FLAG_RHO_NAN);  // :233

    // emission order of the three row-mate emitters: ascending (i of emitter, direction W<E<S<N)
    const int kE = ie * 4 + 0, kW = iw * 4 + 1, kF = fold ? ifd * 4 + 3 : 0x7fffffff;
    const int rE = (kW < kE) + (kF < kE), rW = (kE < kW) + (kF < kW), rF = (kE < kF) + (kW < kF);

    // ---- Tadv (pushTadvectionvalues!, :193-204): entries (row e, -ϕ/(ρ̄ v_e)), (row c, ϕ/(ρ̄ v_c)) ----
    {
        bool anynan = false;
#define ADV_VALUES(ACTIVE, RX, VX, PHI, OFF, DG)                      \
    double OFF = 0.0, DG = 0.0;                                       \
    if (ACTIVE) {                                                     \
        adv_pair((PHI), (RX), rc, (VX), vc, OFF, DG);                 \
        anynan |= isnan(OFF) | isnan(DG);                             \
    }
        ADV_VALUES(aA, rA_, vA_, fA, oA, dA)
        ADV_VALUES(aS, rS_, vS_, -fS, oS, dS)
        ADV_VALUES(aEc, rEc, vEc, fEc, oEc, dEc)
        ADV_VALUES(aWc, rWc, vWc, -fWc, oWc, dWc)
        const double phNq = fold ? -fNq : fNq;
        ADV_VALUES(aNq, rN_, vN_, phNq, oNq, dNq)
        ADV_VALUES(aB, rB_, vB_, -fB, oB, dB)
#undef ADV_VALUES
        if (anynan) raise_flag(p.flags, FLAG_TADV_NAN);  // :39
        if (aA) { acc(col.adv[S_A], col.padv, S_A, oA); acc(col.adv[S_SELF], col.padv, S_SELF, dA); }
        if (aS) { acc(col.adv[S_S], col.padv, S_S, oS); acc(col.adv[S_SELF], col.padv, S_SELF, dS); }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (rE == r) {
                if (aEc) { acc_rowmate(col.adv, col.padv, cEC, oEc); acc(col.adv[S_SELF], col.padv, S_SELF, dEc); }
            } else if (rW == r) {
                if (aWc) { acc_rowmate(col.adv, col.padv, cWC, oWc); acc(col.adv[S_SELF], col.padv, S_SELF, dWc); }
            } else if (rF == r) {
                if (fold && aNq) { acc_rowmate(col.adv, col.padv, cFQ, oNq); acc(col.adv[S_SELF], col.padv, S_SELF, dNq); }
            }
        }
        if (!fold && aNq) { acc(col.adv[S_N], col.padv, S_N, oNq); acc(col.adv[S_SELF], col.padv, S_SELF, dNq); }
        if (aB) { acc(col.adv[S_B], col.padv, S_B, oB); acc(col.adv[S_SELF], col.padv, S_SELF, dB); }
    }

    // ---- TκH (:348-415, pushTmixingvalues! :426-435) ------------------------------------------
    // For each horizontal neighbour X: a = min(thk_c*edge[c->X][c], thk_X*edge[X->c][X]) is shared by
    // c's own push towards X (+Tval on the diagonal) and X's push towards c (-Tval' on row X).
    {
        bool anynan = false;
        double ownW = 0, ownE = 0, ownS = 0, ownN = 0, inW = 0, inE = 0, inS = 0, inN = 0;
#define H_VALUES(WET, TX, VX, E_C, E_X, D_C, D_X, OWN, IN)                         \
    if (WET) {                                                                     \
        h_pair(p.kH, thc, (E_C), (TX), (E_X), (D_C), vc, (D_X), (VX), OWN, IN);     \
        anynan |= isnan(OWN) | isnan(IN);                                          \
    }
        H_VALUES(xWc != 0, tWc, vWc, eW_c, eE_w, dW_c, dE_w, ownW, inW)
        H_VALUES(xEc != 0, tEc, vEc, eE_c, eW_e, dE_c, dW_e, ownE, inE)
        H_VALUES(xS != 0, tS_, vS_, eS_c, eN_s, dS_c, dN_s, ownS, inS)
        // oppdir (:407): through the seam the neighbour's facing edge is its NORTH edge (edgeNc / distNc above)
        H_VALUES(xNq != 0, tN_, vN_, eN_c, eX_n, dN_c, dX_n, ownN, inN)
#undef H_VALUES
        if (anynan) raise_flag(p.flags, FLAG_TKH_NAN);  // :61
        // own pushes, direction order W, E, S, N: (c,c,+Tval); the second push (c,X,-Tval) lands in
        // this column only when X is c itself
        if (xWc) { acc(col.hh[S_SELF], col.phh, S_SELF, ownW); if (cWC == S_SELF) acc(col.hh[S_SELF], col.phh, S_SELF, -ownW); }
        if (xEc) { acc(col.hh[S_SELF], col.phh, S_SELF, ownE); if (cEC == S_SELF) acc(col.hh[S_SELF], col.phh, S_SELF, -ownE); }
        if (xS) { acc(col.hh[S_SELF], col.phh, S_SELF, ownS); }
        if (xNq) { acc(col.hh[S_SELF], col.phh, S_SELF, ownN); if (fold && cFQ == S_SELF) acc(col.hh[S_SELF], col.phh, S_SELF, -ownN); }
        // neighbours' second pushes (X,c,-Tval'), per row in the emitter's direction order
        if (xS) acc(col.hh[S_S], col.phh, S_S, -inS);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (rE == r) {
                if (xEc && cEC != S_SELF) acc_rowmate(col.hh, col.phh, cEC, -inE);
            } else if (rW == r) {
                if (xWc && cWC != S_SELF) acc_rowmate(col.hh, col.phh, cWC, -inW);
            } else if (rF == r) {
                if (fold && xNq && cFQ != S_SELF) acc_rowmate(col.hh, col.phh, cFQ, -inN);
            }
        }
        if (!fold && xNq) acc(col.hh[S_N], col.phh, S_N, -inN);
    }

    // ---- TκVML / TκVdeep (:450-477) -------------------------------------------------------------
    {
        const bool omc = ztk < mld;  // Ω (:85); NaN (missing) compares false
        bool nanml = false, nandp = false;
        if (xB) {  // from bottom (own push first, B then T)
            const double ztb = ztb_;
            const double d = fabs(ztk - ztb);
            double ownD, inD;
            v_pair(p.kDeep * ar, d, vc, vB_, ownD, inD);
            nandp |= isnan(ownD) | isnan(inD);
            acc(col.dp[S_SELF], col.pdp, S_SELF, ownD);
            acc(col.dp[S_B], col.pdp, S_B, -inD);
            if (omc && (ztb < mld)) {
                double ownM, inM;
                v_pair(p.kML * ar, d, vc, vB_, ownM, inM);
                nanml |= isnan(ownM) | isnan(inM);
                acc(col.ml[S_SELF], col.pml, S_SELF, ownM);
                acc(col.ml[S_B], col.pml, S_B, -inM);
            }
        }
        if (xA) {
            const double zta = zta_;
            const double d = fabs(ztk - zta);
            double ownD, inD;
            v_pair(p.kDeep * ar, d, vc, vA_, ownD, inD);
            nandp |= isnan(ownD) | isnan(inD);
            acc(col.dp[S_SELF], col.pdp, S_SELF, ownD);
            acc(col.dp[S_A], col.pdp, S_A, -inD);
            if (omc && (zta < mld)) {
                double ownM, inM;
                v_pair(p.kML * ar, d, vc, vA_, ownM, inM);
                nanml |= isnan(ownM) | isnan(inM);
                acc(col.ml[S_SELF], col.pml, S_SELF, ownM);
                acc(col.ml[S_A], col.pml, S_A, -inM);
            }
        }
        if (nanml) raise_flag(p.flags, FLAG_TKVML_NAN);    // :90
        if (nandp) raise_flag(p.flags, FLAG_TKVDEEP_NAN);  // :114
    }
}

// The 58 values of a regular cell's stencil (what fast_column loads), wherever they came from.
struct Stencil {
    i64 lE, lW, lS, lN, lA, lB;     // Lwet3D of the six neighbours (unmasked)
    double gE, gW, gS, gN, gA, gB;  // the flux each neighbour pushes with: ϕwest[E], ϕeast[W], ϕnorth[S], ϕsouth[N], ϕbottom[A], ϕtop[B]
    double vC, vE, vW, vS, vN, vA, vB;
    double rC, rE, rW, rS, rN, rA, rB;
    double tC, tE, tW, tS, tN;
    double eW_c, eE_c, eS_c, eN_c, dW_c, dE_c, dS_c, dN_c, eE_w, dE_w, eW_e, dW_e, eN_s, dN_s, eS_n, dS_n, ar, mld;
    double ztk, zta, ztb;
    double hg[5];  // (HREAD) the first five entries of the given TκH's column c, in its row order: S, row-mates by index, N -- those that exist
};

// The regular-cell arithmetic on a Stencil -- THE one copy of it (src/matrixbuilding.jl:193-204, :244-296, :348-415, :426-435,
// :450-477): fast_column fills the Stencil with loads.
// Accumulators start at -0.0: (-0.0) + x == x bit for bit for every x, which is exactly sparse()'s "first touch copies, later
// ones add" without tracking the first touch.  The ρ-NaN check (:233) shares the (rarely taken) branch of the Tadv NaN
// check when the fill pass does it (p.rho_in_fill): a separate branch right after the loads splits the scheduling region and
// cost 6 % of the kernel (collecting ALL error checks into one branch at the end measured 2 % slower than this).
#define NEG0 (-0.0)
// q-th of five values, q known at run time (value selects only: an indexed access would put the array into scratch memory)
__device__ __forceinline__ double pick5(const double (&g)[5], unsigned q) {
    return q == 0 ? g[0] : (q == 1 ? g[1] : (q == 2 ? g[2] : (q == 3 ? g[3] : g[4])));
}
template <bool HREAD = false>
__device__ __forceinline__ void column_compute(const TmParams &p, const Stencil &s, int i, int j, int k, i64 c, Column &col) {
    const int nx = p.nx, ny = p.ny, nz = p.nz, up = p.upwind;
    const bool hS = j > 0, hN = j + 1 < ny, hA = k > 0, hB = k + 1 < nz;
    const i64 xE = s.lE, xW = s.lW, xS = hS ? s.lS : 0, xN = hN ? s.lN : 0, xA = hA ? s.lA : 0, xB = hB ? s.lB : 0;
    const bool wE = xE != 0, wW = xW != 0, wS = xS != 0, wN = xN != 0, wA = xA != 0, wB = xB != 0;
    const double vC = s.vC, rC = s.rC, tC = s.tC;

    // ---- advective pushes towards this cell (:244-296) ----
    const double fE = wE ? sel_pos(s.gE, up) : 0.0;  // east cell pushes its west flux
    const double fW = wW ? sel_neg(s.gW, up) : 0.0;  // west cell pushes its east flux
    const double fS = wS ? sel_neg(s.gS, up) : 0.0;  // south cell pushes its north flux
    const double fN = wN ? sel_pos(s.gN, up) : 0.0;  // north cell pushes its south flux
    const double fA = wA ? sel_pos(s.gA, up) : 0.0;  // cell above pushes its bottom flux
    const double fB = wB ? sel_neg(s.gB, up) : 0.0;  // cell below pushes its top flux (its k > 1, :290)
    const bool aE = nonzero(fE), aW = nonzero(fW), aS = nonzero(fS), aN = nonzero(fN), aA = nonzero(fA), aB = nonzero(fB);

    // row order of the column: A, S, row-mates by i, N, B.  Row-mates: W, SELF, E -- except at the
    // periodic wrap (i == 0: SELF, E, W(nx-1);  i == nx-1: E(0), W, SELF)
    const bool wrap0 = (i == 0), wrap1 = (i == nx - 1), swapWE = wrap0 | wrap1;
    {
        const unsigned lo = (1u << S_A) | (1u << S_S), bS = 1u << S_SELF, bE = 1u << S_EC, bW = 1u << S_WC;
        col.bef[S_A] = 0;
        col.bef[S_S] = 1u << S_A;
        col.bef[S_WC] = lo | (wrap0 ? (bS | bE) : (wrap1 ? bE : 0u));
        col.bef[S_SELF] = lo | (wrap0 ? 0u : (wrap1 ? (bE | bW) : bW));
        col.bef[S_EC] = lo | (wrap0 ? bS : (wrap1 ? 0u : (bW | bS)));
        col.bef[S_FQ] = 0;
        col.bef[S_N] = lo | bS | bE | bW;
        col.bef[S_B] = lo | bS | bE | bW | (1u << S_N);
    }
    col.idx[S_A] = xA; col.idx[S_S] = xS; col.idx[S_SELF] = c; col.idx[S_EC] = xE; col.idx[S_WC] = xW;
    col.idx[S_FQ] = 0; col.idx[S_N] = xN; col.idx[S_B] = xB;

    // ---- Tadv (pushTadvectionvalues!, :193-204) ----
    {
        double oA_, dA_, oS_, dS_, oW_, dW_, oE_, dE_, oN_, dN_, oB_, dB_;
        adv_pair(fA, s.rA, rC, s.vA, vC, oA_, dA_);
        adv_pair(-fS, s.rS, rC, s.vS, vC, oS_, dS_);
        adv_pair(-fW, s.rW, rC, s.vW, vC, oW_, dW_);
        adv_pair(fE, s.rE, rC, s.vE, vC, oE_, dE_);
        adv_pair(fN, s.rN, rC, s.vN, vC, oN_, dN_);
        adv_pair(-fB, s.rB, rC, s.vB, vC, oB_, dB_);
        const bool bad = (aA & (isnan(oA_) | isnan(dA_))) | (aS & (isnan(oS_) | isnan(dS_))) | (aW & (isnan(oW_) | isnan(dW_))) |
                         (aE & (isnan(oE_) | isnan(dE_))) | (aN & (isnan(oN_) | isnan(dN_))) | (aB & (isnan(oB_) | isnan(dB_)));
        const bool badrho = p.rho_in_fill && isnan(rC);
        if (bad | badrho) {
            if (badrho) raise_flag(p.flags, FLAG_RHO_NAN);  // :233
            if (bad) raise_flag(p.flags, FLAG_TADV_NAN);    // :39
        }
        // diagonal: contributions in ascending emitter index = A, S, row-mates by i, N, B
        double d = NEG0;
        d += aA ? dA_ : NEG0;
        d += aS ? dS_ : NEG0;
        const double m1 = swapWE ? (aE ? dE_ : NEG0) : (aW ? dW_ : NEG0);
        const double m2 = swapWE ? (aW ? dW_ : NEG0) : (aE ? dE_ : NEG0);
        d += m1;
        d += m2;
        d += aN ? dN_ : NEG0;
        d += aB ? dB_ : NEG0;
        col.adv[S_A] = oA_; col.adv[S_S] = oS_; col.adv[S_WC] = oW_; col.adv[S_EC] = oE_; col.adv[S_N] = oN_;
        col.adv[S_B] = oB_; col.adv[S_SELF] = d; col.adv[S_FQ] = 0;
        col.padv = ((unsigned)aA << S_A) | ((unsigned)aS << S_S) | ((unsigned)aW << S_WC) | ((unsigned)aE << S_EC) |
                   ((unsigned)aN << S_N) | ((unsigned)aB << S_B) | ((unsigned)(aA | aS | aW | aE | aN | aB) << S_SELF);
    }
    // ---- TκH (:348-415, :426-435); oppdir = south away from the seam row (:407) ----
    if (HREAD) {
        // the given matrix's own values: its column holds the derived rows -- S, row-mates in index order, N: those whose cell is wet, and the
        // diagonal iff any of them is (exactly col.phh / col.bef below, which the comparing pass verified against its colptr / rowval)
        const unsigned any = (unsigned)(wW | wE | wS | wN);
        unsigned q = 0;
        const unsigned qS = q; q += wS;
        // row-mates: W, SELF, E -- at the periodic wrap SELF, E, W (i == 0) or E, W, SELF (i == nx - 1)
        const unsigned first_is_E = wrap1, self_first = wrap0;
        unsigned qW, qC, qE;
        if (first_is_E) { qE = q; q += wE; qW = q; q += wW; qC = q; q += any; }
        else if (self_first) { qC = q; q += any; qE = q; q += wE; qW = q; q += wW; }
        else { qW = q; q += wW; qC = q; q += any; qE = q; q += wE; }
        const unsigned qN = q;
        col.hh[S_S] = pick5(s.hg, qS); col.hh[S_WC] = pick5(s.hg, qW); col.hh[S_SELF] = pick5(s.hg, qC); col.hh[S_EC] = pick5(s.hg, qE);
        col.hh[S_N] = pick5(s.hg, qN);
        col.hh[S_A] = 0; col.hh[S_B] = 0; col.hh[S_FQ] = 0;
        col.phh = ((unsigned)wW << S_WC) | ((unsigned)wE << S_EC) | ((unsigned)wS << S_S) | ((unsigned)wN << S_N) | (any << S_SELF);
    } else {
        double ownW, inW, ownE, inE, ownS, inS, ownN, inN;
        h_pair(p.kH, tC, s.eW_c, s.tW, s.eE_w, s.dW_c, vC, s.dE_w, s.vW, ownW, inW);
        h_pair(p.kH, tC, s.eE_c, s.tE, s.eW_e, s.dE_c, vC, s.dW_e, s.vE, ownE, inE);
        h_pair(p.kH, tC, s.eS_c, s.tS, s.eN_s, s.dS_c, vC, s.dN_s, s.vS, ownS, inS);
        h_pair(p.kH, tC, s.eN_c, s.tN, s.eS_n, s.dN_c, vC, s.dS_n, s.vN, ownN, inN);
        const bool bad = (wW & (isnan(ownW) | isnan(inW))) | (wE & (isnan(ownE) | isnan(inE))) |
                         (wS & (isnan(ownS) | isnan(inS))) | (wN & (isnan(ownN) | isnan(inN)));
        if (bad) raise_flag(p.flags, FLAG_TKH_NAN);  // :61
        double h = NEG0;  // own pushes in direction order W, E, S, N
        h += wW ? ownW : NEG0;
        h += wE ? ownE : NEG0;
        h += wS ? ownS : NEG0;
        h += wN ? ownN : NEG0;
        col.hh[S_SELF] = h; col.hh[S_WC] = -inW; col.hh[S_EC] = -inE; col.hh[S_S] = -inS; col.hh[S_N] = -inN;
        col.hh[S_A] = 0; col.hh[S_B] = 0; col.hh[S_FQ] = 0;
        col.phh = ((unsigned)wW << S_WC) | ((unsigned)wE << S_EC) | ((unsigned)wS << S_S) | ((unsigned)wN << S_N) |
                  ((unsigned)(wW | wE | wS | wN) << S_SELF);
    }
    // ---- TκVdeep / TκVML (:450-477) ----
    {
        const double ztk = s.ztk, zta = hA ? s.zta : s.ztk, ztb = hB ? s.ztb : s.ztk;
        const double dB = fabs(ztk - ztb), dA = fabs(ztk - zta);
        const double nD = p.kDeep * s.ar;
        double ownB, inB, ownA, inA;
        v_pair(nD, dB, vC, s.vB, ownB, inB);
        v_pair(nD, dA, vC, s.vA, ownA, inA);
        if ((wB & (isnan(ownB) | isnan(inB))) | (wA & (isnan(ownA) | isnan(inA)))) raise_flag(p.flags, FLAG_TKVDEEP_NAN);  // :114
        double d = NEG0;  // own pushes: bottom then top
        d += wB ? ownB : NEG0;
        d += wA ? ownA : NEG0;
        col.dp[S_SELF] = d; col.dp[S_B] = -inB; col.dp[S_A] = -inA;
        col.pdp = ((unsigned)wB << S_B) | ((unsigned)wA << S_A) | ((unsigned)(wA | wB) << S_SELF);
        const bool omC = ztk < s.mld;  // Ω (:85); NaN compares false
        const bool mB = wB & omC & (ztb < s.mld), mA = wA & omC & (zta < s.mld);
        col.pml = 0;
        col.ml[S_SELF] = 0; col.ml[S_A] = 0; col.ml[S_B] = 0;
        if (mA | mB) {
            const double nM = p.kML * s.ar;
            double mownB, minB, mownA, minA;
            v_pair(nM, dB, vC, s.vB, mownB, minB);
            v_pair(nM, dA, vC, s.vA, mownA, minA);
            if ((mB & (isnan(mownB) | isnan(minB))) | (mA & (isnan(mownA) | isnan(minA)))) raise_flag(p.flags, FLAG_TKVML_NAN);  // :90
            double m = NEG0;
            m += mB ? mownB : NEG0;
            m += mA ? mownA : NEG0;
            col.ml[S_SELF] = m; col.ml[S_B] = -minB; col.ml[S_A] = -minA;
            col.pml = ((unsigned)mB << S_B) | ((unsigned)mA << S_A) | ((unsigned)(mA | mB) << S_SELF);
        }
    }
}

// ---- fast path ---------------------------------------------------------------------------------
// Regular cells: nx >= 3 and not on the tripolar seam row, i.e. the W/E/S/N/A/B neighbours are
// distinct cells.  Same arithmetic as build_column, organised for the hardware:
//  * every load is unconditional and issued up front (addresses depend on (i,j,k) only; a
//    neighbour that does not exist is clamped to the cell itself and masked afterwards), so one
//    memory round trip covers the whole stencil instead of one per `if`;
//  * 32-bit byte offsets from tile-uniform base pointers (scalar base + vector offset loads);
//  * accumulators start at -0.0: (-0.0) + x == x bit for bit for every x, which is exactly
//    sparse()'s "first touch copies, later ones add" without tracking the first touch.
struct TileBase {  // array pointers advanced to the tile's lowest neighbour (uniform per workgroup)
    const char *lw, *v, *thk, *rho, *pe, *pw, *pn, *ps, *pt, *pb, *mk;
    const char *pu, *pv;  // (fused step) umo / vmo, advanced likewise -- in THEIR element size
};
// (fused step) one mass transport value as Float64 (Array{Float64}(umo), :125-126), by byte offset in units of 8-byte elements
template <int FUSED> __device__ __forceinline__ double ld_uv(const char *b, unsigned off8) {
    return FUSED == 2 ? (double)__builtin_nontemporal_load((const float *)(b + (off8 >> 1))) : __builtin_nontemporal_load((const double *)(b + off8));
}
__device__ __forceinline__ double ldd(const char *b, unsigned byteoff) { return *(const double *)(b + byteoff); }
__device__ __forceinline__ double ldv(const char *b, unsigned byteoff) { return ldd(b, byteoff); }
__device__ __forceinline__ i64 ldi(const char *b, unsigned byteoff) { return *(const i64 *)(b + byteoff); }
// The two input checks that need no arithmetic (own pushes land in wet cells, ρ[c] is not NaN) are made with the counts
// (fast_presence / facefluxes_kernel<COUNTS>), not here.
// Returns whether Lwet3D holds c at the cell itself (the canonical-indices check, loaded with the stencil).
// HREAD: TκH's values come from the given matrix (p.hx at the column's offset hq) instead of thkcello and the 2-D metrics.
template <int FUSED = 0, bool HREAD = false>
__device__ __forceinline__ bool fast_column(const TmParams &p, const TileBase &tb, unsigned oC, int i, int j, int k,
                                            i64 c, Column &col, Stamps &st, i64 hq = 0) {
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const bool hS = j > 0, hN = j + 1 < ny, hA = k > 0, hB = k + 1 < nz;
    const int di_e = (i + 1 < nx) ? 1 : 1 - nx, di_w = (i > 0) ? -1 : nx - 1;
    const unsigned nx8 = (unsigned)nx * 8u, P8 = (unsigned)p.P * 8u;
    const unsigned oE = oC + (unsigned)(di_e * 8), oW = oC + (unsigned)(di_w * 8);
    const unsigned oS = hS ? oC - nx8 : oC, oN = hN ? oC + nx8 : oC;
    const unsigned oA = hA ? oC - P8 : oC, oB = hB ? oC + P8 : oC;
    const unsigned s2 = ((unsigned)j * (unsigned)nx + (unsigned)i) * 8u;
    const unsigned sE = s2 + (unsigned)(di_e * 8), sW = s2 + (unsigned)(di_w * 8);
    const unsigned sS = hS ? s2 - nx8 : s2, sN = hN ? s2 + nx8 : s2;

    // ---- all loads ----
    const i64 lC = ldi(tb.lw, oC), lE = ldi(tb.lw, oE), lW = ldi(tb.lw, oW), lS = ldi(tb.lw, oS), lN = ldi(tb.lw, oN),
              lA = ldi(tb.lw, oA), lB = ldi(tb.lw, oB);
    // each of the six flux arrays is read ONCE per cell (at one neighbour): streaming (non-temporal) loads, so that these lines do not
    // displace the v3D / ρ / Lwet3D lines that five neighbours share (A/B over several array placements: -3 % at 0.25 degree)
#define LDPHI(b, o) __builtin_nontemporal_load((const double *)((b) + (o)))
    double gE0, gW0, gS0, gN0, gA0, gB0;
    if (FUSED == 0) {
        gE0 = LDPHI(tb.pw, oE); gW0 = LDPHI(tb.pe, oW); gS0 = LDPHI(tb.pn, oS); gN0 = LDPHI(tb.ps, oN);
        gA0 = LDPHI(tb.pb, oA); gB0 = LDPHI(tb.pt, oB);
    } else {
        // the raw transports; masked and replaced below, once the wet ranks that travel with them are back
        gE0 = ld_uv<FUSED>(tb.pu, oC);  // ϕwest[E] = ϕeast[c]   (:206-211)
        gW0 = ld_uv<FUSED>(tb.pu, oW);  // ϕeast[W]
        gS0 = ld_uv<FUSED>(tb.pv, oS);  // ϕnorth[S]
        gN0 = ld_uv<FUSED>(tb.pv, oC);  // ϕsouth[N] = ϕnorth[c] (:219-224)
        gA0 = LDPHI(tb.pt, oC);         // ϕbottom[A] = ϕtop[c]  (:238-240)
        gB0 = LDPHI(tb.pt, oB);         // ϕtop[B]
    }
    const double vC = ldv(tb.v, oC), vE = ldv(tb.v, oE), vW = ldv(tb.v, oW), vS = ldv(tb.v, oS), vN = ldv(tb.v, oN),
                 vA = ldv(tb.v, oA), vB = ldv(tb.v, oB);
    // (a scalar ρ is filled in AFTER the last load is issued: assigning it here makes the compiler drain the loads above first)
    double rC = 0, rE = 0, rW = 0, rS = 0, rN = 0, rA = 0, rB = 0;
    if (tb.rho) {
        rC = ldv(tb.rho, oC); rS = ldv(tb.rho, oS); rN = ldv(tb.rho, oN); rA = ldv(tb.rho, oA); rB = ldv(tb.rho, oB);
        rE = ldv(tb.rho, oE); rW = ldv(tb.rho, oW);
    }
    double hg0 = 0, hg1 = 0, hg2 = 0, hg3 = 0, hg4 = 0;
    if (HREAD) {  // five consecutive entries from the column's first one (clamped to the array: entries beyond the column are never picked)
        const i64 last = p.hnnz - 1;
        const i64 q0 = hq < last ? hq : last, q1 = hq + 1 < last ? hq + 1 : last, q2 = hq + 2 < last ? hq + 2 : last, q3 = hq + 3 < last ? hq + 3 : last,
                  q4 = hq + 4 < last ? hq + 4 : last;
        hg0 = p.hx[q0]; hg1 = p.hx[q1]; hg2 = p.hx[q2]; hg3 = p.hx[q3]; hg4 = p.hx[q4];
    }
    double tC = 0, tE = 0, tW = 0, tS = 0, tN = 0;
    double eW_c = 0, eE_c = 0, eS_c = 0, eN_c = 0, dW_c = 0, dE_c = 0, dS_c = 0, dN_c = 0, eE_w = 0, dE_w = 0, eW_e = 0, dW_e = 0, eN_s = 0, dN_s = 0, eS_n = 0, dS_n = 0;
    if (!HREAD) {
        tC = ldv(tb.thk, oC); tE = ldv(tb.thk, oE); tW = ldv(tb.thk, oW); tS = ldv(tb.thk, oS);
        tN = ldv(tb.thk, oN);
        const char *eWp = (const char *)p.edge[OTMB_DIR_WEST], *eEp = (const char *)p.edge[OTMB_DIR_EAST],
                   *eSp = (const char *)p.edge[OTMB_DIR_SOUTH], *eNp = (const char *)p.edge[OTMB_DIR_NORTH];
        const char *dWp = (const char *)p.dist[OTMB_DIR_WEST], *dEp = (const char *)p.dist[OTMB_DIR_EAST],
                   *dSp = (const char *)p.dist[OTMB_DIR_SOUTH], *dNp = (const char *)p.dist[OTMB_DIR_NORTH];
        eW_c = ldv(eWp, s2); eE_c = ldv(eEp, s2); eS_c = ldv(eSp, s2); eN_c = ldv(eNp, s2);
        dW_c = ldv(dWp, s2); dE_c = ldv(dEp, s2); dS_c = ldv(dSp, s2); dN_c = ldv(dNp, s2);
        eE_w = ldv(eEp, sW); dE_w = ldv(dEp, sW);  // west cell's east edge / distance to its east nbr
        eW_e = ldv(eWp, sE); dW_e = ldv(dWp, sE);
        eN_s = ldv(eNp, sS); dN_s = ldv(dNp, sS);
        eS_n = ldv(eSp, sN); dS_n = ldv(dSp, sN);  // oppdir = south away from the seam row (:407)
    }
    const double ar = ldv((const char *)p.area, s2), mld = ldd((const char *)p.ml, s2);
    // zt[k-1], zt[k], zt[k+1]: k is (nearly) uniform in a wave, so the four levels around the wave's first k come through the
    // scalar cache instead of three more vector loads; a wave that spans more than two levels (tiny grids) takes vector loads
    const int k0w = __builtin_amdgcn_readfirstlane(k);
    double ztk, zta, ztb;
    if (__builtin_amdgcn_ballot_w64(k - k0w > 1 || k < k0w) == 0) {
        const double z0 = p.zt[k0w > 0 ? k0w - 1 : 0], z1 = p.zt[k0w], z2 = p.zt[k0w + 1 < nz ? k0w + 1 : nz - 1],
                     z3 = p.zt[k0w + 2 < nz ? k0w + 2 : nz - 1];
        const bool lower = k != k0w;  // k == k0w + 1
        ztk = lower ? z2 : z1;
        zta = hA ? (lower ? z1 : z0) : ztk;
        ztb = hB ? (lower ? z3 : z2) : ztk;
    } else {
        ztk = p.zt[k]; zta = p.zt[hA ? k - 1 : k]; ztb = p.zt[hB ? k + 1 : k];
    }

    STAMP(st, 2, 1);  // every stencil load is back
    if (!tb.rho) rC = rE = rW = rS = rN = rA = rB = p.rho_s;
    if (FUSED != 0) {
        // what facefluxes stores: a flux between two wet cells is the transport with NaN / fill replaced by zero, any other is zero
        // (nofluxboundaries! on the emitting cell, :161-175; c itself is wet)
        gE0 = (lE != 0) ? tm_replace(gE0, p.fillv) : 0.0;
        gW0 = (lW != 0) ? tm_replace(gW0, p.fillv) : 0.0;
        gS0 = (hS && lS != 0) ? tm_replace(gS0, p.fillv) : 0.0;
        gN0 = (hN && lN != 0) ? tm_replace(gN0, p.fillv) : 0.0;
    }
    // the stencil as values, then THE arithmetic
    Stencil s;
    s.lE = lE; s.lW = lW; s.lS = lS; s.lN = lN; s.lA = lA; s.lB = lB;
    s.gE = gE0; s.gW = gW0; s.gS = gS0; s.gN = gN0; s.gA = gA0; s.gB = gB0;
    s.vC = vC; s.vE = vE; s.vW = vW; s.vS = vS; s.vN = vN; s.vA = vA; s.vB = vB;
    s.rC = rC; s.rE = rE; s.rW = rW; s.rS = rS; s.rN = rN; s.rA = rA; s.rB = rB;
    s.tC = tC; s.tE = tE; s.tW = tW; s.tS = tS; s.tN = tN;
    s.eW_c = eW_c; s.eE_c = eE_c; s.eS_c = eS_c; s.eN_c = eN_c; s.dW_c = dW_c; s.dE_c = dE_c; s.dS_c = dS_c; s.dN_c = dN_c;
    s.eE_w = eE_w; s.dE_w = dE_w; s.eW_e = eW_e; s.dW_e = dW_e; s.eN_s = eN_s; s.dN_s = dN_s; s.eS_n = eS_n; s.dS_n = dS_n;
    s.ar = ar; s.mld = mld; s.ztk = ztk; s.zta = zta; s.ztb = ztb;
    s.hg[0] = hg0; s.hg[1] = hg1; s.hg[2] = hg2; s.hg[3] = hg3; s.hg[4] = hg4;
    column_compute<HREAD>(p, s, i, j, k, c, col);
    return lC == c;
}

// Presence only (regular cells): which rows the four operator matrices hold in this column -- a function
// of the wet mask, the sign tests on the six incoming fluxes and the mixed-layer mask, no arithmetic.  The
// wet bits and the sign tests come from the push mask (2 bytes per neighbour, see otmb_push_bits) instead
// of Lwet3D and the six ϕ arrays.  The union is an upper bound of T's rows (map(+) drops only exact-zero sums).
__device__ __forceinline__ unsigned ldm(const char *b, unsigned byteoff) { return *(const uint16_t *)(b + byteoff); }
__device__ __forceinline__ void fast_presence(const TmParams &p, const TileBase &tb, unsigned oC, int i, int j, int k,
                                              unsigned &padv, unsigned &phh, unsigned &pml, unsigned &pdp) {
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const bool hS = j > 0, hN = j + 1 < ny, hA = k > 0, hB = k + 1 < nz;
    const int di_e = (i + 1 < nx) ? 1 : 1 - nx, di_w = (i > 0) ? -1 : nx - 1;
    const unsigned mC_o = oC >> 2, nx2 = (unsigned)nx * 2u, P2 = (unsigned)p.P * 2u;  // mask elements are 2 bytes
    const unsigned sh = p.upwind ? 0u : 8u;
    const unsigned mC = ldm(tb.mk, mC_o) >> sh, mE = ldm(tb.mk, mC_o + (unsigned)(di_e * 2)) >> sh,
                   mW = ldm(tb.mk, mC_o + (unsigned)(di_w * 2)) >> sh, mS = ldm(tb.mk, hS ? mC_o - nx2 : mC_o) >> sh,
                   mN = ldm(tb.mk, hN ? mC_o + nx2 : mC_o) >> sh, mA = ldm(tb.mk, hA ? mC_o - P2 : mC_o) >> sh,
                   mB = ldm(tb.mk, hB ? mC_o + P2 : mC_o) >> sh;
    const unsigned s2 = ((unsigned)j * (unsigned)nx + (unsigned)i) * 8u;
    const double mld = ldd((const char *)p.ml, s2);
    const double ztk = p.zt[k], zta = p.zt[hA ? k - 1 : k], ztb = p.zt[hB ? k + 1 : k];
    const bool wE = mE & PM_WET, wW = mW & PM_WET, wS = hS && (mS & PM_WET), wN = hN && (mN & PM_WET), wA = hA && (mA & PM_WET),
               wB = hB && (mB & PM_WET);
    {   // the two input checks that need no arithmetic: own pushes land in wet cells (the reference indexes
        // Lwet3D[C𝑗] unconditionally, :247 etc.) and ρ is not NaN on wet cells (:233)
        const bool bad = ((mC & PM_W) && !wW) | ((mC & PM_E) && !wE) | ((mC & PM_S) && !wS) | ((mC & PM_N) && !wN) |
                         ((mC & PM_B) && !wB) | (hA && (mC & PM_T) && !wA);
        if (bad) raise_flag(p.flags, FLAG_FLUX_INTO_LAND);
        if (!p.rho_in_fill && tb.rho && isnan(ldd(tb.rho, oC))) raise_flag(p.flags, FLAG_RHO_NAN);
    }
    // the east cell pushes through its west face, the cell above through its bottom face, ... (:244-296)
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

// Presence only, ANY cell (tripolar seam row, nx <= 2): build_column's pattern logic -- the canonical slots of row-mates
// that coincide, the fold's north neighbour living in the cell's own row -- on the push mask alone.  Seven 2-byte loads
// in one round trip instead of build_column's chain of dependent loads and divisions; the values are the fill pass's
// business.
__device__ __forceinline__ void general_presence(const TmParams &p, const Cell &cell, unsigned &padv, unsigned &phh, unsigned &pml,
                                                 unsigned &pdp) {
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int i = cell.i, j = cell.j, k = cell.k;
    const int ie = (i + 1 < nx) ? i + 1 : 0, iw = (i > 0) ? i - 1 : nx - 1;
    const i64 L = cell.L, LEc = cell.row0 + ie, LWc = cell.row0 + iw;
    const i64 LS = nb_jm1(cell, nx), LNq = nb_jp1(cell, nx, ny, p.topo);
    const i64 LA = nb_km1(cell, p.P), LB = nb_kp1(cell, nz, p.P);
    const bool fold = (j == ny - 1) && (LNq >= 0);
    const int ifd = nx - 1 - i;
    const unsigned sh = p.upwind ? 0u : 8u;
    const unsigned mC = (unsigned)p.mask[L] >> sh, mE = (unsigned)p.mask[LEc] >> sh, mW = (unsigned)p.mask[LWc] >> sh,
                   mS = (unsigned)p.mask[LS >= 0 ? LS : L] >> sh, mN = (unsigned)p.mask[LNq >= 0 ? LNq : L] >> sh,
                   mA = (unsigned)p.mask[LA >= 0 ? LA : L] >> sh, mB = (unsigned)p.mask[LB >= 0 ? LB : L] >> sh;
    const double mld = p.ml[(i64)j * nx + i];
    const double ztk = p.zt[k], zta = p.zt[k > 0 ? k - 1 : k], ztb = p.zt[k + 1 < nz ? k + 1 : k];
    const bool wE = mE & PM_WET, wW = mW & PM_WET, wS = (LS >= 0) && (mS & PM_WET), wN = (LNq >= 0) && (mN & PM_WET),
               wA = (LA >= 0) && (mA & PM_WET), wB = (LB >= 0) && (mB & PM_WET);
    {
        const bool bad = ((mC & PM_W) && !wW) | ((mC & PM_E) && !wE) | ((mC & PM_S) && !wS) | ((mC & PM_N) && !wN) |
                         ((mC & PM_B) && !wB) | ((k > 0) && (mC & PM_T) && !wA);
        if (bad) raise_flag(p.flags, FLAG_FLUX_INTO_LAND);
        if (!p.rho_in_fill && p.rho && isnan(p.rho[L])) raise_flag(p.flags, FLAG_RHO_NAN);
    }
    // through the seam the north neighbour pushes with its own NORTH flux (:271-278 seen from the other side)
    const bool aE = wE && (mE & PM_W), aW = wW && (mW & PM_E), aS = wS && (mS & PM_N), aN = wN && (mN & (fold ? PM_N : PM_S));
    const bool aA = wA && (mA & PM_B), aB = wB && (mB & PM_T);
    const int cEC = (ie == i) ? S_SELF : S_EC;
    const int cWC = (iw == i) ? S_SELF : ((iw == ie) ? S_EC : S_WC);
    const int cFQ = (ifd == i) ? S_SELF : ((ifd == ie) ? S_EC : ((ifd == iw) ? cWC : S_FQ));
    const int cN = fold ? cFQ : S_N;
    padv = ((unsigned)aA << S_A) | ((unsigned)aS << S_S) | ((unsigned)aE << cEC) | ((unsigned)aW << cWC) | ((unsigned)aN << cN) |
           ((unsigned)aB << S_B) | ((unsigned)(aA | aS | aW | aE | aN | aB) << S_SELF);
    phh = ((unsigned)wS << S_S) | ((unsigned)wE << cEC) | ((unsigned)wW << cWC) | ((unsigned)wN << cN) |
          ((unsigned)(wW | wE | wS | wN) << S_SELF);
    pdp = ((unsigned)wB << S_B) | ((unsigned)wA << S_A) | ((unsigned)(wA | wB) << S_SELF);
    const bool omC = ztk < mld;
    const bool mlB = wB & omC & (ztb < mld), mlA = wA & omC & (zta < mld);
    pml = ((unsigned)mlB << S_B) | ((unsigned)mlA << S_A) | ((unsigned)(mlA | mlB) << S_SELF);
}

// Values of a given operator read where they lie: the given column holds the derived column's rows in the derived order (the comparing
// pass verified it), so slot s's entry is the popc(present & bef[s])-th of the column, which starts at entry q0 of x.
template <unsigned SLOTS>
__device__ __forceinline__ void given_values(double (&val)[NSLOT], unsigned pres, const unsigned (&bef)[NSLOT], const double *__restrict__ x, i64 q0, i64 nnz) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        if (!((SLOTS >> s) & 1u)) continue;
        i64 q = q0 + (i64)__popc(pres & bef[s]);
        q = q < nnz ? q : nnz - 1;  // (slots that are absent are never used; a verified column never reaches the clamp)
        const double g = x[q];
        val[s] = ((pres >> s) & 1u) ? g : val[s];
    }
}

// T[r,c] = ((Tadv + TκH) + TκVML) + TκVdeep, absent operand = +0.0 (:147, map(+) semantics)
__device__ __forceinline__ double t_value(const Column &col, int s) {
    const double a = ((col.padv >> s) & 1u) ? col.adv[s] : 0.0;
    const double h = ((col.phh >> s) & 1u) ? col.hh[s] : 0.0;
    const double m = ((col.pml >> s) & 1u) ? col.ml[s] : 0.0;
    const double d = ((col.pdp >> s) & 1u) ? col.dp[s] : 0.0;
    return ((a + h) + m) + d;
}

