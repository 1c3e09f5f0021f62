// otmb_transportmatrix.hip -- fused assembly of (T, Tadv, TκH, TκVML, TκVdeep) in CSC.
//
// Replaces, on the device, the whole of `transportmatrix` (src/matrixbuilding.jl:128-150):
// the three COO generators (:221-299, :337-418, :438-479), the four sparse() calls
// (:41,63,92,116) and the three sparse adds (:147).
//
// Gather formulation.  The reference scatters: wet cell 𝑖 pushes triplets into its own column
// and its neighbours' columns, then sparse() sorts and sums.  Every triplet of column c comes
// from c itself or from one of the <= 7 cells whose neighbour (in some direction) is c, so one
// thread per grid cell rebuilds its own column of all five matrices directly:
//   * which triplets land in column c, and in which order the reference emits them
//     (ascending emitting wet index, then W,E,S,N,B,T, then first/second push) is a pure
//     function of the local stencil -- duplicates are summed left-to-right in that order,
//     first touch copies the value (sparse() keeps explicit zeros and -0.0);
//   * T[r,c] = ((Tadv + TκH) + TκVML) + TκVdeep with absent operands +0.0, stored iff != 0
//     (SparseArrays' map(+) drops exact zeros);
//   * rows of a column ascend in wet index == linear index (makeindices is monotone; this is
//     verified on the fly and reported as OTMB_ERR_NONCANONICAL_INDICES otherwise).
// Two passes over the grid: COUNT (per-tile nnz of the five matrices + wet count) -> tile scan
// -> FILL (recompute, block-scan for in-tile offsets, write colptr/rowval/nzval).  No COO is
// ever materialised.  All arithmetic is Float64 with contraction off (-ffp-contract=off) so
// each value is bit-identical to the reference's expression.
#include "otmb_common.h"
#include "otmb_topology.h"

#define TM_THREADS 256
#define TM_CHUNKS 4
#define TM_TILE (TM_THREADS * TM_CHUNKS)
#define TM_NF 6  // T, Tadv, TκH, TκVML, TκVdeep, wet

struct TmParams {
    const double *phi[6];
    const double *v, *thk, *rho;
    double rho_s;
    const i64 *lw;
    const double *edge[4], *dist[4];
    const double *area, *zt, *ml;
    double kH, kML, kDeep;
    int nx, ny, nz, topo, upwind;
    i64 P, G;
    // outputs (FILL)
    i64 *colptr[5], *rowval[5];
    double *nzval[5];
    // scan state
    uint32_t *tilesums;    // [ntiles][TM_NF]  (COUNT writes)
    const i64 *tileoffs;   // [ntiles][TM_NF]  (FILL reads)
    int *flags;
};

struct TmPlan {
    otmb_tm_args args;  // device pointers
    i64 ntiles;
    i64 nnz[5];
    bool valid;
};

// slots of a column: the cells that can hold a row of column c
enum { S_A = 0, S_S = 1, S_SELF = 2, S_EC = 3, S_WC = 4, S_FQ = 5, S_N = 6, S_B = 7, NSLOT = 8 };

struct Column {
    i64 idx[NSLOT];      // wet rank of the slot's cell (row index), 0 = no such wet cell
    double adv[NSLOT];   // Tadv values
    double hh[NSLOT];    // TκH values   (slots SELF, EC, WC, FQ, S, N)
    double ml[NSLOT];    // TκVML values (slots SELF, A, B)
    double dp[NSLOT];    // TκVdeep values
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

// Build the column of wet cell `cell` (c = own wet rank, > 0), or -- for a land cell (c == 0) --
// only check that no wet neighbour pushes a non-zero flux into it.
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

    const i64 xEc = p.lw[LEc], xWc = p.lw[LWc];
    const i64 xS = (LS >= 0) ? p.lw[LS] : 0, xNq = (LNq >= 0) ? p.lw[LNq] : 0;
    const i64 xA = (LA >= 0) ? p.lw[LA] : 0, xB = (LB >= 0) ? p.lw[LB] : 0;

    // ---- advective fluxes pushed towards this cell by its neighbours (:244-296) -------------
    // emitter EC pushes its west flux, WC its east flux, N-side its south flux, the fold and
    // S-side cells their north flux, the cell above its bottom flux, the cell below its top flux.
    const double fEc = xEc ? sel_pos(p.phi[OTMB_WEST][LEc], up) : 0.0;
    const double fWc = xWc ? sel_neg(p.phi[OTMB_EAST][LWc], up) : 0.0;
    const double fNq = xNq ? (fold ? sel_neg(p.phi[OTMB_NORTH][LNq], up) : sel_pos(p.phi[OTMB_SOUTH][LNq], up)) : 0.0;
    const double fS = xS ? sel_neg(p.phi[OTMB_NORTH][LS], up) : 0.0;
    const double fA = xA ? sel_pos(p.phi[OTMB_BOTTOM][LA], up) : 0.0;
    const double fB = xB ? sel_neg(p.phi[OTMB_TOP][LB], up) : 0.0;  // emitter has k+1 > 1 (:290)
    const bool aEc = nonzero(fEc), aWc = nonzero(fWc), aNq = nonzero(fNq), aS = nonzero(fS), aA = nonzero(fA),
               aB = nonzero(fB);

    if (c == 0) {
        // land: the reference would index Lwet3D with `missing` for any of these pushes
        if (aEc | aWc | aNq | aS | aA | aB) raise_flag(p.flags, FLAG_FLUX_INTO_LAND);
        return;
    }
    // own pushes towards `nothing` (closed south/north/bottom boundaries)
    {
        bool bad = false;
        if (j == 0) bad |= nonzero(sel_pos(p.phi[OTMB_SOUTH][L], up));
        if (j == ny - 1 && LNq < 0) bad |= nonzero(sel_neg(p.phi[OTMB_NORTH][L], up));
        if (k == nz - 1) bad |= nonzero(sel_pos(p.phi[OTMB_BOTTOM][L], up));
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

    const double vc = p.v[L];
    const double rc = p.rho ? p.rho[L] : p.rho_s;
    if (isnan(rc)) raise_flag(p.flags, FLAG_RHO_NAN);  // :233

    // emission order of the three row-mate emitters: ascending (i of emitter, direction W<E<S<N)
    const int kE = ie * 4 + 0, kW = iw * 4 + 1, kF = fold ? ifd * 4 + 3 : 0x7fffffff;
    const int rE = (kW < kE) + (kF < kE), rW = (kE < kW) + (kF < kW), rF = (kE < kF) + (kW < kF);

    // ---- Tadv (pushTadvectionvalues!, :193-204): entries (row e, -ϕ/(ρ̄ v_e)), (row c, ϕ/(ρ̄ v_c)) ----
    {
        bool anynan = false;
#define ADV_VALUES(ACTIVE, LX, PHI, OFF, DG)                          \
    double OFF = 0.0, DG = 0.0;                                       \
    if (ACTIVE) {                                                     \
        const double rx_ = p.rho ? p.rho[LX] : p.rho_s;               \
        const double rb_ = (rx_ + rc) / 2;                            \
        const double mx_ = rb_ * p.v[LX];                             \
        const double mc_ = rb_ * vc;                                  \
        OFF = -(PHI) / mx_;                                           \
        DG = (PHI) / mc_;                                             \
        anynan |= isnan(OFF) | isnan(DG);                             \
    }
        ADV_VALUES(aA, LA, fA, oA, dA)
        ADV_VALUES(aS, LS, -fS, oS, dS)
        ADV_VALUES(aEc, LEc, fEc, oEc, dEc)
        ADV_VALUES(aWc, LWc, -fWc, oWc, dWc)
        const double phNq = fold ? -fNq : fNq;
        ADV_VALUES(aNq, LNq, phNq, oNq, dNq)
        ADV_VALUES(aB, LB, -fB, oB, dB)
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
        const i64 s = (i64)j * nx + i;
        const double thc = p.thk[L];
        bool anynan = false;
        double ownW = 0, ownE = 0, ownS = 0, ownN = 0, inW = 0, inE = 0, inS = 0, inN = 0;
#define H_VALUES(WET, LX, SX, DCX, EDGE_XC, DIST_XC, OWN, IN)                      \
    if (WET) {                                                                     \
        const i64 sx_ = (SX);                                                      \
        const double aij_ = thc * p.edge[DCX][s];                                  \
        const double aji_ = p.thk[LX] * (EDGE_XC)[sx_];                            \
        const double a_ = jl_min(aij_, aji_);                                      \
        OWN = (p.kH * a_) / (p.dist[DCX][s] * vc);                                 \
        IN = (p.kH * a_) / ((DIST_XC)[sx_] * p.v[LX]);                             \
        anynan |= isnan(OWN) | isnan(IN);                                          \
    }
        H_VALUES(xWc != 0, LWc, (i64)j * nx + iw, OTMB_DIR_WEST, p.edge[OTMB_DIR_EAST], p.dist[OTMB_DIR_EAST], ownW, inW)
        H_VALUES(xEc != 0, LEc, (i64)j * nx + ie, OTMB_DIR_EAST, p.edge[OTMB_DIR_WEST], p.dist[OTMB_DIR_WEST], ownE, inE)
        H_VALUES(xS != 0, LS, s - nx, OTMB_DIR_SOUTH, p.edge[OTMB_DIR_NORTH], p.dist[OTMB_DIR_NORTH], ownS, inS)
        // oppdir (:407): through the seam the neighbour's facing edge is its NORTH edge.  Pointer
        // selects (not p.edge[runtime]) keep the kernel arguments out of scratch memory.
        const double *edgeNc = fold ? p.edge[OTMB_DIR_NORTH] : p.edge[OTMB_DIR_SOUTH];
        const double *distNc = fold ? p.dist[OTMB_DIR_NORTH] : p.dist[OTMB_DIR_SOUTH];
        H_VALUES(xNq != 0, LNq, fold ? (i64)j * nx + ifd : s + nx, OTMB_DIR_NORTH, edgeNc, distNc, ownN, inN)
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
        const i64 s = (i64)j * nx + i;
        const double ar = p.area[s];
        const double ztk = p.zt[k];
        const double mld = p.ml[s];
        const bool omc = ztk < mld;  // Ω (:85); NaN (missing) compares false
        bool nanml = false, nandp = false;
        if (xB) {  // from bottom (own push first, B then T)
            const double ztb = p.zt[k + 1];
            const double d = fabs(ztk - ztb);
            const double ownD = (p.kDeep * ar) / (d * vc), inD = (p.kDeep * ar) / (d * p.v[LB]);
            nandp |= isnan(ownD) | isnan(inD);
            acc(col.dp[S_SELF], col.pdp, S_SELF, ownD);
            acc(col.dp[S_B], col.pdp, S_B, -inD);
            if (omc && (ztb < mld)) {
                const double ownM = (p.kML * ar) / (d * vc), inM = (p.kML * ar) / (d * p.v[LB]);
                nanml |= isnan(ownM) | isnan(inM);
                acc(col.ml[S_SELF], col.pml, S_SELF, ownM);
                acc(col.ml[S_B], col.pml, S_B, -inM);
            }
        }
        if (xA) {
            const double zta = p.zt[k - 1];
            const double d = fabs(ztk - zta);
            const double ownD = (p.kDeep * ar) / (d * vc), inD = (p.kDeep * ar) / (d * p.v[LA]);
            nandp |= isnan(ownD) | isnan(inD);
            acc(col.dp[S_SELF], col.pdp, S_SELF, ownD);
            acc(col.dp[S_A], col.pdp, S_A, -inD);
            if (omc && (zta < mld)) {
                const double ownM = (p.kML * ar) / (d * vc), inM = (p.kML * ar) / (d * p.v[LA]);
                nanml |= isnan(ownM) | isnan(inM);
                acc(col.ml[S_SELF], col.pml, S_SELF, ownM);
                acc(col.ml[S_A], col.pml, S_A, -inM);
            }
        }
        if (nanml) raise_flag(p.flags, FLAG_TKVML_NAN);    // :90
        if (nandp) raise_flag(p.flags, FLAG_TKVDEEP_NAN);  // :114
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

template <bool FILL>
__global__ __launch_bounds__(TM_THREADS) void tm_kernel(const TmParams p) {
    __shared__ u64 wave_tot[TM_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const i64 tile = blockIdx.x;
    i64 run[TM_NF];
#pragma unroll
    for (int f = 0; f < TM_NF; ++f) run[f] = FILL ? p.tileoffs[tile * TM_NF + f] : 0;

    for (int ch = 0; ch < TM_CHUNKS; ++ch) {
        const i64 L = tile * TM_TILE + (i64)ch * TM_THREADS + tid;
        const bool inb = L < p.G;
        Column col;
        i64 c = 0;
        unsigned pT = 0;
        if (inb) {
            const Cell cell = cell_of(L, p.nx, p.ny, p.P);
            c = p.lw[L];
            build_column(p, cell, c, col);
        }
        unsigned nT = 0, nA = 0, nH = 0, nM = 0, nD = 0;
        if (c != 0) {
            const unsigned uni = col.padv | col.phh | col.pml | col.pdp;
#pragma unroll
            for (int s = 0; s < NSLOT; ++s)
                if (((uni >> s) & 1u) && t_value(col, s) != 0.0) pT |= 1u << s;
            nT = __popc(pT); nA = __popc(col.padv); nH = __popc(col.phh); nM = __popc(col.pml); nD = __popc(col.pdp);
        }
        // packed block scan: T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10 | wet:9 bits
        const u64 mine = (u64)nT | ((u64)nA << 11) | ((u64)nH << 22) | ((u64)nM << 33) | ((u64)nD << 43) |
                         ((u64)(c != 0) << 53);
        u64 incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            u64 y = __shfl_up(incl, d);
            if (lane >= d) incl += y;
        }
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
        u64 before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < TM_THREADS / 64; ++w) {
            const u64 v = wave_tot[w];
            if (w < wid) before += v;
            all += v;
        }
        __syncthreads();
        const u64 excl = before + incl - mine;
        if (FILL && c != 0) {
            const i64 off[5] = {run[0] + (i64)(excl & 0x7ff), run[1] + (i64)((excl >> 11) & 0x7ff),
                                run[2] + (i64)((excl >> 22) & 0x7ff), run[3] + (i64)((excl >> 33) & 0x3ff),
                                run[4] + (i64)((excl >> 43) & 0x3ff)};
            const i64 wetrank = run[5] + (i64)((excl >> 53) & 0x1ff);  // wet cells before this one
            if (c != wetrank + 1) raise_flag(p.flags, FLAG_NONCANONICAL);
            else {
#pragma unroll
                for (int m = 0; m < 5; ++m) p.colptr[m][c - 1] = off[m] + 1;
                const unsigned pm[5] = {pT, col.padv, col.phh, col.pml, col.pdp};
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    const i64 row = col.idx[s];
                    if ((pT >> s) & 1u) {
                        const i64 q = off[0] + __popc(pT & col.bef[s]);
                        p.rowval[0][q] = row;
                        p.nzval[0][q] = t_value(col, s);
                    }
                    if ((pm[1] >> s) & 1u) {
                        const i64 q = off[1] + __popc(pm[1] & col.bef[s]);
                        p.rowval[1][q] = row;
                        p.nzval[1][q] = col.adv[s];
                    }
                    if ((pm[2] >> s) & 1u) {
                        const i64 q = off[2] + __popc(pm[2] & col.bef[s]);
                        p.rowval[2][q] = row;
                        p.nzval[2][q] = col.hh[s];
                    }
                    if ((pm[3] >> s) & 1u) {
                        const i64 q = off[3] + __popc(pm[3] & col.bef[s]);
                        p.rowval[3][q] = row;
                        p.nzval[3][q] = col.ml[s];
                    }
                    if ((pm[4] >> s) & 1u) {
                        const i64 q = off[4] + __popc(pm[4] & col.bef[s]);
                        p.rowval[4][q] = row;
                        p.nzval[4][q] = col.dp[s];
                    }
                }
            }
        }
        run[0] += (i64)(all & 0x7ff);
        run[1] += (i64)((all >> 11) & 0x7ff);
        run[2] += (i64)((all >> 22) & 0x7ff);
        run[3] += (i64)((all >> 33) & 0x3ff);
        run[4] += (i64)((all >> 43) & 0x3ff);
        run[5] += (i64)((all >> 53) & 0x1ff);
    }
    if (!FILL && tid == 0) {
#pragma unroll
        for (int f = 0; f < TM_NF; ++f) p.tilesums[tile * TM_NF + f] = (uint32_t)run[f];
    }
}

__global__ void tm_finish_colptr(i64 *c0, i64 *c1, i64 *c2, i64 *c3, i64 *c4, const i64 *tot, i64 N) {
    if (threadIdx.x == 0) {
        c0[N] = tot[0] + 1; c1[N] = tot[1] + 1; c2[N] = tot[2] + 1; c3[N] = tot[3] + 1; c4[N] = tot[4] + 1;
    }
}

// ---- host side ------------------------------------------------------------------------------
static void fill_params(TmParams &p, const otmb_tm_args &a, otmb_ctx *ctx) {
    memset(&p, 0, sizeof p);
    for (int f = 0; f < 6; ++f) p.phi[f] = a.phi[f];
    p.v = a.v3d; p.thk = a.thkcello; p.rho = a.rho; p.rho_s = a.rho_scalar; p.lw = a.lwet3d;
    for (int d = 0; d < 4; ++d) { p.edge[d] = a.edge_length[d]; p.dist[d] = a.dist_nbr[d]; }
    p.area = a.area2d; p.zt = a.zt; p.ml = a.mlotst;
    p.kH = a.kappa_h; p.kML = a.kappa_vml; p.kDeep = a.kappa_vdeep;
    p.nx = (int)a.nx; p.ny = (int)a.ny; p.nz = (int)a.nz; p.topo = a.topology; p.upwind = a.upwind;
    p.P = a.nx * a.ny; p.G = p.P * a.nz;
    p.tilesums = (uint32_t *)ctx->blocksums.p;
    p.tileoffs = (const i64 *)ctx->blockoffs.p;
    p.flags = (int *)ctx->flags.p;
}

static int32_t check_flags(otmb_ctx *ctx) {
    const int *f = ctx->h_flags;
    if (f[FLAG_NONCANONICAL]) return otmb_fail(ctx, OTMB_ERR_NONCANONICAL_INDICES);
    if (f[FLAG_RHO_NAN]) return otmb_fail(ctx, OTMB_ERR_RHO_NAN);  // reference order: :233, loop, :39, :61, :90, :114
    if (f[FLAG_FLUX_INTO_LAND]) return otmb_fail(ctx, OTMB_ERR_FLUX_INTO_LAND);
    if (f[FLAG_TADV_NAN]) return otmb_fail(ctx, OTMB_ERR_TADV_NAN);
    if (f[FLAG_TKH_NAN]) return otmb_fail(ctx, OTMB_ERR_TKH_NAN);
    if (f[FLAG_TKVML_NAN]) return otmb_fail(ctx, OTMB_ERR_TKVML_NAN);
    if (f[FLAG_TKVDEEP_NAN]) return otmb_fail(ctx, OTMB_ERR_TKVDEEP_NAN);
    return OTMB_OK;
}

void otmb_tm_plan_free(otmb_ctx *ctx) {
    delete ctx->plan;
    ctx->plan = nullptr;
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
    if (a->nx < 1 || a->ny < 1 || a->nz < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    const i64 G = a->nx * a->ny * a->nz;
    if (a->nx * a->ny >= (1ll << 31) || G >= (1ll << 40)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid too large");
    if (a->topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);
    if (a->topology != OTMB_BIPOLAR && a->topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "topology");
    for (int f = 0; f < 6; ++f)
        if (!a->phi[f]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "phi");
    for (int d = 0; d < 4; ++d)
        if (!a->edge_length[d] || !a->dist_nbr[d]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "metrics");
    if (!a->v3d || !a->thkcello || !a->lwet3d || !a->area2d || !a->zt || !a->mlotst)
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null input array");
    if (!a->rho && a->rho_scalar != a->rho_scalar && a->n_wet > 0) return otmb_fail(ctx, OTMB_ERR_RHO_NAN);  // :233
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 ntiles = (G + TM_TILE - 1) / TM_TILE;
    int32_t rc;
    if ((rc = otmb_reserve(ctx, ctx->blocksums, (size_t)ntiles * TM_NF * sizeof(uint32_t)))) return rc;
    if ((rc = otmb_reserve(ctx, ctx->blockoffs, (size_t)ntiles * TM_NF * sizeof(i64)))) return rc;
    if (!ctx->plan) ctx->plan = new TmPlan();
    TmPlan &pl = *ctx->plan;
    pl.args = *a;
    pl.ntiles = ntiles;
    TmParams p;
    fill_params(p, *a, ctx);
    int *dflags = (int *)ctx->flags.p;
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS);
    HIP_TRY(ctx, hipMemsetAsync(dflags, 0, OTMB_NFLAGS * sizeof(int) + 16 * sizeof(i64), ctx->stream));
    {
        KernelTimer kt(ctx, K_TM_COUNT);
        hipLaunchKernelGGL(tm_kernel<false>, dim3((unsigned)ntiles), dim3(TM_THREADS), 0, ctx->stream, p);
    }
    {
        KernelTimer kt(ctx, K_TILESCAN);
        otmb_launch_tilescan(ctx->stream, p.tilesums, (i64 *)ctx->blockoffs.p, dtot, ntiles, TM_NF);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_NFLAGS * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_tot, dtot, TM_NF * sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = check_flags(ctx))) return rc;
    if (ctx->h_tot[5] != a->n_wet) {
        char msg[128];
        snprintf(msg, sizeof msg, "n_wet = %lld but Lwet3D has %lld wet cells", (long long)a->n_wet, (long long)ctx->h_tot[5]);
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, msg);
    }
    for (int m = 0; m < 5; ++m) nnz[m] = pl.nnz[m] = ctx->h_tot[m];
    pl.valid = true;
    return OTMB_OK;
}

int32_t otmb_transportmatrix_fill_dev(otmb_ctx *ctx, int64_t *const colptr[5], int64_t *const rowval[5],
                                      double *const nzval[5]) {
    if (!ctx || !colptr || !rowval || !nzval) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (!ctx->plan || !ctx->plan->valid) return otmb_fail(ctx, OTMB_ERR_NO_PLAN);
    TmPlan &pl = *ctx->plan;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    TmParams p;
    fill_params(p, pl.args, ctx);
    for (int m = 0; m < 5; ++m) {
        if (!colptr[m] || (pl.nnz[m] > 0 && (!rowval[m] || !nzval[m]))) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null output");
        p.colptr[m] = (i64 *)colptr[m]; p.rowval[m] = (i64 *)rowval[m]; p.nzval[m] = nzval[m];
    }
    int *dflags = (int *)ctx->flags.p;
    i64 *dtot = (i64 *)(dflags + OTMB_NFLAGS);
    {
        KernelTimer kt(ctx, K_TM_FILL);
        hipLaunchKernelGGL(tm_kernel<true>, dim3((unsigned)pl.ntiles), dim3(TM_THREADS), 0, ctx->stream, p);
    }
    {
        KernelTimer kt(ctx, K_TM_FINISH);
        hipLaunchKernelGGL(tm_finish_colptr, dim3(1), dim3(64), 0, ctx->stream, p.colptr[0], p.colptr[1], p.colptr[2],
                           p.colptr[3], p.colptr[4], dtot, (i64)pl.args.n_wet);
    }
    HIP_TRY(ctx, hipGetLastError());
    // the fill pass can still raise OTMB_ERR_NONCANONICAL_INDICES: otmb_ctx_synchronize reports it
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags, dflags, OTMB_NFLAGS * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return OTMB_OK;
}

}  // extern "C"
