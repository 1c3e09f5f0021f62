/*
 * otmb_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE.
 *
 * A single-threaded, plain-C restatement of the sparse transport-operator
 * assembly path of TMIP-code/OceanTransportMatrixBuilder.jl v0.8.3 (Julia).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object; the product (libotmb_hip.so) never does.
 *
 * PARITY UNPINNED: the reference is pure Julia, no Julia toolchain exists in
 * the build image, and the reference's tests hold no golden vectors for this
 * path.  Faithfulness rests on (1) this line-by-line
 * restatement, each function citing the reference file:line it follows,
 * (2) an independent pure-Python transliteration (oracle/pyref.py) that must
 * agree bit for bit, (3) scipy cross-checks of the sparse()/+ semantics and
 * (4) the physical properties the reference's own tests assert.
 *
 * Third-party arithmetic that is NOT under /root/reference and is restated
 * from its published algorithm (Project.toml compat bounds, no Manifest):
 *   - SparseArrays.sparse(I,J,V,m,n)  (Julia stdlib >= 1.10): counting sort to
 *     CSR, in-order duplicate combine with +, transpose to CSC; keeps zeros.
 *   - SparseArrays.+(A,B) = map(+,A,B): column-wise sorted merge that drops
 *     results that are exactly zero.
 *   - Distances.haversine 0.10 (radius 6371000 m).
 *
 * Conventions: arrays are column-major (nx,ny,nz), i fastest; all indices
 * handed across the API are 1-based Int64 exactly as Julia stores them;
 * "missing"/"nothing" in Lwet3D is 0.  Build: gcc -O2 -ffp-contract=off.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_RHO_NAN (-1)      /* "ρ contains NaNs"        matrixbuilding.jl:233 */
#define ORC_ERR_TADV_NAN (-2)     /* "Tadv contains NaNs."    matrixbuilding.jl:39  */
#define ORC_ERR_TKH_NAN (-3)      /* "TκH contains NaNs."     matrixbuilding.jl:61  */
#define ORC_ERR_TKVML_NAN (-4)    /* "TκVML contains NaNs."   matrixbuilding.jl:90  */
#define ORC_ERR_TKVDEEP_NAN (-5)  /* "TκVdeep contains NaNs." matrixbuilding.jl:114 */
#define ORC_ERR_FLUX_INTO_LAND (-6) /* Lwet3D[nothing] / push!(::Vector{Int}, missing) */
#define ORC_ERR_UNKNOWN_TOPOLOGY (-7) /* "Unknown grid type"  gridtopology.jl:111-116 */
#define ORC_ERR_ALL_MISSING (-8)  /* @assert velocities.jl:199-200 */
#define ORC_ERR_ALLOC (-9)

enum { ORC_BIPOLAR = 0, ORC_TRIPOLAR = 1, ORC_UNKNOWN = 2 };

typedef struct {
    int64_t nx, ny, nz;
    int32_t topo;
} orc_grid;

/* ---- grid topology: gridtopology.jl:57-68, :94 -------------------------- */
/* Cells are addressed by 0-based (i,j,k); a shift returns the 0-based linear
 * index of the neighbour or -1 for Julia's `nothing`. */
static inline int64_t lin(const orc_grid *g, int64_t i, int64_t j, int64_t k) {
    return i + g->nx * (j + g->ny * k);
}
static inline int64_t ip1(const orc_grid *g, int64_t i, int64_t j, int64_t k) { /* :57 */
    return (i + 1 < g->nx) ? lin(g, i + 1, j, k) : lin(g, 0, j, k);
}
static inline int64_t im1(const orc_grid *g, int64_t i, int64_t j, int64_t k) { /* :58 */
    return (i > 0) ? lin(g, i - 1, j, k) : lin(g, g->nx - 1, j, k);
}
static inline int64_t jp1(const orc_grid *g, int64_t i, int64_t j, int64_t k) { /* :62, :94 */
    if (j + 1 < g->ny) return lin(g, i, j + 1, k);
    if (g->topo == ORC_TRIPOLAR) return lin(g, g->nx - 1 - i, g->ny - 1, k);
    return -1;
}
static inline int64_t jm1(const orc_grid *g, int64_t i, int64_t j, int64_t k) { /* :63 */
    return (j > 0) ? lin(g, i, j - 1, k) : -1;
}
static inline int64_t kp1(const orc_grid *g, int64_t i, int64_t j, int64_t k) { /* :67 */
    return (k + 1 < g->nz) ? lin(g, i, j, k + 1) : -1;
}
static inline int64_t km1(const orc_grid *g, int64_t i, int64_t j, int64_t k) { /* :68 */
    return (k > 0) ? lin(g, i, j, k - 1) : -1;
}

/* ---- makeindices: matrixbuilding.jl:10-24 ------------------------------- */
/* Lwet (capacity G) gets the 1-based linear indices of wet cells ascending,
 * Lwet3D (G) the 1-based wet rank or 0 (missing), wet3D (G) 0/1. Returns N. */
int64_t orc_makeindices(const double *v3D, int64_t G, int64_t *Lwet, int64_t *Lwet3D,
                        uint8_t *wet3D) {
    int64_t N = 0;
    for (int64_t L = 0; L < G; ++L) {
        if (!isnan(v3D[L])) { /* :15 */
            if (Lwet) Lwet[N] = L + 1;
            ++N;
            if (Lwet3D) Lwet3D[L] = N; /* :20 */
            if (wet3D) wet3D[L] = 1;   /* :18 */
        } else {
            if (Lwet3D) Lwet3D[L] = 0; /* :19 missing */
            if (wet3D) wet3D[L] = 0;
        }
    }
    return N;
}

/* ---- nofluxboundaries!: velocities.jl:154-179 --------------------------- */
int32_t orc_nofluxboundaries(double *phi_i, double *phi_j, const uint8_t *wet3D,
                             const orc_grid *g) {
    if (g->topo == ORC_UNKNOWN) return ORC_ERR_UNKNOWN_TOPOLOGY;
    for (int64_t k = 0; k < g->nz; ++k)
        for (int64_t j = 0; j < g->ny; ++j)
            for (int64_t i = 0; i < g->nx; ++i) {
                int64_t c = lin(g, i, j, k);
                int64_t E = ip1(g, i, j, k); /* :163 */
                int64_t N = jp1(g, i, j, k); /* :164 */
                if (!wet3D[c]) { phi_i[c] = 0; phi_j[c] = 0; }     /* :167 */
                if (E < 0 || !wet3D[E]) phi_i[c] = 0;              /* :170 */
                if (N < 0 || !wet3D[N]) phi_j[c] = 0;              /* :173 */
            }
    return ORC_OK;
}

/* Julia isequal(x, fill) on Float64: bitwise-equal, or both NaN. */
static inline int isequal_f64(double a, double b) {
    if (isnan(a) && isnan(b)) return 1;
    uint64_t ua, ub;
    memcpy(&ua, &a, 8);
    memcpy(&ub, &b, 8);
    return ua == ub;
}

/* ---- facefluxes: velocities.jl:190-255 ---------------------------------- */
/* umo/vmo are the caller's Float64 COPIES (facefluxesfrommasstransport
 * converts first, :125-126) and are mutated in place as the reference does.
 * Outputs: six (nx,ny,nz) arrays. */
/* top_below / flags: NOT in the reference -- test support for the depth-slab (multi-GPU) path.  A slab
 * holds levels [k0,k1) of a deeper grid; top_below (nx*ny or NULL) is ϕtop of level k1, which is what
 * ϕbottom of the slab's last level equals in the whole-grid recurrence (:240).  With flags != NULL the
 * :199-200 assertion is not raised; flags[0..1] report whether umo / vmo hold any valid value. */
int32_t orc_facefluxes_slab(double *umo, double *vmo, const uint8_t *wet3D, double fill,
                            const orc_grid *g, double *east, double *west, double *north,
                            double *south, double *top, double *bottom, const double *top_below,
                            int32_t *flags) {
    int32_t rc = orc_nofluxboundaries(umo, vmo, wet3D, g); /* :192 */
    if (rc) return rc;
    const int64_t G = g->nx * g->ny * g->nz;
    int all_u = 1, all_v = 1; /* :199-200 */
    for (int64_t c = 0; c < G; ++c) {
        if (!(isnan(umo[c]) || umo[c] == fill)) all_u = 0;
        if (!(isnan(vmo[c]) || vmo[c] == fill)) all_v = 0;
    }
    if (flags) { flags[0] = !all_u; flags[1] = !all_v; }
    else if (all_u || all_v) return ORC_ERR_ALL_MISSING;
    for (int64_t c = 0; c < G; ++c) { /* :203, :215 replace(NaN=>0, Fill=>0) */
        east[c] = (isnan(umo[c]) || isequal_f64(umo[c], fill)) ? 0.0 : umo[c];
        north[c] = (isnan(vmo[c]) || isequal_f64(vmo[c], fill)) ? 0.0 : vmo[c];
    }
    for (int64_t k = 0; k < g->nz; ++k)
        for (int64_t j = 0; j < g->ny; ++j)
            for (int64_t i = 0; i < g->nx; ++i) {
                int64_t c = lin(g, i, j, k);
                west[c] = east[im1(g, i, j, k)];       /* :206-211 (W is never nothing) */
                int64_t S = jm1(g, i, j, k);
                south[c] = (S < 0) ? 0.0 : north[S];   /* :219-224 */
            }
    const int64_t P = g->nx * g->ny;
    for (int64_t k = g->nz - 1; k >= 0; --k) { /* :236-243 */
        for (int64_t p = 0; p < P; ++p) {
            int64_t c = p + P * k;
            bottom[c] = (k == g->nz - 1) ? (top_below ? top_below[p] : 0.0) : top[c + P];
            /* @. a + b + c - d - e lowers to ((((a+b)+c)-d)-e): n-ary + folds left */
            top[c] = (((bottom[c] + west[c]) + south[c]) - east[c]) - north[c];
        }
    }
    return ORC_OK;
}

int32_t orc_facefluxes(double *umo, double *vmo, const uint8_t *wet3D, double fill,
                       const orc_grid *g, double *east, double *west, double *north,
                       double *south, double *top, double *bottom) {
    return orc_facefluxes_slab(umo, vmo, wet3D, fill, g, east, west, north, south, top, bottom, NULL, NULL);
}

/* ---- COO triplet sink ---------------------------------------------------- */
typedef struct {
    int64_t *I, *J;
    double *V;
    int64_t len;
} coo_t;
static inline void push3(coo_t *c, int64_t i, int64_t j, double v) {
    c->I[c->len] = i;
    c->J[c->len] = j;
    c->V[c->len] = v;
    ++c->len;
}

/* pushTadvectionvalues!: matrixbuilding.jl:193-204 (𝑖,𝑗 1-based wet ranks) */
static inline void push_adv(coo_t *c, int64_t wi, int64_t wj, double phi, double rho_i,
                            double rho_j, double v_i, double v_j) {
    double rho = (rho_i + rho_j) / 2; /* :194 */
    double m_i = rho * v_i;           /* :195 */
    double m_j = rho * v_j;           /* :196 */
    push3(c, wi, wj, -phi / m_i);     /* :197-199 */
    push3(c, wj, wj, phi / m_j);      /* :200-202 */
}
/* pushTmixingvalues!: matrixbuilding.jl:426-435 */
static inline void push_mix(coo_t *c, int64_t wi, int64_t wj, double kappa, double a,
                            double d, double V) {
    double Tval = kappa * a / (d * V); /* :427  (κ*a)/(d*V) */
    push3(c, wi, wi, Tval);
    push3(c, wi, wj, -Tval);
}

static inline double jl_max0(double x) { /* Julia max(x, 0): NaN propagates */
    if (isnan(x)) return x;
    return (x > 0.0) ? x : 0.0;
}
static inline double jl_min0(double x) {
    if (isnan(x)) return x;
    return (x < 0.0) ? x : 0.0;
}

/* ---- advection_operator_sparse_entries: matrixbuilding.jl:221-299 -------- */
/* phi order: [0]=east [1]=west [2]=north [3]=south [4]=top [5]=bottom.
 * rho: 3-D array, or NULL with rho_scalar (the reference fills an array, :223).
 * I,J,V must have capacity 12*N.  Returns the triplet count or an error. */
int64_t orc_advection_entries(const double *const phi[6], const double *v3D, const double *rho,
                              double rho_scalar, const int64_t *Lwet, const int64_t *Lwet3D,
                              int64_t N, const orc_grid *g, int32_t upwind, int64_t *I,
                              int64_t *J, double *V) {
    const double *east = phi[0], *west = phi[1], *north = phi[2], *south = phi[3],
                 *top = phi[4], *bottom = phi[5];
    if (g->topo == ORC_UNKNOWN && N > 0) return ORC_ERR_UNKNOWN_TOPOLOGY;
    for (int64_t w = 0; w < N; ++w) { /* :233 any(isnan, ρ[Lwet]) */
        double r = rho ? rho[Lwet[w] - 1] : rho_scalar;
        if (isnan(r)) return ORC_ERR_RHO_NAN;
    }
    coo_t c = {I, J, V, 0};
#define RHO(L) (rho ? rho[L] : rho_scalar)
#define ADV_DIR(FLUXEXPR, NBR, SIGN)                                                  \
    do {                                                                              \
        double f = (FLUXEXPR);                                                        \
        if ((f > 0) || (f < 0)) {                                                     \
            int64_t Cj = (NBR);                                                       \
            if (Cj < 0) return ORC_ERR_FLUX_INTO_LAND;                                \
            int64_t wj = Lwet3D[Cj];                                                  \
            if (wj == 0) return ORC_ERR_FLUX_INTO_LAND;                               \
            push_adv(&c, w + 1, wj, (SIGN) * f, rho_i, RHO(Cj), v_i, v3D[Cj]);        \
        }                                                                             \
    } while (0)
    for (int64_t w = 0; w < N; ++w) { /* :237 */
        int64_t L = Lwet[w] - 1;
        int64_t i = L % g->nx, j = (L / g->nx) % g->ny, k = L / (g->nx * g->ny);
        double v_i = v3D[L];
        double rho_i = RHO(L);
        ADV_DIR(upwind ? jl_max0(west[L]) : west[L] / 2, im1(g, i, j, k), 1.0);     /* :244-251 */
        ADV_DIR(upwind ? jl_min0(east[L]) : east[L] / 2, ip1(g, i, j, k), -1.0);    /* :253-260 */
        ADV_DIR(upwind ? jl_max0(south[L]) : south[L] / 2, jm1(g, i, j, k), 1.0);   /* :262-269 */
        ADV_DIR(upwind ? jl_min0(north[L]) : north[L] / 2, jp1(g, i, j, k), -1.0);  /* :271-278 */
        ADV_DIR(upwind ? jl_max0(bottom[L]) : bottom[L] / 2, kp1(g, i, j, k), 1.0); /* :280-287 */
        if (k > 0)                                                                   /* :290 */
            ADV_DIR(upwind ? jl_min0(top[L]) : top[L] / 2, km1(g, i, j, k), -1.0);  /* :289-296 */
    }
#undef ADV_DIR
#undef RHO
    return c.len;
}

/* ---- horizontal_diffusion_operator_sparse_entries: matrixbuilding.jl:337-418
 * edge[d], dist[d] are (nx,ny) arrays, d: 0=west 1=east 2=south 3=north
 * (edge_length_2D[dir], distance_to_neighbour_2D[dir]).  OmegaH: N bytes or
 * NULL for trues(N) (:56).  Capacity 8*N. */
int64_t orc_hdiff_entries(const double *v3D, const double *thk, const double *const edge[4],
                          const double *const dist[4], const int64_t *Lwet,
                          const int64_t *Lwet3D, int64_t N, const orc_grid *g, double kappaH,
                          const uint8_t *OmegaH, int64_t *I, int64_t *J, double *V) {
    enum { W = 0, E = 1, S = 2, Nn = 3 };
    if (g->topo == ORC_UNKNOWN && N > 0) return ORC_ERR_UNKNOWN_TOPOLOGY;
    coo_t c = {I, J, V, 0};
    const int64_t P = g->nx * g->ny;
    for (int64_t w = 0; w < N; ++w) { /* :348 */
        if (OmegaH && !OmegaH[w]) continue; /* :349 */
        int64_t L = Lwet[w] - 1;
        int64_t i = L % g->nx, j = (L / g->nx) % g->ny, k = L / P;
        int64_t s = i + g->nx * j; /* horizontalindex, gridcellgeometry.jl:197-198 */
        double Vv = v3D[L];
        int64_t nb[4] = {im1(g, i, j, k), ip1(g, i, j, k), jm1(g, i, j, k), jp1(g, i, j, k)};
        int opp[4] = {E, W, Nn, (j == g->ny - 1) ? Nn : S}; /* :364,378,392,407 */
        for (int d = 0; d < 4; ++d) {                        /* W, E, S, N */
            int64_t Cj = nb[d];
            if (Cj < 0) continue; /* isnothing */
            int64_t wj = Lwet3D[Cj];
            if (wj == 0) continue; /* ismissing */
            if (OmegaH && !OmegaH[wj - 1]) continue;
            int64_t sj = Cj % P;
            /* verticalfacearea = height*width, gridcellgeometry.jl:230-234 */
            double aij = thk[L] * edge[d][s];
            double aji = thk[Cj] * edge[opp[d]][sj];
            double a = fmin(aij, aji); /* NaN handled below to match Julia min */
            if (isnan(aij) || isnan(aji)) a = NAN;
            double dd = dist[d][s];
            push_mix(&c, w + 1, wj, kappaH, a, dd, Vv);
        }
    }
    return c.len;
}

/* ---- vertical_diffusion_operator_sparse_entries: matrixbuilding.jl:438-479
 * Omega: N bytes or NULL for trues(N) (:109).  Capacity 4*N. */
int64_t orc_vdiff_entries(const double *v3D, const double *area2D, const double *zt,
                          const int64_t *Lwet, const int64_t *Lwet3D, int64_t N,
                          const orc_grid *g, double kappaV, const uint8_t *Omega, int64_t *I,
                          int64_t *J, double *V) {
    if (g->topo == ORC_UNKNOWN && N > 0) return ORC_ERR_UNKNOWN_TOPOLOGY;
    coo_t c = {I, J, V, 0};
    const int64_t P = g->nx * g->ny;
    for (int64_t w = 0; w < N; ++w) { /* :450 */
        if (Omega && !Omega[w]) continue;
        int64_t L = Lwet[w] - 1;
        int64_t i = L % g->nx, j = (L / g->nx) % g->ny, k = L / P;
        double Vv = v3D[L];
        double a = area2D[i + g->nx * j];
        int64_t nb[2] = {kp1(g, i, j, k), km1(g, i, j, k)}; /* bottom :458, top :468 */
        int64_t kk[2] = {k + 1, k - 1};
        for (int d = 0; d < 2; ++d) {
            int64_t Cj = nb[d];
            if (Cj < 0) continue;
            int64_t wj = Lwet3D[Cj];
            if (wj == 0) continue;
            if (Omega && !Omega[wj - 1]) continue;
            double dd = fabs(zt[k] - zt[kk[d]]); /* :463, :473 */
            push_mix(&c, w + 1, wj, kappaV, a, dd, Vv);
        }
    }
    return c.len;
}

/* buildTκVML mask, matrixbuilding.jl:85: Ω[𝑖] = zt[k] < mlotst[i,j], missing (NaN here) => false */
void orc_ml_mask(const double *zt, const double *mlotst, const int64_t *Lwet, int64_t N,
                 const orc_grid *g, uint8_t *Omega) {
    const int64_t P = g->nx * g->ny;
    for (int64_t w = 0; w < N; ++w) {
        int64_t L = Lwet[w] - 1;
        Omega[w] = (zt[L / P] < mlotst[L % P]) ? 1 : 0;
    }
}

/* ---- SparseArrays.sparse(I,J,V,m,n) (stdlib; called at matrixbuilding.jl:41,63,92,116)
 * Restatement of sparse!: (1) counting sort of the triplets by row into an
 * unsorted-column CSR (stable: input order kept within a row); (2) one sweep
 * per row combining repeated (row,col) with + in input order (first touch
 * copies the value, later ones do acc = acc + v), counting columns;
 * (3) counting sort of the CSR into CSC => rows ascend within a column.
 * Stored zeros are kept.  colptr has n+1 entries, 1-based.  rowval/nzval need
 * capacity len.  Returns nnz or ORC_ERR_ALLOC. */
int64_t orc_sparse(const int64_t *I, const int64_t *J, const double *V, int64_t len, int64_t m,
                   int64_t n, int64_t *colptr, int64_t *rowval, double *nzval) {
    int64_t *csrrowptr = (int64_t *)calloc((size_t)m + 2, 8);
    int64_t *csrcolval = (int64_t *)malloc((size_t)(len > 0 ? len : 1) * 8);
    double *csrnzval = (double *)malloc((size_t)(len > 0 ? len : 1) * 8);
    int64_t *klasttouch = (int64_t *)calloc((size_t)n + 1, 8);
    if (!csrrowptr || !csrcolval || !csrnzval || !klasttouch) {
        free(csrrowptr); free(csrcolval); free(csrnzval); free(klasttouch);
        return ORC_ERR_ALLOC;
    }
    /* 1-based arrays emulated with 0-based storage: X[a] (Julia) == X[a-1] (C) */
    for (int64_t k = 0; k < len; ++k) csrrowptr[I[k]] += 1; /* csrrowptr[Ik+1] += 1 */
    {
        int64_t countsum = 1;
        csrrowptr[0] = 1;
        for (int64_t i = 2; i <= m + 1; ++i) {
            int64_t overwritten = csrrowptr[i - 1];
            csrrowptr[i - 1] = countsum;
            countsum += overwritten;
        }
    }
    for (int64_t k = 0; k < len; ++k) {
        int64_t Ik = I[k];
        int64_t csrk = csrrowptr[Ik]; /* csrrowptr[Ik+1] */
        csrrowptr[Ik] = csrk + 1;
        csrcolval[csrk - 1] = J[k];
        csrnzval[csrk - 1] = V[k];
    }
    for (int64_t j = 0; j <= n; ++j) colptr[j] = 0;
    int64_t writek = 1, newcsrrowptri = 1, origcsrrowptri = 1;
    int64_t origcsrrowptrip1 = (m >= 1) ? csrrowptr[1] : 1;
    for (int64_t i = 1; i <= m; ++i) {
        for (int64_t readk = origcsrrowptri; readk <= origcsrrowptrip1 - 1; ++readk) {
            int64_t j = csrcolval[readk - 1];
            if (klasttouch[j - 1] < newcsrrowptri) {
                klasttouch[j - 1] = writek;
                if (writek != readk) {
                    csrcolval[writek - 1] = j;
                    csrnzval[writek - 1] = csrnzval[readk - 1];
                }
                writek += 1;
                colptr[j] += 1; /* csccolptr[j+1] += 1 */
            } else {
                int64_t klt = klasttouch[j - 1];
                csrnzval[klt - 1] = csrnzval[klt - 1] + csrnzval[readk - 1]; /* combine = + */
            }
        }
        newcsrrowptri = writek;
        origcsrrowptri = origcsrrowptrip1;
        if (origcsrrowptrip1 != writek) csrrowptr[i] = writek; /* csrrowptr[i+1] */
        if (i < m) origcsrrowptrip1 = csrrowptr[i + 1];        /* csrrowptr[i+2] */
    }
    /* column pointers, shifted by one (reused as write cursors below) */
    {
        int64_t countsum = 1;
        colptr[0] = 1;
        for (int64_t j = 2; j <= n + 1; ++j) {
            int64_t overwritten = colptr[j - 1];
            colptr[j - 1] = countsum;
            countsum += overwritten;
        }
    }
    /* counting sort CSR -> CSC; rows visited ascending => sorted rows per column.
     * colptr slot j (= Julia csccolptr[j+1]) currently holds the start of column j and is
     * used as the write cursor; afterwards it holds the start of column j+1, i.e. the array
     * is the final 1-based colptr. */
    int64_t nnz = writek - 1;
    for (int64_t i = 1; i <= m; ++i) {
        for (int64_t csrk = csrrowptr[i - 1]; csrk <= csrrowptr[i] - 1; ++csrk) {
            int64_t j = csrcolval[csrk - 1];
            int64_t csck = colptr[j];
            colptr[j] = csck + 1;
            rowval[csck - 1] = i;
            nzval[csck - 1] = csrnzval[csrk - 1];
        }
    }
    free(csrrowptr); free(csrcolval); free(csrnzval); free(klasttouch);
    return nnz;
}

/* ---- SparseArrays.+(A,B) = map(+,A,B)  (stdlib; matrixbuilding.jl:147) ----
 * _map_zeropres!: per column, merge the two sorted row lists; Cx = Ax+Bx,
 * Ax+0.0 or 0.0+Bx; store only if Cx != 0.  C arrays need capacity
 * nnz(A)+nnz(B).  Returns nnz(C). */
int64_t orc_spadd(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax,
                  const int64_t *Bp, const int64_t *Bi, const double *Bx, int64_t *Cp,
                  int64_t *Ci, double *Cx) {
    int64_t Ck = 1;
    for (int64_t j = 0; j < n; ++j) {
        Cp[j] = Ck;
        int64_t Ak = Ap[j], stopA = Ap[j + 1];
        int64_t Bk = Bp[j], stopB = Bp[j + 1];
        while (Ak < stopA || Bk < stopB) {
            double x;
            int64_t r;
            if (Ak < stopA && Bk < stopB && Ai[Ak - 1] == Bi[Bk - 1]) {
                x = Ax[Ak - 1] + Bx[Bk - 1]; r = Ai[Ak - 1]; ++Ak; ++Bk;
            } else if (Bk >= stopB || (Ak < stopA && Ai[Ak - 1] < Bi[Bk - 1])) {
                x = Ax[Ak - 1] + 0.0; r = Ai[Ak - 1]; ++Ak;
            } else {
                x = 0.0 + Bx[Bk - 1]; r = Bi[Bk - 1]; ++Bk;
            }
            if (x != 0.0) { /* !_iszero(Cx); NaN != 0 is true and is kept */
                Ci[Ck - 1] = r;
                Cx[Ck - 1] = x;
                ++Ck;
            }
        }
    }
    Cp[n] = Ck;
    return Ck - 1;
}

/* ---- transportmatrix: matrixbuilding.jl:128-150 (+ buildT* :31-120) ------
 * Builds the five CSC matrices.  out arrays for matrix m (0=T 1=Tadv 2=TκH
 * 3=TκVML 4=TκVdeep): colptr[m] (N+1), rowval[m]/nzval[m] with capacities
 * cap[m] (safe: Tadv 12N, TκH 8N, TκV* 4N, T 28N; tight: 8N each).
 * stage_seconds (optional, 4 doubles) is filled by the caller's timer hooks. */
typedef struct {
    const double *phi[6];
    const double *v3D, *thk, *rho;
    double rho_scalar;
    const int64_t *Lwet, *Lwet3D;
    int64_t N;
    orc_grid g;
    const double *edge[4], *dist[4];
    const double *area2D, *zt, *mlotst;
    double kappaH, kappaVML, kappaVdeep;
    int32_t upwind;
} orc_tm_args;

static int has_nan(const double *v, int64_t n) {
    for (int64_t k = 0; k < n; ++k)
        if (isnan(v[k])) return 1;
    return 0;
}

int32_t orc_transportmatrix(const orc_tm_args *a, int64_t *const colptr[5],
                            int64_t *const rowval[5], double *const nzval[5], int64_t nnz[5]) {
    const int64_t N = a->N;
    const int64_t cap = 12 * (N > 0 ? N : 1);
    int64_t *I = (int64_t *)malloc((size_t)cap * 8), *J = (int64_t *)malloc((size_t)cap * 8);
    double *V = (double *)malloc((size_t)cap * 8);
    uint8_t *Om = (uint8_t *)malloc((size_t)(N > 0 ? N : 1));
    int32_t rc = ORC_OK;
    int64_t len;
    if (!I || !J || !V || !Om) { rc = ORC_ERR_ALLOC; goto done; }
    /* buildTadv :31-44 */
    len = orc_advection_entries(a->phi, a->v3D, a->rho, a->rho_scalar, a->Lwet, a->Lwet3D, N,
                                &a->g, a->upwind, I, J, V);
    if (len < 0) { rc = (int32_t)len; goto done; }
    if (has_nan(V, len)) { rc = ORC_ERR_TADV_NAN; goto done; }
    nnz[1] = orc_sparse(I, J, V, len, N, N, colptr[1], rowval[1], nzval[1]);
    /* buildTκH :51-66 */
    len = orc_hdiff_entries(a->v3D, a->thk, a->edge, a->dist, a->Lwet, a->Lwet3D, N, &a->g,
                            a->kappaH, NULL, I, J, V);
    if (len < 0) { rc = (int32_t)len; goto done; }
    if (has_nan(V, len)) { rc = ORC_ERR_TKH_NAN; goto done; }
    nnz[2] = orc_sparse(I, J, V, len, N, N, colptr[2], rowval[2], nzval[2]);
    /* buildTκVML :74-95 */
    orc_ml_mask(a->zt, a->mlotst, a->Lwet, N, &a->g, Om);
    len = orc_vdiff_entries(a->v3D, a->area2D, a->zt, a->Lwet, a->Lwet3D, N, &a->g,
                            a->kappaVML, Om, I, J, V);
    if (len < 0) { rc = (int32_t)len; goto done; }
    if (has_nan(V, len)) { rc = ORC_ERR_TKVML_NAN; goto done; }
    nnz[3] = orc_sparse(I, J, V, len, N, N, colptr[3], rowval[3], nzval[3]);
    /* buildTκVdeep :103-120 */
    len = orc_vdiff_entries(a->v3D, a->area2D, a->zt, a->Lwet, a->Lwet3D, N, &a->g,
                            a->kappaVdeep, NULL, I, J, V);
    if (len < 0) { rc = (int32_t)len; goto done; }
    if (has_nan(V, len)) { rc = ORC_ERR_TKVDEEP_NAN; goto done; }
    nnz[4] = orc_sparse(I, J, V, len, N, N, colptr[4], rowval[4], nzval[4]);
    if (nnz[1] < 0 || nnz[2] < 0 || nnz[3] < 0 || nnz[4] < 0) { rc = ORC_ERR_ALLOC; goto done; }
    /* T = Tadv + TκH + TκVML + TκVdeep :147 -- left fold of binary + */
    {
        int64_t c1 = nnz[1] + nnz[2] + 1, c2 = c1 + nnz[3];
        int64_t *p1 = (int64_t *)malloc((size_t)(N + 1) * 8), *i1 = (int64_t *)malloc((size_t)c1 * 8);
        double *x1 = (double *)malloc((size_t)c1 * 8);
        int64_t *p2 = (int64_t *)malloc((size_t)(N + 1) * 8), *i2 = (int64_t *)malloc((size_t)c2 * 8);
        double *x2 = (double *)malloc((size_t)c2 * 8);
        if (!p1 || !i1 || !x1 || !p2 || !i2 || !x2) {
            rc = ORC_ERR_ALLOC;
        } else {
            int64_t n1 = orc_spadd(N, colptr[1], rowval[1], nzval[1], colptr[2], rowval[2],
                                   nzval[2], p1, i1, x1);
            int64_t n2 = orc_spadd(N, p1, i1, x1, colptr[3], rowval[3], nzval[3], p2, i2, x2);
            (void)n1;
            (void)n2;
            nnz[0] = orc_spadd(N, p2, i2, x2, colptr[4], rowval[4], nzval[4], colptr[0],
                               rowval[0], nzval[0]);
        }
        free(p1); free(i1); free(x1); free(p2); free(i2); free(x2);
    }
done:
    free(I); free(J); free(V); free(Om);
    return rc;
}


/* ---- the same algorithm on several host threads (bench.py's second CPU figure) ---------------------------
 * The reference's formulation (scatter triplets, then sparse() = two counting sorts, then three adds) has little
 * parallelism to offer without changing it: what IS independent are the four operator builds (:140-143), and the
 * adds are independent per column.  So: four concurrent operator builds (triplets + NaN check + sparse()), then
 * the three adds column-parallel in two passes (count, prefix, write).  Output is bit-identical to
 * orc_transportmatrix (tests/test_oracle.py). */
#ifdef _OPENMP
#include <omp.h>

/* one column of map(+): writes into Ci/Cx when they are not NULL; returns the number of stored entries */
static inline int64_t spadd_col(int64_t j, const int64_t *Ap, const int64_t *Ai, const double *Ax, const int64_t *Bp,
                                const int64_t *Bi, const double *Bx, int64_t *Ci, double *Cx) {
    int64_t Ak = Ap[j], stopA = Ap[j + 1], Bk = Bp[j], stopB = Bp[j + 1], n = 0;
    while (Ak < stopA || Bk < stopB) {
        double x;
        int64_t r;
        if (Ak < stopA && Bk < stopB && Ai[Ak - 1] == Bi[Bk - 1]) {
            x = Ax[Ak - 1] + Bx[Bk - 1]; r = Ai[Ak - 1]; ++Ak; ++Bk;
        } else if (Bk >= stopB || (Ak < stopA && Ai[Ak - 1] < Bi[Bk - 1])) {
            x = Ax[Ak - 1] + 0.0; r = Ai[Ak - 1]; ++Ak;
        } else {
            x = 0.0 + Bx[Bk - 1]; r = Bi[Bk - 1]; ++Bk;
        }
        if (x != 0.0) {
            if (Ci) { Ci[n] = r; Cx[n] = x; }
            ++n;
        }
    }
    return n;
}

int64_t orc_spadd_omp(int64_t n, const int64_t *Ap, const int64_t *Ai, const double *Ax, const int64_t *Bp,
                      const int64_t *Bi, const double *Bx, int64_t *Cp, int64_t *Ci, double *Cx) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) Cp[j + 1] = spadd_col(j, Ap, Ai, Ax, Bp, Bi, Bx, NULL, NULL);
    Cp[0] = 1;
    for (int64_t j = 0; j < n; ++j) Cp[j + 1] += Cp[j]; /* serial prefix: n adds */
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) spadd_col(j, Ap, Ai, Ax, Bp, Bi, Bx, Ci + Cp[j] - 1, Cx + Cp[j] - 1);
    return Cp[n] - 1;
}

int32_t orc_omp_threads(void) { return omp_get_max_threads(); }

int32_t orc_transportmatrix_omp(const orc_tm_args *a, int64_t *const colptr[5], int64_t *const rowval[5],
                                double *const nzval[5], int64_t nnz[5]) {
    const int64_t N = a->N;
    int32_t rcs[4] = {ORC_OK, ORC_OK, ORC_OK, ORC_OK};
#pragma omp parallel for schedule(static, 1) num_threads(4)
    for (int m = 1; m <= 4; ++m) {
        /* triplet capacity per cell: 12 advective, 8 horizontal, 4 vertical (sizehints :235, :346, :447) */
        const int64_t cap = (m == 1 ? 12 : (m == 2 ? 8 : 4)) * (N > 0 ? N : 1);
        int64_t *I = (int64_t *)malloc((size_t)cap * 8), *J = (int64_t *)malloc((size_t)cap * 8);
        double *V = (double *)malloc((size_t)cap * 8);
        uint8_t *Om = (m == 3) ? (uint8_t *)malloc((size_t)(N > 0 ? N : 1)) : NULL;
        int64_t len = ORC_ERR_ALLOC;
        if (I && J && V && (m != 3 || Om)) {
            if (m == 1)
                len = orc_advection_entries(a->phi, a->v3D, a->rho, a->rho_scalar, a->Lwet, a->Lwet3D, N, &a->g,
                                            a->upwind, I, J, V);
            else if (m == 2)
                len = orc_hdiff_entries(a->v3D, a->thk, a->edge, a->dist, a->Lwet, a->Lwet3D, N, &a->g, a->kappaH,
                                        NULL, I, J, V);
            else if (m == 3) {
                orc_ml_mask(a->zt, a->mlotst, a->Lwet, N, &a->g, Om);
                len = orc_vdiff_entries(a->v3D, a->area2D, a->zt, a->Lwet, a->Lwet3D, N, &a->g, a->kappaVML, Om, I, J, V);
            } else
                len = orc_vdiff_entries(a->v3D, a->area2D, a->zt, a->Lwet, a->Lwet3D, N, &a->g, a->kappaVdeep, NULL, I,
                                        J, V);
        }
        if (len < 0)
            rcs[m - 1] = (int32_t)len;
        else if (has_nan(V, len))
            rcs[m - 1] = (m == 1) ? ORC_ERR_TADV_NAN : (m == 2) ? ORC_ERR_TKH_NAN : (m == 3) ? ORC_ERR_TKVML_NAN : ORC_ERR_TKVDEEP_NAN;
        else {
            nnz[m] = orc_sparse(I, J, V, len, N, N, colptr[m], rowval[m], nzval[m]);
            if (nnz[m] < 0) rcs[m - 1] = ORC_ERR_ALLOC;
        }
        free(I); free(J); free(V); free(Om);
    }
    for (int m = 0; m < 4; ++m) /* the serial order of the reference decides which error is reported */
        if (rcs[m] != ORC_OK) return rcs[m];
    int64_t c1 = nnz[1] + nnz[2] + 1, c2 = c1 + nnz[3];
    int64_t *p1 = (int64_t *)malloc((size_t)(N + 1) * 8), *i1 = (int64_t *)malloc((size_t)c1 * 8);
    double *x1 = (double *)malloc((size_t)c1 * 8);
    int64_t *p2 = (int64_t *)malloc((size_t)(N + 1) * 8), *i2 = (int64_t *)malloc((size_t)c2 * 8);
    double *x2 = (double *)malloc((size_t)c2 * 8);
    int32_t rc = ORC_OK;
    if (!p1 || !i1 || !x1 || !p2 || !i2 || !x2) {
        rc = ORC_ERR_ALLOC;
    } else {
        orc_spadd_omp(N, colptr[1], rowval[1], nzval[1], colptr[2], rowval[2], nzval[2], p1, i1, x1);
        orc_spadd_omp(N, p1, i1, x1, colptr[3], rowval[3], nzval[3], p2, i2, x2);
        nnz[0] = orc_spadd_omp(N, p2, i2, x2, colptr[4], rowval[4], nzval[4], colptr[0], rowval[0], nzval[0]);
    }
    free(p1); free(i1); free(x1); free(p2); free(i2); free(x2);
    return rc;
}
#endif /* _OPENMP */

/* ---- lump_and_spray: src/extratools.jl:38-119 ---------------------------------------------------------------
 * wet3D, mask (NULL = trues, :38): nx*ny*nz bytes; vol: N volumes of the wet cells; Tp/Ti: colptr/rowval of T
 * (1-based; only the pattern is used, findnz(T) :45).  Outputs: LUMP is Nc x N with exactly one entry per column
 * (lump_row[N] 1-based, lump_val[N]); SPRAY is its transposed pattern filled with ones (spray_colptr[Nc+1],
 * spray_row[N] 1-based); vol_c[Nc].  Capacities N (+1).  Third-party semantics restated: Graphs.SimpleGraph(adjmx)
 * requires a symmetric matrix (ArgumentError otherwise -> ORC_ERR_ASYMMETRIC); Graphs.connected_components lists
 * the components by ascending smallest vertex; SparseArrays products as cited inline. */
#define ORC_ERR_ASYMMETRIC (-20)
static int col_has_row(const int64_t *Tp, const int64_t *Ti, int64_t col, int64_t row) { /* T[row,col] stored? */
    for (int64_t q = Tp[col - 1]; q < Tp[col]; ++q)
        if (Ti[q - 1] == row) return 1;
    return 0;
}
int32_t orc_lump_and_spray(const uint8_t *wet3D, const uint8_t *mask, int64_t nx, int64_t ny, int64_t nz,
                           const double *vol, const int64_t *Tp, const int64_t *Ti, int64_t di, int64_t dj,
                           int64_t dk, int64_t *lump_row, double *lump_val, int64_t *spray_colptr,
                           int64_t *spray_row, double *vol_c, int64_t *Nc_out) {
    const int64_t ex = nx + di - 1, ey = ny + dj - 1, ez = nz + dk - 1; /* :41-43 */
    const int64_t G = nx * ny * nz, Gext = ex * ey * ez, bv = di * dj * dk;
    int64_t *LUMPidx = (int64_t *)calloc((size_t)Gext, 8);
    int64_t *rank = (int64_t *)calloc((size_t)G, 8); /* wet rank (1-based) of an original cell, 0 = dry */
    int64_t *cellof = (int64_t *)malloc((size_t)(G > 0 ? G : 1) * 8); /* original linear index (0-based) of wet rank r */
    int64_t *loc = (int64_t *)malloc((size_t)bv * 8), *lab = (int64_t *)malloc((size_t)bv * 8), *queue = (int64_t *)malloc((size_t)bv * 8);
    int32_t rc = ORC_OK;
    if (!LUMPidx || !rank || !cellof || !loc || !lab || !queue) { rc = ORC_ERR_ALLOC; goto done; }
    int64_t N = 0;
    for (int64_t L = 0; L < G; ++L)
        if (wet3D[L]) { cellof[N] = L; rank[L] = ++N; }
#define EXT(i, j, k) ((i) + ex * ((j) + ey * (k))) /* 0-based coordinates -> 0-based extended linear index, :47 */
    int64_t c = 2; /* :55 */
    for (int64_t k = 0; k < nz; ++k)
        for (int64_t j = 0; j < ny; ++j)
            for (int64_t i = 0; i < nx; ++i) { /* eachindex(C), :57 */
                const int64_t L = i + nx * (j + ny * k);
                const int inmask = mask ? (mask[L] != 0) : 1;
                if (LUMPidx[EXT(i, j, k)] > 0 && inmask) continue; /* :61 */
                if (!inmask) { /* :78-81 */
                    LUMPidx[EXT(i, j, k)] = c++;
                    continue;
                }
                /* the block C𝑖 .+ neighbours in vec order (i fastest), :64; wet members become the graph's vertices */
                int64_t nv = 0;
                for (int64_t d = 0; d < dk; ++d)
                    for (int64_t b = 0; b < dj; ++b)
                        for (int64_t a = 0; a < di; ++a) {
                            const int64_t ii = i + a, jj = j + b, kk = k + d;
                            const int inside = ii < nx && jj < ny && kk < nz;
                            if (inside && wet3D[ii + nx * (jj + ny * kk)])
                                loc[nv++] = ii + nx * (jj + ny * kk); /* original linear index of vertex nv */
                            else
                                LUMPidx[EXT(ii, jj, kk)] = 1; /* :66-68 (ghost cells of the extension are dry) */
                        }
                /* vertex of an original cell inside this block, or -1 */
#define VERTEX_OF(Lc, out)                                                                      \
    do {                                                                                        \
        const int64_t Lc_ = (Lc);                                                               \
        const int64_t ci = Lc_ % nx, cj = (Lc_ / nx) % ny, ck = Lc_ / (nx * ny);                \
        (out) = -1;                                                                             \
        if (ci >= i && ci < i + di && cj >= j && cj < j + dj && ck >= k && ck < k + dk)         \
            for (int64_t q_ = 0; q_ < nv; ++q_)                                                 \
                if (loc[q_] == Lc_) { (out) = q_; break; }                                      \
    } while (0)
                /* SimpleGraph(view(connectivitymatrix, wetidx, wetidx)): symmetric or ArgumentError, :71-72.  Every
                 * asymmetric pair has one direction stored, so walking the stored entries finds it. */
                for (int64_t v = 0; v < nv && rc == ORC_OK; ++v) {
                    const int64_t col = rank[loc[v]];
                    for (int64_t q = Tp[col - 1]; q < Tp[col]; ++q) {
                        const int64_t row = Ti[q - 1];
                        int64_t u;
                        VERTEX_OF(cellof[row - 1], u);
                        if (u >= 0 && !col_has_row(Tp, Ti, row, col)) { rc = ORC_ERR_ASYMMETRIC; break; }
                    }
                }
                if (rc != ORC_OK) goto done;
                /* connected_components: vertices ascending; a search from an unlabelled vertex labels its component */
                for (int64_t v = 0; v < nv; ++v) lab[v] = -1;
                for (int64_t u0 = 0; u0 < nv; ++u0) {
                    if (lab[u0] >= 0) continue;
                    int64_t head = 0, tail = 0;
                    lab[u0] = u0;
                    queue[tail++] = u0;
                    while (head < tail) {
                        const int64_t src = queue[head++], col = rank[loc[src]];
                        for (int64_t q = Tp[col - 1]; q < Tp[col]; ++q) {
                            int64_t u;
                            VERTEX_OF(cellof[Ti[q - 1] - 1], u);
                            if (u >= 0 && lab[u] < 0) { lab[u] = u0; queue[tail++] = u; }
                        }
                    }
                    for (int64_t v = 0; v < nv; ++v) /* LUMPidx[wetidx[comp]] .= c, :75 */
                        if (lab[v] == u0) {
                            const int64_t Lc = loc[v];
                            LUMPidx[EXT(Lc % nx, (Lc / nx) % ny, Lc / (nx * ny))] = c;
                        }
                    ++c; /* :76 */
                }
#undef VERTEX_OF
            }
    /* LUMP = sparse(LUMPidx[C][:], 1:length(C), 1) :85; wet_c = LUMP * wet .> 0 :88; LUMP = LUMP[wet_c, wet] :91 */
    {
        int64_t *newrow = (int64_t *)calloc((size_t)c + 1, 8);
        if (!newrow) { rc = ORC_ERR_ALLOC; goto done; }
        for (int64_t w = 0; w < N; ++w) {
            const int64_t Lc = cellof[w];
            newrow[LUMPidx[EXT(Lc % nx, (Lc / nx) % ny, Lc / (nx * ny))]] = 1;
        }
        int64_t Nc = 0;
        for (int64_t r = 1; r <= c; ++r)
            if (newrow[r]) newrow[r] = ++Nc;
        for (int64_t w = 0; w < N; ++w) {
            const int64_t Lc = cellof[w];
            lump_row[w] = newrow[LUMPidx[EXT(Lc % nx, (Lc / nx) % ny, Lc / (nx * ny))]];
        }
        free(newrow);
        *Nc_out = Nc;
        /* vol_c = LUMP * vol :96 -- mul!: for every column j ascending, y[row] += 1 * vol[j] */
        for (int64_t I = 0; I < Nc; ++I) vol_c[I] = 0.0;
        for (int64_t w = 0; w < N; ++w) vol_c[lump_row[w] - 1] = vol_c[lump_row[w] - 1] + 1 * vol[w];
        /* LUMP = sparse(Diagonal(1 ./ vol_c)) * LUMP * sparse(Diagonal(vol)) :97, evaluated left to right */
        for (int64_t w = 0; w < N; ++w) lump_val[w] = ((1.0 / vol_c[lump_row[w] - 1]) * 1) * vol[w];
        /* SPRAY = copy(LUMP'); SPRAY.nzval .= 1 :101-102 -- transposition = stable counting sort by row */
        for (int64_t I = 0; I <= Nc; ++I) spray_colptr[I] = 0;
        for (int64_t w = 0; w < N; ++w) spray_colptr[lump_row[w]] += 1;
        spray_colptr[0] = 1;
        for (int64_t I = 1; I <= Nc; ++I) spray_colptr[I] += spray_colptr[I - 1];
        int64_t *cur = (int64_t *)malloc((size_t)(Nc > 0 ? Nc : 1) * 8);
        if (!cur) { rc = ORC_ERR_ALLOC; goto done; }
        for (int64_t I = 0; I < Nc; ++I) cur[I] = spray_colptr[I];
        for (int64_t w = 0; w < N; ++w) spray_row[cur[lump_row[w] - 1]++ - 1] = w + 1;
        free(cur);
    }
#undef EXT
done:
    free(LUMPidx); free(rank); free(cellof); free(loc); free(lab); free(queue);
    return rc;
}

/* ---- velocity2fluxes / fluxes2velocity: velocities.jl:10-39, :50-74, nanmean2 :89-93, nanmin2 :108 -----
 * Default C-grid only (interpolateontodefaultCgrid passes C-grid fields through, gridcellgeometry.jl:104).
 * rho: 3-D array or NULL with rho_scalar (twocellnanmean(x::Number) = x, :81).  Loops over ALL cells.
 * A bipolar top row has j₊₁ == nothing, and thkcello[nothing] throws in the reference -> error. */
static inline double nanmean2(double a, double b) { /* Bool weights: false * NaN == 0.0 in Julia */
    int wa = !isnan(a), wb = !isnan(b);
    return ((wa ? a : 0.0) + (wb ? b : 0.0)) / (double)(wa + wb);
}
static inline double nanmin2(double a, double b) { return isnan(a) ? b : (isnan(b) ? a : (a < b ? a : b)); }

int32_t orc_velocity_flux(const double *in_i, const double *in_j, const double *rho, double rho_scalar,
                          const double *thk, const double *edge_east, const double *edge_north,
                          const orc_grid *g, int32_t to_velocity, double *out_i, double *out_j) {
    if (g->topo == ORC_UNKNOWN) return ORC_ERR_UNKNOWN_TOPOLOGY;
    if (g->topo == ORC_BIPOLAR) return ORC_ERR_FLUX_INTO_LAND; /* x[nothing] at j == ny */
    for (int64_t k = 0; k < g->nz; ++k)
        for (int64_t j = 0; j < g->ny; ++j)
            for (int64_t i = 0; i < g->nx; ++i) {
                int64_t c = lin(g, i, j, k), s = i + g->nx * j;
                int64_t E = ip1(g, i, j, k), N = jp1(g, i, j, k);
                double mE = rho ? nanmean2(rho[c], rho[E]) : rho_scalar;
                double mN = rho ? nanmean2(rho[c], rho[N]) : rho_scalar;
                double tE = nanmin2(thk[c], thk[E]), tN = nanmin2(thk[c], thk[N]);
                if (!to_velocity) {
                    out_i[c] = in_i[c] * mE * tE * edge_east[s];  /* :31 */
                    out_j[c] = in_j[c] * mN * tN * edge_north[s]; /* :33 */
                } else {
                    out_i[c] = in_i[c] / (mE * tE * edge_east[s]);  /* :68 */
                    out_j[c] = in_j[c] / (mN * tN * edge_north[s]); /* :70 */
                }
            }
    return ORC_OK;
}


/* ---- bolus_GM_velocity: RediGM.jl:46-79 (experimental in the reference; never enters T; PARITY UNPINNED:
 * no reference test asserts anything about it, test/derivatives.jl only plots).
 * globalverticalfacetriadderivative triads.jl:134-146 (group :84-112, derivative :114-133),
 * globalverticaldyadderivative dyads.jl:66-78 (group :38-56, derivative :57-65).
 * dist_e / dist_n are gridmetrics.distance_to_neighbour_2D[:east] / [:north]: horizontaldistance(lon,lat,C,E)
 * (gridcellgeometry.jl:182-188) is the same haversine between the same two centroids.
 * wet3D: cells of indices.Lwet; every other cell of u, v stays NaN (fill(NaN, size(χ))). */
static inline double gnan(const double *x, int64_t L) { return (L < 0) ? NAN : x[L]; } /* getindexornan */
static inline double nansum_over_count4(const double v[4]) { /* sum(w*v)/sum(w), false*NaN == 0.0 */
    double s = 0.0; int n = 0;
    for (int q = 0; q < 4; ++q) { int w = !isnan(v[q]); s = (q == 0) ? (w ? v[0] : 0.0) : s + (w ? v[q] : 0.0); n += w; }
    return s / (double)n;
}
static double triad_slope(const double *chi, const double *Z, const double *dist2d, const orc_grid *g,
                          int64_t i, int64_t j, int64_t k, int dirJ) {
    int64_t I = lin(g, i, j, k), N = km1(g, i, j, k), S = kp1(g, i, j, k);
    int64_t E = dirJ ? jp1(g, i, j, k) : ip1(g, i, j, k);
    int64_t NE = -1, SE = -1;
    if (E >= 0) {
        int64_t ie = E % g->nx, je = (E / g->nx) % g->ny;
        NE = km1(g, ie, je, k); SE = kp1(g, ie, je, k);
    }
    double vC = chi[I], vN = gnan(chi, N), vS = gnan(chi, S), vE = gnan(chi, E), vNE = gnan(chi, NE), vSE = gnan(chi, SE);
    double dCN = fabs(gnan(Z, N) - Z[I]), dCS = fabs(gnan(Z, S) - Z[I]);         /* verticaldistance(Z,I,J) = |Z[J]-Z[I]| */
    double dCE = (E < 0) ? NAN : dist2d[i + g->nx * j];
    double dENE = fabs(gnan(Z, NE) - gnan(Z, E)), dESE = fabs(gnan(Z, SE) - gnan(Z, E));
    double CN = (vN - vC) / dCN, CS = (vC - vS) / dCS, CE = (vE - vC) / dCE, ENE = (vNE - vE) / dENE, ESE = (vE - vSE) / dESE;
    double r[4] = {CE / CN, CE / CS, CE / ENE, CE / ESE};
    return nansum_over_count4(r);
}
static double dyad_deriv(const double *chi, const double *Z, const orc_grid *g, int64_t i, int64_t j, int64_t k) {
    int64_t I = lin(g, i, j, k), N = km1(g, i, j, k), S = kp1(g, i, j, k);
    double dCN = fabs(gnan(Z, N) - Z[I]), dCS = fabs(gnan(Z, S) - Z[I]);
    double a = (gnan(chi, N) - chi[I]) / dCN, b = (chi[I] - gnan(chi, S)) / dCS;
    int wa = !isnan(a), wb = !isnan(b);
    return ((wa ? a : 0.0) + (wb ? b : 0.0)) / (double)(wa + wb);
}
static inline double jl_clamp(double x, double lo, double hi) { return (x > hi) ? hi : ((x < lo) ? lo : x); }

int32_t orc_bolus_gm_velocity(const double *rho, const double *Z3D, const uint8_t *wet3D, const double *dist_e,
                              const double *dist_n, const orc_grid *g, double kappaGM, double maxslope,
                              double *u, double *v) {
    if (g->topo == ORC_UNKNOWN) return ORC_ERR_UNKNOWN_TOPOLOGY;
    if (g->topo == ORC_BIPOLAR) return ORC_ERR_FLUX_INTO_LAND; /* k₋₁(nothing) throws at j == ny (triads.jl:87-88) */
    const int64_t G = g->nx * g->ny * g->nz;
    double *Si = (double *)malloc((size_t)G * 8), *Sj = (double *)malloc((size_t)G * 8);
    if (!Si || !Sj) { free(Si); free(Sj); return ORC_ERR_ALLOC; }
    for (int64_t k = 0; k < g->nz; ++k)
        for (int64_t j = 0; j < g->ny; ++j)
            for (int64_t i = 0; i < g->nx; ++i) {
                int64_t I = lin(g, i, j, k);
                double si = NAN, sj = NAN;
                if (wet3D[I]) {
                    si = triad_slope(rho, Z3D, dist_e, g, i, j, k, 0); /* RediGM.jl:52 */
                    sj = triad_slope(rho, Z3D, dist_n, g, i, j, k, 1); /* :53 */
                }
                si = jl_clamp(si, -maxslope, maxslope);                /* :56-57 */
                sj = jl_clamp(sj, -maxslope, maxslope);
                double taper = 0.5 * (1 + tanh((0.004 - sqrt(si * si + sj * sj)) / 0.001)); /* :59-62 */
                Si[I] = kappaGM * (taper * si);                        /* :63-64, :76-77 κGM .* Sᵢ */
                Sj[I] = kappaGM * (taper * sj);
            }
    for (int64_t k = 0; k < g->nz; ++k)
        for (int64_t j = 0; j < g->ny; ++j)
            for (int64_t i = 0; i < g->nx; ++i) {
                int64_t I = lin(g, i, j, k);
                u[I] = wet3D[I] ? dyad_deriv(Si, Z3D, g, i, j, k) : NAN; /* :76 */
                v[I] = wet3D[I] ? dyad_deriv(Sj, Z3D, g, i, j, k) : NAN; /* :77 */
            }
    free(Si); free(Sj);
    return ORC_OK;
}

/* ---- Distances.haversine 0.10 (radius 6371000), points are (lon°, lat°) --- */
double orc_haversine(double lon1, double lat1, double lon2, double lat2) {
    const double d2r = M_PI / 180.0; /* deg2rad(z) = z * (pi/180) */
    double dl = (lon2 - lon1) * d2r;          /* Δλ = deg2rad(y[1] - x[1]) */
    double p1 = lat1 * d2r, p2 = lat2 * d2r;  /* φ₁ = deg2rad(x[2]), φ₂ = deg2rad(y[2]) */
    double dp = p2 - p1;                      /* Δφ = φ₂ - φ₁: the latitudes are converted FIRST and subtracted after (Distances.jl 0.10
                                               * haversine.jl); (lat2 - lat1) * d2r, as rounds 1-4 had it, differs in the last ulp */
    double s1 = sin(dp / 2), s2 = sin(dl / 2);
    double a = s1 * s1 + cos(p1) * cos(p2) * (s2 * s2);
    double r = sqrt(a);
    if (r > 1.0) r = 1.0; /* min(√a, 1) */
    return 2 * (6371000.0 * asin(r));
}

/* ==== makegridmetrics and the velocity-grid helpers: src/gridcellgeometry.jl (rows a14, f2, f4 of SURVEY.md §8) ====
 * Scalar restatement, one cell at a time like the reference's comprehensions.  Distances.haversine is a third-party
 * dependency (Distances.jl 0.10, not on disk): orc_haversine above states its published formula with sin/cos of
 * deg2rad(x); newer Distances releases evaluate the same formula with sind/cosd, whose results can differ in the last
 * ulps -- the north star's tolerance for floating point (1e-12 relative) covers that, and the tests that compare
 * transcendental results use it.  Everything else here (replace, divisions, cumsum, midpoints, index pairing) is
 * exact arithmetic and compared bit for bit.                                                                    */

/* midpointonsphere, :249-255 */
void orc_midpointonsphere(double lonA, double latA, double lonB, double latB, double *lon, double *lat) {
    if (fabs(lonA - lonB) < 180) {
        *lon = (lonA + lonB) / 2;
        *lat = (latA + latB) / 2;
    } else { /* the edge crosses the longitudinal edge of the map */
        *lon = (lonA + lonB) / 2 + 180;
        *lat = (latA + latB) / 2 + 0;
    }
}

#define VTX(a, v, i, j) ((a)[(v) + 4 * ((i) + nx * (j))]) /* (4,nx,ny) column-major */

/* vertexpermutation, :158-178: 0-based permutation that sorts the vertices of cell (1,1) into SW, SE, NE, NW using the
 * vertices it shares with its east and north neighbours.  -1: some `only(...)` would throw.                       */
int32_t orc_vertexpermutation(const double *lonv, const double *latv, int64_t nx, int64_t ny, int32_t perm[4]) {
    if (nx < 2 || ny < 2) return -1;
    int in_e[4], in_n[4];
    for (int v = 0; v < 4; ++v) {
        in_e[v] = in_n[v] = 0;
        for (int w = 0; w < 4; ++w) { /* Set(points) ∩ Set(points_east): tuple equality (==; -0.0 == 0.0, NaN never) */
            if (VTX(lonv, v, 0, 0) == VTX(lonv, w, 1, 0) && VTX(latv, v, 0, 0) == VTX(latv, w, 1, 0)) in_e[v] = 1;
            if (VTX(lonv, v, 0, 0) == VTX(lonv, w, 0, 1) && VTX(latv, v, 0, 0) == VTX(latv, w, 0, 1)) in_n[v] = 1;
        }
    }
    int idx3 = -1, idx2 = -1, idx4 = -1, idx1 = -1, c;
    c = 0; for (int v = 0; v < 4; ++v) if (in_e[v] && in_n[v]) { idx3 = v; ++c; }          /* common to all 3 cells */
    if (c != 1) return -1;
    c = 0; for (int v = 0; v < 4; ++v) if (in_e[v] && v != idx3) { idx2 = v; ++c; }        /* (i,j) and (i+1,j) only */
    if (c != 1) return -1;
    c = 0; for (int v = 0; v < 4; ++v) if (in_n[v] && v != idx3) { idx4 = v; ++c; }        /* (i,j) and (i,j+1) only */
    if (c != 1) return -1;
    c = 0; for (int v = 0; v < 4; ++v) if (v != idx2 && v != idx3 && v != idx4) { idx1 = v; ++c; }
    if (c != 1) return -1;
    perm[0] = idx1; perm[1] = idx2; perm[2] = idx3; perm[3] = idx4;
    return 0;
}

/* getgridtopology, gridtopology.jl:33-53 with isapprox_lon :23-26.  Vertices in the default order.  isapprox on arrays
 * compares 2-norms: norm(x - y) <= max(atol, rtol * max(norm(x), norm(y))), rtol = sqrt(eps) unless atol > 0.    */
int32_t orc_getgridtopology(const double *lonv, const double *latv, int64_t nx, int64_t ny) {
    int all90 = 1;
    for (int64_t i = 0; i < nx; ++i)
        for (int v = 2; v < 4; ++v)
            if (!(VTX(latv, v, i, ny - 1) == 90)) all90 = 0;
    if (all90) return 0; /* BipolarGridTopology */
    /* rot180 of the (2,nx) view: element (v,i) <-> (1-v, nx-1-i), i.e. vertex 3 of cell i <-> vertex 4 of cell nx+1-i */
    double nlon = 0, ndlat = 0, nlat = 0;
    for (int64_t i = 0; i < nx; ++i)
        for (int v = 0; v < 2; ++v) {
            const double a = VTX(lonv, 2 + v, i, ny - 1), b = VTX(lonv, 2 + (1 - v), nx - 1 - i, ny - 1);
            const double x = a - b + 180;
            double md = fmod(x, 360.0); /* Julia mod: result has the sign of the divisor */
            if (md != 0 && md < 0) md += 360.0;
            const double d = md - 180;
            nlon += d * d;
            const double p = VTX(latv, 2 + v, i, ny - 1), q = VTX(latv, 2 + (1 - v), nx - 1 - i, ny - 1);
            ndlat += (p - q) * (p - q);
            nlat += p * p; /* norm(NPlat) == norm(rot180(NPlat)) */
        }
    const double eps180 = 2.8421709430404007e-14; /* eps(180.0) */
    const int lon_ok = sqrt(nlon) <= eps180;
    const int lat_ok = sqrt(ndlat) <= 1.4901161193847656e-08 * sqrt(nlat); /* rtol = sqrt(eps(Float64)) */
    return (lon_ok && lat_ok) ? 1 : 2; /* Tripolar : Unknown */
}

/* replace(x, toreplace...) of makegridmetrics :269-280: missing/nothing (NaN here), 0 and the two _FillValues become NaN.
 * replace() matches with isequal: -0.0 is NOT isequal to 0, so a negative zero stays what it is.                 */
static inline double mgm_replace(double x, double fill_a, int has_a, double fill_v, int has_v) {
    if (x == 0 && !signbit(x)) return NAN;
    if (has_a && isequal_f64(x, fill_a)) return NAN;
    if (has_v && isequal_f64(x, fill_v)) return NAN;
    return x;
}

/* makegridmetrics, :265-311.  lonv_in/latv_in as given (any vertex order); lonv/latv receive the sorted vertices (:297-298).
 * The per-direction outputs are in this file's W, E, S, N order.  Returns the topology (0, 1) or -7 for an unknown one
 * (horizontaldistance would call j₊₁ on UnknownGridTopology -> error, gridtopology.jl:111-116), -1 when
 * vertexpermutation throws.                                                                                      */
int32_t orc_makegridmetrics(const double *volcello, const double *areacello, double fill_area, int32_t has_fill_area,
                            double fill_vol, int32_t has_fill_vol, const double *lon, const double *lat,
                            const double *lonv_in, const double *latv_in, int64_t nx, int64_t ny, int64_t nz, double *area2D,
                            double *v3D, double *thk, double *Z3D, double *lonv, double *latv, double *const edge[4],
                            double *const dist_edge[4], double *const dist_nbr[4]) {
    const int64_t P = nx * ny;
    for (int64_t s = 0; s < P; ++s) area2D[s] = mgm_replace(areacello[s], fill_area, has_fill_area, fill_vol, has_fill_vol);
    for (int64_t s = 0; s < P; ++s) {
        double zbot = 0; /* cumsum(thkcello, dims = 3): sequential sum from the surface, NaN sticks */
        for (int64_t k = 0; k < nz; ++k) {
            const int64_t L = s + P * k;
            v3D[L] = mgm_replace(volcello[L], fill_area, has_fill_area, fill_vol, has_fill_vol);
            thk[L] = v3D[L] / area2D[s];                     /* :283 */
            zbot = (k == 0) ? thk[L] : zbot + thk[L];        /* :284 */
            Z3D[L] = zbot - 0.5 * thk[L];                    /* :285 */
        }
    }
    int32_t perm[4];
    if (orc_vertexpermutation(lonv_in, latv_in, nx, ny, perm)) return -1; /* :296 */
    for (int64_t s = 0; s < P; ++s)
        for (int v = 0; v < 4; ++v) {
            lonv[v + 4 * s] = lonv_in[perm[v] + 4 * s];
            latv[v + 4 * s] = latv_in[perm[v] + 4 * s];
        }
    const int32_t topo = orc_getgridtopology(lonv, latv, nx, ny); /* :302 */
    if (topo == 2) return -7;
    orc_grid g = {nx, ny, 1, topo};
    /* vertexindices :209-215 (0-based) for W, E, S, N */
    static const int va[4] = {0, 1, 0, 2}, vb[4] = {3, 2, 1, 3};
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            const int64_t s = i + nx * j;
            for (int d = 0; d < 4; ++d) {
                const double aL = VTX(lonv, va[d], i, j), aT = VTX(latv, va[d], i, j);
                const double bL = VTX(lonv, vb[d], i, j), bT = VTX(latv, vb[d], i, j);
                edge[d][s] = orc_haversine(aL, aT, bL, bT); /* verticalfacewidth :217-222 */
                double mL, mT;
                orc_midpointonsphere(aL, aT, bL, bT, &mL, &mT);
                dist_edge[d][s] = orc_haversine(lon[s], lat[s], mL, mT); /* centroid2edgedistance :240-247 */
                /* horizontaldistance(lon, lat, 𝑖, 𝑗(𝑖, gridtopology)) with west<->i₋₁, east<->i₊₁, south<->j₋₁, north<->j₊₁ (:304-308) */
                const int64_t nb = (d == 0) ? im1(&g, i, j, 0) : (d == 1) ? ip1(&g, i, j, 0) : (d == 2) ? jm1(&g, i, j, 0) : jp1(&g, i, j, 0);
                dist_nbr[d][s] = (nb < 0) ? NAN : orc_haversine(lon[s], lat[s], lon[nb], lat[nb]); /* :182-189 */
            }
        }
    return topo;
}

/* getarakawagrid, :50-95 on cell (1,1).  *kind: 0 A, 1 B, 2 C; u_pos/v_pos: index into (C, SW, SE, NE, NW, S, N, W, E), the
 * field order of the reference's `cell` NamedTuple (findmin returns the FIRST minimum in that order).  -1: "Unknown
 * Arakawa grid type".  *relerr: (u_distance + v_distance) / perimeter (:89-90).                                   */
int32_t orc_getarakawagrid(double u_lon, double u_lat, double v_lon, double v_lat, const double *lon, const double *lat,
                           const double *lonv, const double *latv, int32_t *kind, int32_t *u_pos, int32_t *v_pos, double *relerr) {
    double PL[9], PT[9];
    PL[0] = lon[0]; PT[0] = lat[0];
    for (int v = 0; v < 4; ++v) { PL[1 + v] = lonv[v]; PT[1 + v] = latv[v]; } /* SW, SE, NE, NW */
    orc_midpointonsphere(PL[1], PT[1], PL[2], PT[2], &PL[5], &PT[5]); /* S = mid(SW, SE) */
    orc_midpointonsphere(PL[3], PT[3], PL[4], PT[4], &PL[6], &PT[6]); /* N = mid(NE, NW) */
    orc_midpointonsphere(PL[1], PT[1], PL[4], PT[4], &PL[7], &PT[7]); /* W = mid(SW, NW) */
    orc_midpointonsphere(PL[2], PT[2], PL[3], PT[3], &PL[8], &PT[8]); /* E = mid(SE, NE) */
    double ud = 0, vd = 0;
    int up = -1, vp = -1;
    for (int q = 0; q < 9; ++q) {
        const double du = orc_haversine(PL[q], PT[q], u_lon, u_lat), dv = orc_haversine(PL[q], PT[q], v_lon, v_lat);
        if (up < 0 || du < ud) { ud = du; up = q; }
        if (vp < 0 || dv < vd) { vd = dv; vp = q; }
    }
    *u_pos = up; *v_pos = vp;
    const int corner_u = up >= 1 && up <= 4;
    if (up == 0 && vp == 0) *kind = 0;
    else if (up == vp && corner_u) *kind = 1;
    else if ((up == 8 || up == 7) && (vp == 6 || vp == 5)) *kind = 2;
    else return -1;
    const double per = orc_haversine(PL[1], PT[1], PL[2], PT[2]) + orc_haversine(PL[2], PT[2], PL[3], PT[3]) +
                       orc_haversine(PL[3], PT[3], PL[4], PT[4]) + orc_haversine(PL[4], PT[4], PL[1], PT[1]);
    *relerr = (ud + vd) / per;
    return 0;
}

/* interpolateontodefaultCgrid(…, ::BGridCell), :106-140 (NE-corner B-grid): _FillValue -> 0 (replace: isequal), then
 * u2 = 0.5 (u2 + [zeros ;; u2[:, 1:end-1, :]]) (one row south), v2 = 0.5 (v2 + [zeros; v2[1:end-1, :, :]]) (one cell west);
 * the new velocity points are midpointonsphere(NE, SE) and midpointonsphere(NW, NE) -- the reference zips
 * (NE_points, SE_points) and (NW_points, NE_points) in that order (:131-132).                                     */
void orc_bgrid_to_cgrid(const double *u, const double *v, double fill, int64_t nx, int64_t ny, int64_t nz, const double *lonv,
                        const double *latv, double *u2, double *v2, double *u2_lon, double *u2_lat, double *v2_lon,
                        double *v2_lat) {
    const int64_t P = nx * ny;
    for (int64_t k = 0; k < nz; ++k)
        for (int64_t j = 0; j < ny; ++j)
            for (int64_t i = 0; i < nx; ++i) {
                const int64_t L = i + nx * j + P * k;
                const double uc = isequal_f64(u[L], fill) ? 0.0 : u[L], vc = isequal_f64(v[L], fill) ? 0.0 : v[L];
                const double us = (j == 0) ? 0.0 : (isequal_f64(u[L - nx], fill) ? 0.0 : u[L - nx]);
                const double vw = (i == 0) ? 0.0 : (isequal_f64(v[L - 1], fill) ? 0.0 : v[L - 1]);
                u2[L] = 0.5 * (uc + us);
                v2[L] = 0.5 * (vc + vw);
            }
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            const int64_t s = i + nx * j;
            orc_midpointonsphere(VTX(lonv, 2, i, j), VTX(latv, 2, i, j), VTX(lonv, 1, i, j), VTX(latv, 1, i, j), &u2_lon[s], &u2_lat[s]);
            orc_midpointonsphere(VTX(lonv, 3, i, j), VTX(latv, 3, i, j), VTX(lonv, 2, i, j), VTX(latv, 2, i, j), &v2_lon[s], &v2_lat[s]);
        }
}
