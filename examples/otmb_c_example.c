/* A plain-C caller of libotmb_hip.so (include/otmb.h): the calls a Julia `ccall` shim makes, without any shim.
 *
 *   makeindices(v3D) -> facefluxes(umo, vmo) -> transportmatrix(ϕ, …)      (src/matrixbuilding.jl:10-24, src/velocities.jl:190-255,
 *                                                                           src/matrixbuilding.jl:128-150)
 * on a small analytic grid: nx x ny x nz boxes of 1 degree x 1 degree x 10 m, a land column at i = 2, tripolar seam, mass transports
 * that are smooth functions of (i, j, k).  Prints N, the five nnz and the sum of T's values, then builds a second time slice with TκH and
 * TκVdeep passed back (otmb_tm_args.given); tests/test_c_example.py compiles and runs it on the GPU box and compares with the Python host
 * layer on the same inputs.
 *
 *   gcc -O2 -I include examples/otmb_c_example.c -L oceantransportmatrixbuilder.jl_amd/lib -lotmb_hip -Wl,-rpath,$PWD/oceantransportmatrixbuilder.jl_amd/lib -lm -o otmb_c_example
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "otmb.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int32_t rc_ = (call);                                                                        \
        if (rc_ != OTMB_OK) {                                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? otmb_last_error(ctx) : otmb_status_string(rc_)); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char **argv) {
    const int64_t nx = argc > 1 ? atoll(argv[1]) : 12, ny = argc > 2 ? atoll(argv[2]) : 8, nz = argc > 3 ? atoll(argv[3]) : 5;
    const int64_t P = nx * ny, G = P * nz;
    otmb_ctx *ctx = NULL;
    CHECK(otmb_ctx_create(0, &ctx));

    /* gridmetrics, as makegridmetrics would hand them over (column-major (nx,ny,nz), NaN on land) */
    double *v3d = malloc(G * 8), *thk = malloc(G * 8), *rho = malloc(G * 8), *umo = malloc(G * 8), *vmo = malloc(G * 8);
    double *edge[4], *dist[4], *area = malloc(P * 8), *ml = malloc(P * 8), *zt = malloc(nz * 8);
    for (int d = 0; d < 4; ++d) { edge[d] = malloc(P * 8); dist[d] = malloc(P * 8); }
    for (int64_t k = 0; k < nz; ++k) zt[k] = 5.0 + 10.0 * k;
    for (int64_t j = 0; j < ny; ++j)
        for (int64_t i = 0; i < nx; ++i) {
            const int64_t s = j * nx + i;
            const double dx = 111e3 * cos((j - ny / 2.0) * 0.01), dy = 111e3;
            area[s] = dx * dy;
            ml[s] = 12.0 + 3.0 * ((i + j) % 4);
            edge[OTMB_DIR_WEST][s] = edge[OTMB_DIR_EAST][s] = dy;
            edge[OTMB_DIR_SOUTH][s] = edge[OTMB_DIR_NORTH][s] = dx;
            dist[OTMB_DIR_WEST][s] = dist[OTMB_DIR_EAST][s] = dx;
            dist[OTMB_DIR_SOUTH][s] = (j > 0) ? dy : NAN;  /* no neighbour: NaN, as in the reference */
            dist[OTMB_DIR_NORTH][s] = dy;                   /* tripolar: the top row's north neighbour is the folded cell */
            for (int64_t k = 0; k < nz; ++k) {
                const int64_t L = k * P + s;
                const int land = (i == 2) || (k == nz - 1 && (i + j) % 3 == 0);
                thk[L] = 10.0;
                v3d[L] = land ? NAN : area[s] * 10.0;
                rho[L] = 1025.0 + 0.01 * k + 0.001 * i;
                umo[L] = 1e6 * sin(0.3 * i + 0.2 * j + 0.1 * k);
                vmo[L] = 1e6 * cos(0.2 * i - 0.3 * j + 0.2 * k);
            }
        }

    /* makeindices */
    int64_t *lwet3d = malloc(G * 8), *lwet = malloc(G * 8), N = 0;
    uint8_t *wet = malloc(G);
    CHECK(otmb_makeindices(ctx, v3d, nx, ny, nz, lwet3d, lwet, wet, &N));

    /* facefluxesfrommasstransport */
    double *phi[6];
    for (int f = 0; f < 6; ++f) phi[f] = malloc(G * 8);
    CHECK(otmb_facefluxes(ctx, umo, vmo, 0, wet, 1e20, nx, ny, nz, OTMB_TRIPOLAR, phi));

    /* transportmatrix: plan (sizes) -> the caller allocates -> fetch */
    otmb_tm_args a = {0};
    a.nx = nx; a.ny = ny; a.nz = nz; a.topology = OTMB_TRIPOLAR; a.upwind = 1; a.n_wet = N;
    for (int f = 0; f < 6; ++f) a.phi[f] = phi[f];
    a.v3d = v3d; a.thkcello = thk; a.rho = rho; a.lwet3d = lwet3d; a.lwet = lwet;
    for (int d = 0; d < 4; ++d) { a.edge_length[d] = edge[d]; a.dist_nbr[d] = dist[d]; }
    a.area2d = area; a.zt = zt; a.mlotst = ml;
    a.kappa_h = 500.0; a.kappa_vml = 0.1; a.kappa_vdeep = 1e-5;
    int64_t nnz[5], final[5];
    CHECK(otmb_transportmatrix_plan(ctx, &a, nnz));
    int64_t *colptr[5], *rowval[5];
    double *nzval[5];
    for (int m = 0; m < 5; ++m) {
        colptr[m] = malloc((N + 1) * 8);
        rowval[m] = malloc((nnz[m] ? nnz[m] : 1) * 8);
        nzval[m] = malloc((nnz[m] ? nnz[m] : 1) * 8);
    }
    CHECK(otmb_transportmatrix_fetch(ctx, colptr, rowval, nzval, final));

    double sumT = 0.0, sumabs = 0.0;
    for (int64_t e = 0; e < final[OTMB_T]; ++e) { sumT += nzval[OTMB_T][e]; sumabs += fabs(nzval[OTMB_T][e]); }
    printf("N=%lld nnz=%lld,%lld,%lld,%lld,%lld colptrT_last=%lld sumT=%.17g sumabsT=%.17g\n", (long long)N, (long long)final[0],
           (long long)final[1], (long long)final[2], (long long)final[3], (long long)final[4], (long long)colptr[OTMB_T][N], sumT, sumabs);

    /* The next time slice: TκH and TκVdeep depend on the grid and κ alone, so the caller passes them back (transportmatrix's TκH = / TκVdeep =
     * keywords, src/matrixbuilding.jl:133-147; otmb_tm_args.given).  They are then neither built, counted nor copied home -- their nnz come
     * back 0, their output pointers may be NULL -- and T is formed with the matrices passed: the same T, bit for bit. */
    const int keep[2] = {OTMB_TKH, OTMB_TKVDEEP};
    for (int q = 0; q < 2; ++q) {
        const int m = keep[q];
        a.given[m].colptr = colptr[m]; a.given[m].rowval = rowval[m]; a.given[m].nzval = nzval[m]; a.given[m].nnz = final[m];
    }
    int64_t nnz2[5], final2[5];
    CHECK(otmb_transportmatrix_plan(ctx, &a, nnz2));
    int64_t *colptr2[5] = {0}, *rowval2[5] = {0};
    double *nzval2[5] = {0};
    for (int m = 0; m < 5; ++m) {
        if (m == OTMB_TKH || m == OTMB_TKVDEEP) continue;
        colptr2[m] = malloc((N + 1) * 8);
        rowval2[m] = malloc((nnz2[m] ? nnz2[m] : 1) * 8);
        nzval2[m] = malloc((nnz2[m] ? nnz2[m] : 1) * 8);
    }
    CHECK(otmb_transportmatrix_fetch(ctx, colptr2, rowval2, nzval2, final2));
    int same = final2[OTMB_T] == final[OTMB_T];
    for (int64_t e = 0; same && e < final[OTMB_T]; ++e) same = rowval2[OTMB_T][e] == rowval[OTMB_T][e] && nzval2[OTMB_T][e] == nzval[OTMB_T][e];
    printf("given_state=%d,%d nnz2=%lld,%lld,%lld,%lld,%lld T_same=%d\n", (int)otmb_ctx_given_state(ctx, OTMB_TKH), (int)otmb_ctx_given_state(ctx, OTMB_TKVDEEP),
           (long long)final2[0], (long long)final2[1], (long long)final2[2], (long long)final2[3], (long long)final2[4], same);
    otmb_ctx_destroy(ctx);
    return 0;
}
