// otmb_pack.hip -- the grid's inputs laid out for the assembly kernel (include/otmb.h: cell_records, metric_records).
//
// tm_kernel<fill> is bound by the NUMBER of vector-memory instructions a wave issues (measured: removing ten of its 57
// eight-byte loads takes 10 % off, whichever ten), not by the bytes they move.  A 16-byte load costs the CU's memory
// pipeline what an 8-byte load costs, so values that are always wanted together at the same cell go side by side:
//   cell record   (32 B per cell)    { v3D, Lwet3D | ρ, thkcello }           7 + 7 + 7 + 5 loads  ->  14
//   metric record (80 B per column)  { edge, dist } x W, E, S, N | { area2D, mlotst }   18 loads  ->   9
// Same bytes, half the instructions.  Pure data movement (no arithmetic on the values: parity is untouched).
#include "otmb_common.h"

typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pack_cells_kernel(const double *__restrict__ v, const i64 *__restrict__ lw,
                                                         const double *__restrict__ rho, double rho_s,
                                                         const double *__restrict__ thk, i64 G, d2 *__restrict__ rec) {
    const i64 L = (i64)blockIdx.x * 256 + threadIdx.x;
    if (L >= G) return;
    d2 a, b;
    a.x = v[L];
    a.y = __longlong_as_double(lw[L]);
    b.x = rho ? rho[L] : rho_s;
    b.y = thk[L];
    rec[2 * L] = a;
    rec[2 * L + 1] = b;
}

__global__ __launch_bounds__(256) void pack_metrics_kernel(const double *__restrict__ eW, const double *__restrict__ eE,
                                                           const double *__restrict__ eS, const double *__restrict__ eN,
                                                           const double *__restrict__ dW, const double *__restrict__ dE,
                                                           const double *__restrict__ dS, const double *__restrict__ dN,
                                                           const double *__restrict__ area, const double *__restrict__ ml, i64 P,
                                                           d2 *__restrict__ rec) {
    const i64 s = (i64)blockIdx.x * 256 + threadIdx.x;
    if (s >= P) return;
    d2 q;
    q.x = eW[s]; q.y = dW[s]; rec[5 * s + 0] = q;
    q.x = eE[s]; q.y = dE[s]; rec[5 * s + 1] = q;
    q.x = eS[s]; q.y = dS[s]; rec[5 * s + 2] = q;
    q.x = eN[s]; q.y = dN[s]; rec[5 * s + 3] = q;
    q.x = area[s]; q.y = ml[s]; rec[5 * s + 4] = q;
}

int32_t otmb_launch_pack_cells(otmb_ctx *ctx, const otmb_tm_args *a, void *rec) {
    const i64 G = a->nx * a->ny * a->nz;
    if (G <= 0) return OTMB_OK;
    KernelTimer kt(ctx, K_PACK);
    hipLaunchKernelGGL(pack_cells_kernel, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, ctx->stream, a->v3d, (const i64 *)a->lwet3d,
                       a->rho, a->rho_scalar, a->thkcello, G, (d2 *)rec);
    return OTMB_OK;
}

int32_t otmb_launch_pack_metrics(otmb_ctx *ctx, const otmb_tm_args *a, void *rec) {
    const i64 P = a->nx * a->ny;
    if (P <= 0) return OTMB_OK;
    KernelTimer kt(ctx, K_PACK);
    hipLaunchKernelGGL(pack_metrics_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, ctx->stream, a->edge_length[OTMB_DIR_WEST],
                       a->edge_length[OTMB_DIR_EAST], a->edge_length[OTMB_DIR_SOUTH], a->edge_length[OTMB_DIR_NORTH],
                       a->dist_nbr[OTMB_DIR_WEST], a->dist_nbr[OTMB_DIR_EAST], a->dist_nbr[OTMB_DIR_SOUTH], a->dist_nbr[OTMB_DIR_NORTH],
                       a->area2d, a->mlotst, P, (d2 *)rec);
    return OTMB_OK;
}

static int32_t check_pack_args(otmb_ctx *ctx, const otmb_tm_args *a, void *rec) {
    if (!ctx || !a || !rec) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (a->nx < 1 || a->ny < 1 || a->nz < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    return OTMB_OK;
}

extern "C" int32_t otmb_pack_cells_dev(otmb_ctx *ctx, const otmb_tm_args *a, void *rec) {
    int32_t rc;
    if ((rc = check_pack_args(ctx, a, rec))) return rc;
    if (!a->v3d || !a->thkcello || !a->lwet3d) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null input array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((rc = otmb_launch_pack_cells(ctx, a, rec))) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}

extern "C" int32_t otmb_pack_metrics_dev(otmb_ctx *ctx, const otmb_tm_args *a, void *rec) {
    int32_t rc;
    if ((rc = check_pack_args(ctx, a, rec))) return rc;
    for (int d = 0; d < 4; ++d)
        if (!a->edge_length[d] || !a->dist_nbr[d]) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "metrics");
    if (!a->area2d || !a->mlotst) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null input array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((rc = otmb_launch_pack_metrics(ctx, a, rec))) return rc;
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}
