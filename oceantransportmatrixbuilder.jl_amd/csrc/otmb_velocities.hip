// otmb_velocities.hip -- velocity2fluxes / fluxes2velocity on the device (src/velocities.jl:10-39, :50-74,
// nanmean2 :89-93, nanmin2 :108): mass flux through the east/north face = velocity x mean density of the
// two cells sharing the face x their minimum thickness x the edge length.  Default C-grid (the reference's
// interpolateontodefaultCgrid passes C-grid fields through unchanged, src/gridcellgeometry.jl:104).
// One thread per grid cell (the reference loops over every cell, wet or not); pure streaming, HBM-bound:
// reads u, v, thk (+rho) and the two (nx,ny) edge arrays, writes two arrays.
#include "otmb_common.h"
#include "otmb_topology.h"

__device__ __forceinline__ double vf_nanmean2(double a, double b) {  // Bool weights: false * NaN == 0.0 in Julia
    const bool wa = !isnan(a), wb = !isnan(b);
    return ((wa ? a : 0.0) + (wb ? b : 0.0)) / (double)((int)wa + (int)wb);
}
__device__ __forceinline__ double vf_nanmin2(double a, double b) { return isnan(a) ? b : (isnan(b) ? a : ((a < b) ? a : b)); }

template <typename T, bool TO_VELOCITY>
__global__ __launch_bounds__(256) void velocity_flux_kernel(const T *__restrict__ in_i, const T *__restrict__ in_j,
                                                             const double *__restrict__ rho, double rho_s,
                                                             const double *__restrict__ thk, const double *__restrict__ edge_e,
                                                             const double *__restrict__ edge_n, int nx, int ny, int nz, i64 P,
                                                             i64 G, double *__restrict__ out_i, double *__restrict__ out_j) {
    const i64 L = (i64)blockIdx.x * 256 + threadIdx.x;
    if (L >= G) return;
    const Cell c = cell_of(L, nx, ny, P);
    const i64 E = nb_ip1(c, nx), N = nb_jp1(c, nx, ny, OTMB_TRIPOLAR);  // the host rejects topologies with j₊₁ == nothing
    const i64 s2 = (i64)c.j * nx + c.i;
    const double tc = thk[L], tE = vf_nanmin2(tc, thk[E]), tN = vf_nanmin2(tc, thk[N]);
    double mE = rho_s, mN = rho_s;
    if (rho) {
        const double rc = rho[L];
        mE = vf_nanmean2(rc, rho[E]);
        mN = vf_nanmean2(rc, rho[N]);
    }
    const double a = (double)in_i[L], b = (double)in_j[L];
    if (TO_VELOCITY) {
        out_i[L] = a / (mE * tE * edge_e[s2]);  // :68
        out_j[L] = b / (mN * tN * edge_n[s2]);  // :70
    } else {
        out_i[L] = a * mE * tE * edge_e[s2];  // :31
        out_j[L] = b * mN * tN * edge_n[s2];  // :33
    }
}

// B-grid (velocities on the NE corner) -> default C-grid: interpolateontodefaultCgrid(…, ::BGridCell),
// src/gridcellgeometry.jl:106-140.  u2 = 0.5 (u + u shifted by one row in j), v2 = 0.5 (v + v shifted by one cell
// in i), with _FillValue replaced by 0 first (:125-128); the shifted-in row/column is zero (no periodic wrap).
template <typename T>
__global__ __launch_bounds__(256) void bgrid_to_cgrid_kernel(const T *__restrict__ u, const T *__restrict__ v, double fill, int nx,
                                                              int ny, i64 P, i64 G, double *__restrict__ u2, double *__restrict__ v2) {
    const i64 L = (i64)blockIdx.x * 256 + threadIdx.x;
    if (L >= G) return;
    const i64 r = L % P;
    const int j = (int)(r / nx), i = (int)(r - (i64)j * nx);
    auto rep = [fill](double x) { return (__double_as_longlong(x) == __double_as_longlong(fill)) ? 0.0 : x; };  // replace(u, fill => 0.0)
    const double uc = rep((double)u[L]), vc = rep((double)v[L]);
    const double us = (j > 0) ? rep((double)u[L - nx]) : 0.0;  // [zeros(nx,1,nz);; u2[:, 1:end-1, :]]  (:127)
    const double vw = (i > 0) ? rep((double)v[L - 1]) : 0.0;   // [zeros(1,ny,nz); v2[1:end-1, :, :]]   (:128)
    u2[L] = 0.5 * (uc + us);
    v2[L] = 0.5 * (vc + vw);
}

extern "C" int32_t otmb_bgrid_to_cgrid_dev(otmb_ctx *ctx, const void *u, const void *v, int32_t src_is_f32, double fill, int64_t nx,
                                           int64_t ny, int64_t nz, double *u2, double *v2) {
    if (!ctx || !u || !v || !u2 || !v2) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1 || nx * ny * nz >= (1ll << 32)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 P = nx * ny, G = P * nz;
    const unsigned nb = (unsigned)((G + 255) / 256);
    KernelTimer kt(ctx, K_VELFLUX);
    if (src_is_f32)
        hipLaunchKernelGGL(bgrid_to_cgrid_kernel<float>, dim3(nb), dim3(256), 0, ctx->stream, (const float *)u, (const float *)v, fill,
                           (int)nx, (int)ny, P, G, u2, v2);
    else
        hipLaunchKernelGGL(bgrid_to_cgrid_kernel<double>, dim3(nb), dim3(256), 0, ctx->stream, (const double *)u, (const double *)v,
                           fill, (int)nx, (int)ny, P, G, u2, v2);
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}

static int32_t vf_launch(otmb_ctx *ctx, const void *in_i, const void *in_j, int32_t src_is_f32, const double *rho,
                         double rho_scalar, const double *thk, const double *edge_e, const double *edge_n, int64_t nx,
                         int64_t ny, int64_t nz, int32_t topology, bool to_velocity, double *out_i, double *out_j) {
    if (!ctx || !in_i || !in_j || !thk || !edge_e || !edge_n || !out_i || !out_j) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1 || nx * ny * nz >= (1ll << 32)) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    if (topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);
    // bipolar: j₊₁ of the top row is `nothing` and thkcello[nothing] throws in the reference (velocities.jl:33)
    if (topology != OTMB_TRIPOLAR) return otmb_fail(ctx, OTMB_ERR_FLUX_INTO_LAND, "velocity2fluxes indexes j₊₁ == nothing on a bipolar grid");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 P = nx * ny, G = P * nz;
    const unsigned nb = (unsigned)((G + 255) / 256);
    KernelTimer kt(ctx, K_VELFLUX);
#define VF(T, TV)                                                                                                      \
    hipLaunchKernelGGL((velocity_flux_kernel<T, TV>), dim3(nb), dim3(256), 0, ctx->stream, (const T *)in_i, (const T *)in_j, \
                       rho, rho_scalar, thk, edge_e, edge_n, (int)nx, (int)ny, (int)nz, P, G, out_i, out_j)
    if (src_is_f32) { if (to_velocity) VF(float, true); else VF(float, false); }
    else { if (to_velocity) VF(double, true); else VF(double, false); }
#undef VF
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}

extern "C" {
int32_t otmb_velocity2fluxes_dev(otmb_ctx *ctx, const void *u, const void *v, int32_t src_is_f32, const double *rho,
                                 double rho_scalar, const double *thkcello, const double *edge_east, const double *edge_north,
                                 int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *phi_i, double *phi_j) {
    return vf_launch(ctx, u, v, src_is_f32, rho, rho_scalar, thkcello, edge_east, edge_north, nx, ny, nz, topology, false, phi_i, phi_j);
}
int32_t otmb_fluxes2velocity_dev(otmb_ctx *ctx, const void *phi_i, const void *phi_j, int32_t src_is_f32, const double *rho,
                                 double rho_scalar, const double *thkcello, const double *edge_east, const double *edge_north,
                                 int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *u, double *v) {
    return vf_launch(ctx, phi_i, phi_j, src_is_f32, rho, rho_scalar, thkcello, edge_east, edge_north, nx, ny, nz, topology, true, u, v);
}
}
