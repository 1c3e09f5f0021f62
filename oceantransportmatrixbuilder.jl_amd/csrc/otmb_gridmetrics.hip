// otmb_gridmetrics.hip -- the array work of makegridmetrics on the device (src/gridcellgeometry.jl:265-311):
//   * v3D / area2D: {0, _FillValue(area), _FillValue(vol)} -> NaN (:269-280); thkcello = v3D ./ area2D (:283);
//     Z3D = cumsum(thkcello, dims=3) - thkcello/2 (:284-285), a sequential sum down each water column;
//   * the twelve (nx,ny) metric arrays: edge_length_2D (haversine between the two vertices of an edge, :306,
//     :217-222), distance_to_edge_2D (centroid to edge midpoint, :307, :240-255) and distance_to_neighbour_2D
//     (centroid to the topological neighbour's centroid, NaN where it is `nothing`, :308, :182-189).
// Vertex permutation (:158-178) and topology detection (gridtopology.jl:33-53) look at a handful of values and
// stay on the host; the permutation is applied here while reading the vertices.
// Distances.haversine 0.10: 2r·asin(min(√(sin²(Δφ/2)+cosφ₁cosφ₂sin²(Δλ/2)),1)), r = 6371000, degrees in.
// sin/cos/asin come from the device math library (<= 1-2 ulp from the host's): values agree with the host
// implementation to ~1e-15 relative, inside the 1e-12 the path allows, but are not bit-identical.
#include "otmb_common.h"

__device__ __forceinline__ double gm_haversine(double lon1, double lat1, double lon2, double lat2) {
    const double d2r = 3.14159265358979323846 / 180.0;
    const double dl = (lon2 - lon1) * d2r;          // Δλ = deg2rad(y[1] - x[1])
    const double p1 = lat1 * d2r, p2 = lat2 * d2r;  // φ₁ = deg2rad(x[2]), φ₂ = deg2rad(y[2])
    const double dp = p2 - p1;                      // Δφ = φ₂ - φ₁: converted first, subtracted after (Distances.jl 0.10, haversine.jl)
    const double s1 = sin(dp / 2), s2 = sin(dl / 2);
    const double a = s1 * s1 + cos(p1) * cos(p2) * (s2 * s2);
    const double r = sqrt(a);
    return 2 * (6371000.0 * asin(r < 1.0 ? r : 1.0));
}

// direction order of the outputs: 0 = west, 1 = east, 2 = south, 3 = north (OTMB_DIR_*)
__global__ __launch_bounds__(256) void gridmetrics2d_kernel(const double *__restrict__ lon, const double *__restrict__ lat,
                                                            const double *__restrict__ lonv, const double *__restrict__ latv,
                                                            int p0, int p1, int p2, int p3, int nx, int ny, int topo,
                                                            double *e_w, double *e_e, double *e_s, double *e_n, double *c_w,
                                                            double *c_e, double *c_s, double *c_n, double *d_w, double *d_e,
                                                            double *d_s, double *d_n) {
    const i64 s = (i64)blockIdx.x * 256 + threadIdx.x;
    if (s >= (i64)nx * ny) return;
    const int j = (int)(s / nx), i = (int)(s - (i64)j * nx);
    const int perm[4] = {p0, p1, p2, p3};
    double vl[4], vt[4];  // vertices in the default order SW, SE, NE, NW (:150-155)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int src = (q == 0) ? perm[0] : (q == 1) ? perm[1] : (q == 2) ? perm[2] : perm[3];
        vl[q] = lonv[4 * s + src];
        vt[q] = latv[4 * s + src];
    }
    const double cl = lon[s], ct = lat[s];
    // vertexindices (:209-215): south (1,2), east (2,3), north (3,4), west (1,4)
#define EDGE(A, B, OUT_E, OUT_C)                                                                   \
    {                                                                                              \
        OUT_E[s] = gm_haversine(vl[A], vt[A], vl[B], vt[B]);                                       \
        const bool cross = !(fabs(vl[A] - vl[B]) < 180.0); /* midpointonsphere :249-255 */         \
        const double ml = (vl[A] + vl[B]) / 2 + (cross ? 180.0 : 0.0), mt = (vt[A] + vt[B]) / 2 + 0.0; \
        OUT_C[s] = gm_haversine(cl, ct, ml, mt);                                                   \
    }
    EDGE(0, 1, e_s, c_s)
    EDGE(1, 2, e_e, c_e)
    EDGE(2, 3, e_n, c_n)
    EDGE(0, 3, e_w, c_w)
#undef EDGE
    // neighbours: west i-1, east i+1 (periodic), south j-1, north j+1 (closed / tripolar fold)
    const i64 sw = (i64)j * nx + ((i > 0) ? i - 1 : nx - 1), se = (i64)j * nx + ((i + 1 < nx) ? i + 1 : 0);
    d_w[s] = gm_haversine(cl, ct, lon[sw], lat[sw]);
    d_e[s] = gm_haversine(cl, ct, lon[se], lat[se]);
    d_s[s] = (j > 0) ? gm_haversine(cl, ct, lon[s - nx], lat[s - nx]) : __builtin_nan("");
    i64 sn = (j + 1 < ny) ? s + nx : ((topo == OTMB_TRIPOLAR) ? (i64)j * nx + (nx - 1 - i) : -1);
    d_n[s] = (sn >= 0) ? gm_haversine(cl, ct, lon[sn], lat[sn]) : __builtin_nan("");
}

__global__ __launch_bounds__(64) void gridmetrics3d_kernel(const double *__restrict__ vol, const double *__restrict__ area_in,
                                                           double f0, double f1, int nx, int ny, int nz, double *__restrict__ area2d,
                                                           double *__restrict__ v3d, double *__restrict__ thk, double *__restrict__ z3d) {
    const i64 P = (i64)nx * ny;
    const i64 s = (i64)blockIdx.x * 64 + threadIdx.x;
    if (s >= P) return;
    double a = area_in[s];
    // :269-280 (missing already arrives as NaN); replace() matches with isequal: -0.0 is not isequal to 0
    if ((a == 0.0 && !signbit(a)) || a == f0 || a == f1) a = __builtin_nan("");
    area2d[s] = a;
    double zbot = 0.0;
    for (int k = 0; k < nz; ++k) {
        double v = vol[(i64)k * P + s];
        if ((v == 0.0 && !signbit(v)) || v == f0 || v == f1) v = __builtin_nan("");
        const double t = v / a;                       // :283
        zbot = (k == 0) ? t : zbot + t;               // cumsum(thkcello, dims = 3), :284
        v3d[(i64)k * P + s] = v;
        thk[(i64)k * P + s] = t;
        z3d[(i64)k * P + s] = zbot - 0.5 * t;         // :285
    }
}

extern "C" int32_t otmb_makegridmetrics_dev(otmb_ctx *ctx, const double *volcello, const double *areacello, double fill_area,
                                            double fill_vol, const double *lon, const double *lat, const double *lon_vertices,
                                            const double *lat_vertices, const int32_t perm[4], int64_t nx, int64_t ny, int64_t nz,
                                            int32_t topology, double *area2d, double *v3d, double *thkcello, double *z3d,
                                            double *const edge_length[4], double *const dist_edge[4], double *const dist_nbr[4]) {
    if (!ctx || !volcello || !areacello || !lon || !lat || !lon_vertices || !lat_vertices || !perm || !area2d || !v3d || !thkcello ||
        !z3d || !edge_length || !dist_edge || !dist_nbr)
        return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "null argument");
    if (nx < 1 || ny < 1 || nz < 1) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "grid size");
    if (topology == OTMB_UNKNOWN_TOPOLOGY) return otmb_fail(ctx, OTMB_ERR_UNKNOWN_TOPOLOGY);  // :308 calls j₊₁ -> gridtopology.jl:111
    for (int d = 0; d < 4; ++d)
        if (!edge_length[d] || !dist_edge[d] || !dist_nbr[d] || perm[d] < 0 || perm[d] > 3) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "outputs / perm");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const i64 P = nx * ny;
    KernelTimer kt(ctx, K_GRIDMETRICS);
    hipLaunchKernelGGL(gridmetrics2d_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, ctx->stream, lon, lat, lon_vertices,
                       lat_vertices, (int)perm[0], (int)perm[1], (int)perm[2], (int)perm[3], (int)nx, (int)ny, (int)topology,
                       edge_length[0], edge_length[1], edge_length[2], edge_length[3], dist_edge[0], dist_edge[1], dist_edge[2],
                       dist_edge[3], dist_nbr[0], dist_nbr[1], dist_nbr[2], dist_nbr[3]);
    hipLaunchKernelGGL(gridmetrics3d_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, ctx->stream, volcello, areacello, fill_area,
                       fill_vol, (int)nx, (int)ny, (int)nz, area2d, v3d, thkcello, z3d);
    HIP_TRY(ctx, hipGetLastError());
    return OTMB_OK;
}
