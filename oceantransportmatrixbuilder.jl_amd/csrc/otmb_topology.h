// otmb_topology.h -- device index shifts of the C-grid topology.
// Mirrors src/gridtopology.jl:57-68 (i±1 periodic, j±1 and k±1 closed) and :94 (tripolar fold:
// j₊₁ of (i,ny) is (nx-i+1,ny)).  0-based (i,j,k); a shift returns the 0-based linear index of
// the neighbour, or -1 for Julia's `nothing`.  UnknownGridTopology is rejected on the host.
#pragma once
#include "otmb_common.h"

struct Cell {
    int i, j, k;
    i64 L;     // linear index
    i64 row0;  // linear index of (0,j,k)
};

__device__ __forceinline__ Cell cell_of(i64 L, int nx, int ny, i64 P) {
    Cell c;
    c.L = L;
    // 32-bit division: the host rejects grids with 2^32 or more cells
    unsigned k = (unsigned)L / (unsigned)P;
    unsigned r = (unsigned)L - k * (unsigned)P;
    unsigned j = r / (unsigned)nx;
    c.k = (int)k;
    c.j = (int)j;
    c.i = (int)(r - j * (unsigned)nx);
    c.row0 = L - c.i;
    return c;
}
__device__ __forceinline__ i64 nb_ip1(const Cell &c, int nx) { return c.row0 + ((c.i + 1 < nx) ? c.i + 1 : 0); }
__device__ __forceinline__ i64 nb_im1(const Cell &c, int nx) { return c.row0 + ((c.i > 0) ? c.i - 1 : nx - 1); }
__device__ __forceinline__ i64 nb_jm1(const Cell &c, int nx) { return (c.j > 0) ? c.L - nx : -1; }
__device__ __forceinline__ i64 nb_jp1(const Cell &c, int nx, int ny, int topo) {
    if (c.j + 1 < ny) return c.L + nx;
    return (topo == OTMB_TRIPOLAR) ? c.row0 + (nx - 1 - c.i) : -1;
}
__device__ __forceinline__ i64 nb_kp1(const Cell &c, int nz, i64 P) { return (c.k + 1 < nz) ? c.L + P : -1; }
__device__ __forceinline__ i64 nb_km1(const Cell &c, i64 P) { return (c.k > 0) ? c.L - P : -1; }
