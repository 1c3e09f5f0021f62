#!/usr/bin/env python3
"""EXPERIMENT (measured, not kept: profiles/r03/README.md 5b): the fill pass without its workgroup barrier.  The counting pass also stores each WAVE's
packed sums (wavesums), every wave of the fill pass fetches the tile's offsets and the four wave sums itself and places itself from them.
Bit-identical (198 GPU tests), +3 % at 1 degree and 0 at 0.25 degree on the final kernels.  This script applies the change to a checkout:
    python tools/experiments/nobarrier_patch.py && python tools/dense_ab.py --build nobar="" && git checkout oceantransportmatrixbuilder.jl_amd/csrc"""
import os
root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oceantransportmatrixbuilder.jl_amd', 'csrc') + os.sep
p=root+'otmb_transportmatrix.hip'
s=open(p).read()
def rep(old,new):
    global s
    assert s.count(old)==1, old[:60]
    s=s.replace(old,new,1)
def cut(a_marker,b_marker,new):
    global s
    a=s.index(a_marker); b=s.index(b_marker,a); assert a<b
    s=s[:a]+new+s[b:]
rep("""        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0) wave_tot[q][wid] = x;
    }
    __syncthreads();
    if (tid < TPB * TM_NF) {""","""        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
        if (lane == 0) {
            wave_tot[q][wid] = x;
            const i64 tile = (i64)blockIdx.x * TPB + q;
            if (tile < ntiles) p.wavesums[tile * (TM_THREADS / 64) + wid] = x;
        }
    }
    __syncthreads();
    if (tid < TPB * TM_NF) {""")
cut("    unsigned pre_sum = 0;\n    i64 pre_off = 0;\n    if (MODE == MODE_FILL && tid < TM_NF) {","    const i64 base_elem = (Lmin > p.P) ? Lmin - p.P : 0;\n    const bool span_ok","""    i64 pre_off = 0;
    u64 ws_l = 0;
    if (MODE == MODE_FILL) {
        if (lane < TM_THREADS / 64) ws_l = p.wavesums[tile * (TM_THREADS / 64) + lane];
        if (lane < TM_NF) {
            pre_off = p.tileoffs[tile * TM_NF + lane];
            if (p.gsum) {
                const i64 g = tile / OTMB_SCAN_GROUP;
                for (i64 q0 = 0; q0 < g; q0 += 8) {
                    i64 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = p.gsum[((q0 + u < g) ? q0 + u : 0) * TM_NF + lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) pre_off += (q0 + u < g) ? t[u] : 0;
                }
            }
        }
    }
""")
cut("    if (lane == 63) wave_tot[wid] = incl;\n    if (MODE == MODE_FILL && tid < TM_NF) {","    const u64 excl = before + incl - mine;","""    u64 before = 0, all = 0;
    i64 g0[5] = {0, 0, 0, 0, 0};
    if (MODE == MODE_FILL) {
        const int widu = __builtin_amdgcn_readfirstlane(wid);
        u64 reserved = 0;
#pragma unroll
        for (int q = 0; q < TM_THREADS / 64; ++q) {
            const u64 v = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ws_l >> 32), q) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ws_l, q);
            if (q < widu) before += v;
            if (q == widu) reserved = v;
            all += v;
        }
        const u64 wsum = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(incl >> 32), 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)incl, 63);
        if (wsum != reserved) {
            if (lane == 0) raise_flag(p.flags, FLAG_COUNT_MISMATCH);
            return;
        }
#pragma unroll
        for (int m = 0; m < TM_NF; ++m)
            g0[m] = (i64)(((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)((u64)pre_off >> 32), m) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u64)pre_off, m));
    } else {
        if (lane == 63) wave_tot[wid] = incl;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TM_THREADS / 64; ++q) {
            const u64 v = wave_tot[q];
            if (q < wid) before += v;
            all += v;
        }
    }
""")
cut("    if (MODE == MODE_ONEPASS) __syncthreads();  // s_prefix comes from the look-back of wave 0","    if (w0 + TM_THREADS >= p.n_own) {  // last tile","""    if (MODE == MODE_ONEPASS) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < TM_NF; ++m) g0[m] = s_prefix[m];
    }

""")
assert s.count("otmb_reserve(ctx, ctx->tm_sums, (size_t)(ntiles + 1) * TM_NF * sizeof(uint32_t))")==2
s=s.replace("otmb_reserve(ctx, ctx->tm_sums, (size_t)(ntiles + 1) * TM_NF * sizeof(uint32_t))","otmb_reserve(ctx, ctx->tm_sums, tm_sums_bytes(ntiles))")
rep("    p.tilesums = (uint32_t *)ctx->tm_sums.p;\n    p.tileoffs = (const i64 *)ctx->tm_offs.p;\n    p.flags = (int *)ctx->flags.p;\n}\n","    p.tilesums = (uint32_t *)ctx->tm_sums.p;\n    p.wavesums = (u64 *)((char *)ctx->tm_sums.p + tm_tilesums_bytes((a.n_wet + TM_THREADS - 1) / TM_THREADS));\n    p.tileoffs = (const i64 *)ctx->tm_offs.p;\n    p.flags = (int *)ctx->flags.p;\n}\n")
rep("template <int MODE>\n__global__ __launch_bounds__(TM_THREADS, TM_WAVES_PER_SIMD) void tm_kernel(","static inline size_t tm_tilesums_bytes(i64 ntiles) { return (((size_t)(ntiles + 1) * TM_NF * sizeof(uint32_t)) + 15) & ~(size_t)15; }\nstatic inline size_t tm_sums_bytes(i64 ntiles) { return tm_tilesums_bytes(ntiles) + (size_t)(ntiles + 1) * (TM_THREADS / 64) * sizeof(u64); }\n\ntemplate <int MODE>\n__global__ __launch_bounds__(TM_THREADS, TM_WAVES_PER_SIMD) void tm_kernel(")
open(p,'w').write(s)
p=root+'otmb_tm_column.h'
s=open(p).read()
rep("    const i64 *tileoffs;   // [ntiles][5]  (FILL reads)\n","    u64 *wavesums;\n    const i64 *tileoffs;   // [ntiles][5]  (FILL reads)\n")
open(p,'w').write(s)
