// otmb_common.h -- shared declarations of libotmb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/otmb.h"

typedef int64_t i64;
typedef uint64_t u64;

// One growable device buffer owned by the context (cached between calls).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct TmPlan;  // otmb_transportmatrix.hip

// pending plan of the general path (otmb_coo.hip): COO generator and sparse()
struct CooPlan { int which = -1; otmb_tm_args args; int64_t ntiles = 0, len = 0; };
struct SpPlan { const int64_t *I = nullptr, *J = nullptr; const double *V = nullptr; int64_t len = 0, m = 0, n = 0, nnz = 0; };

// kernel ids for the optional HIP-event timing (otmb_ctx_timing_*)
enum {
    K_TM_COUNT = 0, K_TILESCAN, K_TM_FILL, K_TM_FINISH, K_FACEFLUXES, K_IDX_COUNT, K_IDX_WRITE, K_TM_ONEPASS, K_VELFLUX, K_GM, K_GRIDMETRICS, K_NKERNELS
};
#define OTMB_TIMING_POOL 2048

struct otmb_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // own_stream or a borrowed one
    std::string err;
    // scratch for the scans / flags
    DevBuf blocksums, blockoffs, flags, lookback, tcount, tfix[3];
    DevBuf sort[5];            // radix-sort keys/values/temporary of the general sparse() path
    DevBuf tm_sums, tm_offs;  // tile sums/offsets of the pending transportmatrix plan (must survive until fill)
    int *h_flags = nullptr;  // pinned host mirror of the flag words
    i64 *h_tot = nullptr;    // pinned host mirror of scan totals
    TmPlan *plan = nullptr;
    CooPlan coo;
    SpPlan sp;
    // staging for the host-pointer entry points
    std::vector<DevBuf> stage;
    // optional per-kernel timing with HIP events recorded on the launch stream
    bool timing = false;
    std::vector<hipEvent_t> ev;   // 2 * OTMB_TIMING_POOL events, created on first enable
    std::vector<int> ev_kernel;   // kernel id of each recorded pair
    double t_ms[K_NKERNELS] = {0};
    long t_n[K_NKERNELS] = {0};
};

// Brackets one kernel launch with events when timing is on.
struct KernelTimer {
    otmb_ctx *c;
    int slot;
    KernelTimer(otmb_ctx *ctx, int kernel_id) : c(ctx), slot(-1) {
        if (c->timing && c->ev_kernel.size() < (size_t)OTMB_TIMING_POOL) {
            slot = (int)c->ev_kernel.size();
            c->ev_kernel.push_back(kernel_id);
            (void)hipEventRecord(c->ev[2 * slot], c->stream);
        }
    }
    ~KernelTimer() {
        if (slot >= 0) (void)hipEventRecord(c->ev[2 * slot + 1], c->stream);
    }
};

#define OTMB_NFLAGS 16
enum {
    FLAG_RHO_NAN = 0, FLAG_TADV_NAN, FLAG_TKH_NAN, FLAG_TKVML_NAN, FLAG_TKVDEEP_NAN,
    FLAG_FLUX_INTO_LAND, FLAG_NONCANONICAL, FLAG_LOOKBACK_TIMEOUT, FLAG_CAPACITY,
    FLAG_T_CANCEL,  // some T entry summed to exactly zero: T was written with gaps and needs compaction
    OTMB_NFLAGS_TM = 12,            // words [0, OTMB_NFLAGS_TM) belong to transportmatrix and are reset by it
    FLAG_U_VALID = 12, FLAG_V_VALID = 13  // owned by facefluxes: untouched by a transportmatrix call in between
};

int32_t otmb_fail(otmb_ctx *ctx, int32_t status, const char *detail = nullptr);
int32_t otmb_reserve(otmb_ctx *ctx, DevBuf &b, size_t bytes);
void otmb_tm_plan_free(otmb_ctx *ctx);                               // otmb_transportmatrix.hip
int32_t otmb_tm_plan_query(otmb_ctx *ctx, int64_t *nnz, int64_t *N);  // otmb_transportmatrix.hip

#define HIP_TRY(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            char msg_[256];                                                             \
            snprintf(msg_, sizeof msg_, "%s -> %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return otmb_fail((ctx), OTMB_ERR_HIP, msg_);                                \
        }                                                                               \
    } while (0)

// ---- device-side scan of per-tile sums (otmb_scan.hip) -------------------------------------
// sums: [ntiles][nf] u32 ; offs: [ntiles][nf] i64 exclusive prefix ; tot: [nf] i64 (device)
// gsum: scratch of (ntiles / 1024 + 1) * nf i64 (see otmb_scan_scratch)
void otmb_launch_tilescan(hipStream_t s, const uint32_t *sums, i64 *offs, i64 *tot, i64 ntiles, int nf, i64 *gsum);
static inline size_t otmb_scan_scratch(i64 ntiles, int nf) { return (size_t)(ntiles / 1024 + 2) * nf * sizeof(i64); }
