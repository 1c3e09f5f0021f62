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
struct OtmbXfer;  // otmb_xfer.h: pinned staging ring + host copy threads of the host-pointer entry points

// pending plan of the general path (otmb_coo.hip): COO generator and sparse()
struct CooPlan { int which = -1; otmb_tm_args args; int64_t ntiles = 0, len = 0; };
struct SpPlan { const int64_t *I = nullptr, *J = nullptr; const double *V = nullptr; int64_t len = -1, m = 0, n = 0, nnz = 0; int rowbits = 32; };  // rowbits: the sort keys are (column << rowbits) | row

// kernel ids for the optional HIP-event timing (otmb_ctx_timing_*)
enum {
    K_TM_COUNT = 0, K_TILESCAN, K_TM_FILL, K_TM_FINISH, K_FACEFLUXES, K_IDX_COUNT, K_IDX_WRITE, K_VELFLUX, K_GM, K_GRIDMETRICS, K_PUSHMASK, K_TM_ORDER, K_FF_BASES, K_NKERNELS
};
#define OTMB_TIMING_POOL 2048

struct otmb_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // own_stream or a borrowed one
    std::string err;
    // scratch for the scans / flags
    DevBuf blocksums, blockoffs, flags, stamps, tcount, tfix[3];  // (stamps: diagnostic builds, OTMB_DBG_STAMPS)
    DevBuf sort[5];            // radix-sort keys/values/temporary of the general sparse() path
    DevBuf tm_sums, tm_offs;  // tile sums/offsets of the pending transportmatrix plan (must survive until fill)
    DevBuf mask;              // push mask derived by the library when the caller passes none
    DevBuf order;             // tile order of the fill pass (march order: otmb_ctx_set_tile_order) + its bucket scratch
    int march_rows = -1;      // rows per band of the march order; 0 = wet-rank order; -1 = the library's default
    int march_cols = -1;      // columns (i) per block of a band; 0 = whole rows; -1 = the library's default (experiments: OTMB_MARCH_COLS)
    struct OrderKey {
        const void *lwet = nullptr; int64_t n = 0, nx = 0, ny = 0, nz = 0; int rows = 0, topo = -1, cols = 0;
        bool operator==(const OrderKey &o) const { return lwet == o.lwet && n == o.n && nx == o.nx && ny == o.ny && nz == o.nz && rows == o.rows && topo == o.topo && cols == o.cols; }
    } order_key;              // what ctx->order was built for
    int deal_heavy = 1;         // 0: the heavy tiles stay in the first XCD's share (experiments: OTMB_DEAL_HEAVY)
    unsigned order_nheavy = 0;  // the order's first entries are this many heavy tiles (tripolar seam row), dealt over the XCDs
    int ff_xcd_chunks = 1;    // facefluxes: XCD x takes the x-th contiguous eighth of the column blocks (0 = blockIdx order; experiments: OTMB_FF_XCD)
    int ff_nt = -1;           // facefluxes' ϕ stores non-temporal (1), plain (0), by size (-1: beyond a gigabyte of fluxes) (experiments: OTMB_FF_NT)
    int ff_lds_south = 1;     // four-row facefluxes workgroups take a wave's south row from the neighbouring wave through LDS (experiments: OTMB_FF_LDS_SOUTH)
    int ff_rows = 0;          // facefluxes: rows per workgroup, 1 or 4; 0 = by grid size (experiments: OTMB_FF_ROWS)
    int count_order = 2;      // counting pass: 0 = blockIdx (wet-rank) order, 1 = XCD-contiguous eighths of wet-rank order, 2 = of the fill pass's tile order (default: HBM fetch 1.06-1.11 x its inputs instead of 2.1-2.7 x, +2-4 % of this pass's time; OTMB_COUNT_ORDER)
    // ---- counts in facefluxes (otmb_facefluxes_counts_dev): the five per-tile row counts of the transportmatrix that will be built from the
    // fluxes a facefluxes call is writing are accumulated by that call (packed 64-bit atomics, one word per tile of 256 columns), so the
    // device-resident step has no counting pass.  Two buffers: the call after next may run beside the fill pass that consumes this one.
    int count_in_ff = 1;      // 0: otmb_facefluxes_counts_dev behaves as otmb_facefluxes_flags_dev (experiments / A-B: OTMB_COUNT_IN_FF)
    DevBuf ffc_sums[2];
    bool ffc_dirty[2] = {true, true};  // the buffer is not known to be all zero
    int ffc_next = 0;
    struct FfCountsKey {      // what the pending counts describe; consumed (valid = false) by the transportmatrix that uses them
        bool valid = false;
        int buf = 0, gen = 0;
        const void *phi[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        const void *mask = nullptr, *mlotst = nullptr, *zt = nullptr, *lwet3d = nullptr;
        const void *stat = nullptr;  // the grid's static per-tile counts (otmb_count_tables_dev), added by the scan
        int64_t nx = 0, ny = 0, nz = 0, n_wet = 0;  // (nz: of the grid the transportmatrix sees -- a depth slab's extended local grid)
        int64_t k_own0 = 0, wet_base = 0;           // depth slab: phi / mask name its owned levels, k_own0 levels into the extended arrays
        int topo = -1, upwind = -1, only_t = -1;
        bool pieces_open = false;  // the last facefluxes call counted: a further row band of it (otmb_facefluxes_slab_counts_dev, first = 0) adds to the same buffer
    } ffc;
    const void *ffc_partial_mask = nullptr;  // the push_mask argument of the last counting facefluxes call: NOT written by it, never a counting pass's input
    DevBuf lump[11];          // lump_and_spray scratch (otmb_lump.hip)
    DevBuf lump_host;         // staging of the host-pointer entry point
    bool lump_valid = false;
    i64 lump_N = 0, lump_Nc = 0;
    int *h_flags = nullptr;  // pinned host mirror of the state block (flag words first)
    i64 *h_tot = nullptr;    // the totals inside it (h_flags + OTMB_NFLAGS)
    int ff_gen = 0;          // facefluxes call counter: a validity flag is set by writing the current value (no reset pass)
    int ff_first = 1;        // oldest facefluxes call whose validity flags have not been handed out (otmb_facefluxes_pending_flags)
    // asynchronous pipeline: every otmb_transportmatrix_dev call gets its own state block (flags + totals) and every
    // facefluxes call its own pair of validity words, in rings of OTMB_RING slots (device) mirrored in pinned host
    // memory, so that a failure in step 3 of 12 is still there when the host finally looks (otmb_transportmatrix_result)
    DevBuf ring;
    int *h_ring = nullptr;
    unsigned long long ring_clean = 0;  // ring slots (bit per slot) known to be zero on the device and unused
    i64 tm_first = 0, tm_next = 0;   // pending asynchronous transportmatrix steps [tm_first, tm_next)
    int32_t tm_sticky = 0;           // first failure folded out of the ring when it wrapped (status, step index)
    i64 tm_sticky_step = -1;
    std::string tm_sticky_msg;
    i64 tm_failed_step = -1;         // what the last otmb_transportmatrix_result found
    struct TmStepRec { void *colptrT, *rowvalT, *nzvalT; i64 n_wet, nnz_base0; int ignore_ops; };
    std::vector<TmStepRec> tm_rec;   // T's output arrays of the pending steps [tm_first, tm_next) (compaction after exact cancellation)
    struct TmStepResult { int32_t status; i64 nnz[5]; };
    std::vector<TmStepResult> tm_hist;  // verdict and nnz of every step since the previous otmb_transportmatrix_result
    bool tm_hist_final = false;      // tm_hist describes a finished pipeline (cleared by the next otmb_transportmatrix_dev)
    int gm_lds_limit = -1;    // bolus_GM_velocity: LDS bytes the fused kernel may use (-1: not queried yet; 0: use the two streaming kernels)
    size_t gm_lds_set = 0;    // ... and the dynamic-LDS size its function attribute currently allows
    TmPlan *plan = nullptr;
    // ---- otmb_tm_args.given: is a given TκH / TκVdeep bit for bit what the fill pass derives?  One verdict per operator, keyed to every
    // array address and scalar the answer depends on, and to an epoch that otmb_ctx_forget_given (and every host upload of such an array) bumps.
    struct GivenVerdict {
        bool valid = false, derived = false, pattern = false;  // pattern: the same rows in the same order, other values (another κ)
        uint64_t epoch = 0;
        otmb_csc g = {nullptr, nullptr, nullptr, 0};
        const void *lwet3d = nullptr, *lwet = nullptr, *v3d = nullptr, *thk = nullptr, *edge[4] = {nullptr, nullptr, nullptr, nullptr},
                   *dist[4] = {nullptr, nullptr, nullptr, nullptr}, *area = nullptr, *zt = nullptr;
        int64_t nx = 0, ny = 0, nz = 0, n_wet = 0, wet_base = 0;
        int topo = -1;
        double kappa = 0.0;
    } given_verdict[5];
    uint64_t given_epoch = 1;
    int given_state[5] = {0, 0, 0, 0, 0};  // the last plan's treatment of operator m: 0 not given, 1 derived, 2 foreign, 3 derived pattern with other values (otmb_ctx_given_state)
    long given_checks = 0;                 // comparing passes run so far (tests: the verdict is cached)
    DevBuf given_tmp[6];                   // temporaries of the foreign path's sparse adds: two (colptr, rowval, nzval) triples
    CooPlan coo;
    SpPlan sp;
    // staging for the host-pointer entry points
    std::vector<DevBuf> stage;
    OtmbXfer *xfer = nullptr;
    DevBuf xfer_narrow;       // Int32 copies of the `narrow` items of a download (otmb_xfer.h)
    int xfer_threads = 0;     // host copy threads of this context's transfer engine; 0 = default (8, OTMB_XFER_THREADS); otmb_mgpu shares the cores among its slabs
    bool reuse_grid = false;  // otmb_ctx_set_reuse_grid: grid-constant host arrays are uploaded once (see include/otmb.h)
    i64 uploaded_bytes = 0;     // host -> device bytes of the host-pointer entry points (otmb_ctx_uploaded_bytes)
    bool reuse_fluxes = false;  // otmb_ctx_set_reuse_fluxes: ϕ that otmb_facefluxes left in the staging slots is not uploaded again
    struct StageKey { const void *host = nullptr; size_t bytes = 0; };
    std::vector<StageKey> stage_key;  // what each staging slot currently holds (host pointer it was uploaded from)
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
    FLAG_FLUX_INTO_LAND, FLAG_NONCANONICAL, FLAG_GIVEN_MISMATCH /* the comparing pass (otmb_tm_args.given): an entry differs */, FLAG_CAPACITY,
    FLAG_T_CANCEL,  // some T entry summed to exactly zero: T was written with gaps and needs compaction
    FLAG_COUNT_MISMATCH,  // a tile's fill pass found other counts than its counting pass: push_mask does not describe ϕ
    OTMB_NFLAGS_TM = 12,            // words [0, OTMB_NFLAGS_TM) belong to transportmatrix and are reset by it
    // owned by facefluxes, behind the 16 scan totals: untouched by transportmatrix, which resets and fetches the
    // flag words and its totals as ONE block (one fill and one copy kernel per call instead of two of each)
    FLAG_U_VALID = 16 + 2 * 16, FLAG_V_VALID = FLAG_U_VALID + 1
};
// device state block (ctx->flags) and its pinned host mirror (ctx->h_flags): 16 flag words | 16 i64 totals | 4 words
#define OTMB_STATE_BYTES (OTMB_NFLAGS * sizeof(int) + 16 * sizeof(i64) + 4 * sizeof(int))
#define OTMB_TM_STATE_BYTES (OTMB_NFLAGS * sizeof(int) + 8 * sizeof(i64))  // flags + the totals transportmatrix uses
#define OTMB_RING 64
#define OTMB_RING_BYTES (OTMB_RING * OTMB_TM_STATE_BYTES + OTMB_RING * 2 * sizeof(int))
static inline int *otmb_ring_tm(int *ring, i64 step) { return (int *)((char *)ring + (size_t)(step % OTMB_RING) * OTMB_TM_STATE_BYTES); }
static inline int *otmb_ring_ff(int *ring, int gen) { return (int *)((char *)ring + OTMB_RING * OTMB_TM_STATE_BYTES) + 2 * (gen % OTMB_RING); }

int32_t otmb_fail(otmb_ctx *ctx, int32_t status, const char *detail = nullptr);
int32_t otmb_reserve(otmb_ctx *ctx, DevBuf &b, size_t bytes);
// counts in facefluxes: bit layout of a tile's packed word (the block scan's: T:11 | Tadv:11 | TκH:11 | TκVML:10 | TκVdeep:10) + two flag bits
#define FFC_TILE_SHIFT 8                  // tiles of 256 columns (TM_THREADS)
#define FFC_BAD_FLUX (1ull << 63)         // some cell of the tile pushes a non-zero flux into land / out of the grid
#define FFC_BAD_TABLE (1ull << 62)        // the bases table was built for another wave geometry
void otmb_launch_tilescan_packed(hipStream_t s, unsigned long long *packed, const unsigned long long *stat, uint32_t *sums, i64 *offs, i64 *tot, i64 *gsum, i64 ntiles,
                                 int *flags, unsigned long long keep, bool all_levels);  // otmb_scan.hip (keep: count fields of the matrices that are materialised)
int32_t otmb_launch_push_mask(otmb_ctx *ctx, const double *const phi[6], const int64_t *lwet3d, int64_t first, int64_t count,
                              uint16_t *push_mask);  // otmb_facefluxes.hip
int32_t otmb_facefluxes_top_counts(otmb_ctx *ctx, const void *umo, const void *vmo, int32_t src_is_f32, const uint8_t *wetflags, double fill,
                                   int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *const phi[6], uint16_t *token,
                                   const otmb_ff_counts *counts);  // otmb_facefluxes.hip (the fused step)
void otmb_tm_plan_free(otmb_ctx *ctx);                               // otmb_transportmatrix.hip
void otmb_tm_plan_invalidate(otmb_ctx *ctx);                         // otmb_transportmatrix.hip
void otmb_xfer_free(otmb_ctx *ctx);                                  // otmb_host.hip
bool otmb_host_is_pinned(const otmb_ctx *ctx, const void *p, size_t bytes);  // otmb_host.hip: inside a block of otmb_host_alloc
int32_t otmb_tm_plan_query(otmb_ctx *ctx, int64_t *nnz, int64_t *N);  // otmb_transportmatrix.hip
bool otmb_tm_plan_foreign(otmb_ctx *ctx);                             // otmb_transportmatrix.hip: T of the last plan came out of the sparse adds
unsigned otmb_tm_plan_skip(otmb_ctx *ctx);                            // otmb_transportmatrix.hip: matrices (bit m) the pending plan does not hand out

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
#define OTMB_SCAN_GROUP 1024  // tiles per first-level scan group (SCAN_THREADS of otmb_scan.hip)
void otmb_launch_tilescan_groups(hipStream_t s, const uint32_t *sums, i64 *offs, i64 *gsum, i64 ntiles, int nf);
static inline size_t otmb_scan_scratch(i64 ntiles, int nf) { return (size_t)(ntiles / 1024 + 2) * nf * sizeof(i64); }

// ---- push mask (include/otmb.h, otmb_push_mask_dev): bits 0-5 west, east, south, north, bottom, top;
// bit 6 wet; the centred-weighting variant in the high byte.  x / 2 is the reference's ϕ / 2 (:244-289).
enum { PM_W = 1u << 0, PM_E = 1u << 1, PM_S = 1u << 2, PM_N = 1u << 3, PM_B = 1u << 4, PM_T = 1u << 5, PM_WET = 1u << 6 };
__device__ __forceinline__ unsigned otmb_push_bits(double w, double e, double s, double n, double b, double t, bool wet) {
    const double hw = w / 2, he = e / 2, hs = s / 2, hn = n / 2, hb = b / 2, ht = t / 2;
    const unsigned lo = (w > 0.0 ? PM_W : 0u) | (e < 0.0 ? PM_E : 0u) | (s > 0.0 ? PM_S : 0u) | (n < 0.0 ? PM_N : 0u) |
                        (b > 0.0 ? PM_B : 0u) | (t < 0.0 ? PM_T : 0u) | (wet ? PM_WET : 0u);
    const unsigned hi = (((hw > 0.0) | (hw < 0.0)) ? PM_W : 0u) | (((he > 0.0) | (he < 0.0)) ? PM_E : 0u) |
                        (((hs > 0.0) | (hs < 0.0)) ? PM_S : 0u) | (((hn > 0.0) | (hn < 0.0)) ? PM_N : 0u) |
                        (((hb > 0.0) | (hb < 0.0)) ? PM_B : 0u) | (((ht > 0.0) | (ht < 0.0)) ? PM_T : 0u) | (wet ? PM_WET : 0u);
    return lo | (hi << 8);
}
