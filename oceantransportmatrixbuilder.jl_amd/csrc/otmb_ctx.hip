// otmb_ctx.hip -- context, error strings, scratch buffers.
#include <cstdlib>

#include "otmb_common.h"

int32_t otmb_fail(otmb_ctx *ctx, int32_t status, const char *detail) {
    if (ctx) {
        ctx->err = otmb_status_string(status);
        if (detail && *detail) {
            ctx->err += ": ";
            ctx->err += detail;
        }
    }
    return status;
}

int32_t otmb_reserve(otmb_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return OTMB_OK;
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        return otmb_fail(ctx, OTMB_ERR_ALLOC, "hipMalloc");
    }
    b.cap = want;
    return OTMB_OK;
}

extern "C" {

const char *otmb_version(void) { return "otmb_hip 0.1.0 (gfx950)"; }

const char *otmb_status_string(int32_t s) {
    switch (s) {
        case OTMB_OK: return "ok";
        case OTMB_ERR_RHO_NAN: return "ρ contains NaNs";
        case OTMB_ERR_TADV_NAN: return "Tadv contains NaNs.";
        case OTMB_ERR_TKH_NAN: return "TκH contains NaNs.";
        case OTMB_ERR_TKVML_NAN: return "TκVML contains NaNs.";
        case OTMB_ERR_TKVDEEP_NAN: return "TκVdeep contains NaNs.";
        case OTMB_ERR_FLUX_INTO_LAND: return "non-zero flux into a land cell or outside the grid";
        case OTMB_ERR_UNKNOWN_TOPOLOGY: return "Unknown grid type";
        case OTMB_ERR_ALL_MISSING: return "AssertionError: all umo or vmo values are NaN or _FillValue";
        case OTMB_ERR_ALLOC: return "device allocation failed";
        case OTMB_ERR_HIP: return "HIP error";
        case OTMB_ERR_INVALID_ARG: return "invalid argument";
        case OTMB_ERR_NO_PLAN: return "no transportmatrix plan";
        case OTMB_ERR_NONCANONICAL_INDICES: return "Lwet3D is not the wet rank in linear-index order (makeindices)";
        case OTMB_ERR_CAPACITY: return "output capacity too small";
        case OTMB_ERR_PUSH_MASK: return "push_mask does not describe these face fluxes and wet mask";
        case OTMB_ERR_ASYMMETRIC_PATTERN: return "ArgumentError: Adjacency / distance matrices must be symmetric";
        case OTMB_ERR_GIVEN_FOREIGN: return "a given operator is not what the library derives for these arguments: use the two-phase protocol";
        default: return "unknown status";
    }
}

int32_t otmb_ctx_create(int32_t device_id, otmb_ctx **out) {
    if (!out) return OTMB_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return OTMB_ERR_HIP;
    if (device_id < 0 || device_id >= n) return OTMB_ERR_INVALID_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return OTMB_ERR_HIP;
    otmb_ctx *c = new otmb_ctx();
    c->device = device_id;
    if (const char *e = getenv("OTMB_MARCH_ROWS")) c->march_rows = atoi(e);  // experiments; otmb_ctx_set_tile_order is the API
    if (const char *e = getenv("OTMB_MARCH_COLS")) c->march_cols = atoi(e);  // experiments
    if (const char *e = getenv("OTMB_FF_XCD")) c->ff_xcd_chunks = atoi(e);       // experiments (A/B in one library)
    if (const char *e = getenv("OTMB_FF_ROWS")) c->ff_rows = atoi(e);
    if (const char *e = getenv("OTMB_FF_LDS_SOUTH")) c->ff_lds_south = atoi(e);
    if (const char *e = getenv("OTMB_FF_NT")) c->ff_nt = atoi(e);
    if (const char *e = getenv("OTMB_COUNT_ORDER")) c->count_order = atoi(e);
    if (const char *e = getenv("OTMB_DEAL_HEAVY")) c->deal_heavy = atoi(e);
    if (const char *e = getenv("OTMB_COUNT_IN_FF")) c->count_in_ff = atoi(e);  // A/B in one library
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return OTMB_ERR_HIP;
    }
    c->stream = c->own_stream;
    if (hipHostMalloc((void **)&c->h_flags, OTMB_STATE_BYTES) != hipSuccess) {
        otmb_ctx_destroy(c);
        return OTMB_ERR_ALLOC;
    }
    memset(c->h_flags, 0, OTMB_STATE_BYTES);
    c->h_tot = (i64 *)(c->h_flags + OTMB_NFLAGS);
    if (otmb_reserve(c, c->flags, OTMB_STATE_BYTES) != OTMB_OK || hipMemset(c->flags.p, 0, OTMB_STATE_BYTES) != hipSuccess) {
        otmb_ctx_destroy(c);
        return OTMB_ERR_ALLOC;
    }
    if (hipHostMalloc((void **)&c->h_ring, OTMB_RING_BYTES) != hipSuccess) {
        otmb_ctx_destroy(c);
        return OTMB_ERR_ALLOC;
    }
    memset(c->h_ring, 0, OTMB_RING_BYTES);
    if (otmb_reserve(c, c->ring, OTMB_RING_BYTES) != OTMB_OK || hipMemset(c->ring.p, 0, OTMB_RING_BYTES) != hipSuccess) {
        otmb_ctx_destroy(c);
        return OTMB_ERR_ALLOC;
    }
    static_assert(OTMB_RING == 64, "ring_clean holds one bit per ring slot");
    c->ring_clean = ~0ull;
    *out = c;
    return OTMB_OK;
}

void otmb_ctx_destroy(otmb_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    otmb_tm_plan_free(ctx);
    otmb_xfer_free(ctx);
    for (DevBuf *b : {&ctx->blocksums, &ctx->blockoffs, &ctx->flags, &ctx->ring, &ctx->stamps, &ctx->tcount, &ctx->tfix[0], &ctx->tfix[1], &ctx->tfix[2], &ctx->tm_sums, &ctx->tm_offs, &ctx->sort[0], &ctx->sort[1], &ctx->sort[2], &ctx->sort[3], &ctx->sort[4], &ctx->ffc_sums[0], &ctx->ffc_sums[1], &ctx->xfer_narrow, &ctx->given_tmp[0], &ctx->given_tmp[1], &ctx->given_tmp[2], &ctx->given_tmp[3], &ctx->given_tmp[4], &ctx->given_tmp[5]})
        if (b->p) (void)hipFree(b->p);
    for (DevBuf &b : ctx->stage)
        if (b.p) (void)hipFree(b.p);
    for (DevBuf &b : ctx->lump)
        if (b.p) (void)hipFree(b.p);
    if (ctx->mask.p) (void)hipFree(ctx->mask.p);
    if (ctx->order.p) (void)hipFree(ctx->order.p);
    if (ctx->lump_host.p) (void)hipFree(ctx->lump_host.p);
    // (pinned blocks of otmb_host_alloc are NOT freed here: they belong to the caller's arrays and to a process-wide pool, otmb_host.hip)
    for (auto &e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int32_t otmb_ctx_set_stream(otmb_ctx *ctx, void *s) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    hipStream_t ns = s ? (hipStream_t)s : ctx->own_stream;
    if (ns != ctx->stream) ctx->order_key = otmb_ctx::OrderKey();  // the cached tile order may still be in flight on the previous stream: build it again here
    ctx->stream = ns;
    return OTMB_OK;
}

int32_t otmb_ctx_use_default_stream(otmb_ctx *ctx) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    if (ctx->stream != nullptr) ctx->order_key = otmb_ctx::OrderKey();
    ctx->stream = nullptr;  // HIP's null stream: what torch calls its default stream
    return OTMB_OK;
}

int32_t otmb_ctx_set_tile_order(otmb_ctx *ctx, int32_t rows_per_band) {
    if (!ctx || rows_per_band < -1) return OTMB_ERR_INVALID_ARG;
    ctx->march_rows = rows_per_band;
    return OTMB_OK;
}

int32_t otmb_ctx_forget_given(otmb_ctx *ctx) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    ctx->given_epoch += 1;
    return OTMB_OK;
}

int32_t otmb_ctx_given_state(const otmb_ctx *ctx, int32_t m) { return (ctx && m >= 0 && m < 5) ? ctx->given_state[m] : -1; }
int64_t otmb_ctx_given_checks(const otmb_ctx *ctx) { return ctx ? (int64_t)ctx->given_checks : -1; }

int32_t otmb_ctx_synchronize(otmb_ctx *ctx) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_flags[FLAG_NONCANONICAL]) {
        ctx->h_flags[FLAG_NONCANONICAL] = 0;
        return otmb_fail(ctx, OTMB_ERR_NONCANONICAL_INDICES);
    }
    return OTMB_OK;
}

int32_t otmb_ctx_timing_enable(otmb_ctx *ctx, int32_t on) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (on && ctx->ev.empty()) {
        ctx->ev.resize(2 * OTMB_TIMING_POOL);
        for (auto &e : ctx->ev) HIP_TRY(ctx, hipEventCreate(&e));
    }
    ctx->timing = on != 0;
    return OTMB_OK;
}

int32_t otmb_ctx_timing_collect(otmb_ctx *ctx, double *ms_sum, int64_t *count, int32_t n) {
    if (!ctx) return OTMB_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t s = 0; s < ctx->ev_kernel.size(); ++s) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[2 * s], ctx->ev[2 * s + 1]) == hipSuccess) {
            ctx->t_ms[ctx->ev_kernel[s]] += ms;
            ctx->t_n[ctx->ev_kernel[s]] += 1;
        }
    }
    ctx->ev_kernel.clear();
    for (int k = 0; k < n && k < K_NKERNELS; ++k) {
        if (ms_sum) ms_sum[k] = ctx->t_ms[k];
        if (count) count[k] = ctx->t_n[k];
        ctx->t_ms[k] = 0;
        ctx->t_n[k] = 0;
    }
    return OTMB_OK;
}

const char *otmb_kernel_name(int32_t k) {
    static const char *names[K_NKERNELS] = {"tm_count_kernel", "tilescan_kernel", "tm_kernel<fill>", "tm_finish_colptr",
                                            "facefluxes_kernel", "indices_kernel<count>", "indices_kernel<write>",
                                            "velocity_flux_kernel", "gm_slopes+gm_dyad", "gridmetrics2d+3d",
                                            "push_mask_kernel", "tm_order_kernels", "ff_count_bases_kernel"};
    return (k >= 0 && k < K_NKERNELS) ? names[k] : "";
}

const char *otmb_last_error(const otmb_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

}  // extern "C"
