// otmb_xfer.hip -- see otmb_xfer.h
#include "otmb_xfer.h"

#include <cstdlib>

OtmbThreadPool::OtmbThreadPool(int n) {
    for (int q = 0; q < n - 1; ++q) workers_.emplace_back([this] { loop(); });
}
OtmbThreadPool::~OtmbThreadPool() {
    {
        std::lock_guard<std::mutex> l(m_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : workers_) t.join();
}
void OtmbThreadPool::loop() {
    unsigned long seen = 0;
    for (;;) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || (generation_ != seen && next_ < parts_); });
        if (stop_) return;
        seen = generation_;
        while (next_ < parts_) {
            const int part = next_++;
            const auto *fn = fn_;
            l.unlock();
            (*fn)(part);
            l.lock();
            if (--pending_ == 0) done_.notify_all();
        }
    }
}
void OtmbThreadPool::parallel_for(int parts, const std::function<void(int)> &fn) {
    if (parts <= 0) return;
    if (parts == 1 || workers_.empty()) {
        for (int p = 0; p < parts; ++p) fn(p);
        return;
    }
    std::unique_lock<std::mutex> l(m_);
    fn_ = &fn;
    parts_ = parts;
    next_ = 0;
    pending_ = parts;
    ++generation_;
    cv_.notify_all();
    while (next_ < parts_) {  // the caller works too
        const int part = next_++;
        l.unlock();
        fn(part);
        l.lock();
        --pending_;
    }
    done_.wait(l, [&] { return pending_ == 0; });
    fn_ = nullptr;
}

OtmbXfer::~OtmbXfer() {
    delete pool;
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);
    if (pin) (void)hipHostFree(pin);
}

static int32_t xfer_init(otmb_ctx *ctx) {
    if (ctx->xfer) return OTMB_OK;
    OtmbXfer *x = new OtmbXfer();
    if (const char *e = getenv("OTMB_XFER_CHUNK_MB")) x->chunk = (size_t)(atoi(e) > 0 ? atoi(e) : 32) << 20;
    if (hipHostMalloc((void **)&x->pin, OtmbXfer::NSLOT * x->chunk) != hipSuccess) {
        x->pin = nullptr;
        delete x;
        return otmb_fail(ctx, OTMB_ERR_ALLOC, "pinned staging ring");
    }
    for (auto &e : x->ev)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            delete x;
            return otmb_fail(ctx, OTMB_ERR_HIP, "hipEventCreate");
        }
    int nt = ctx->xfer_threads > 0 ? ctx->xfer_threads : 8;
    if (const char *e = getenv("OTMB_XFER_THREADS")) nt = atoi(e);
    const int hw = (int)std::thread::hardware_concurrency();
    if (hw > 0 && nt > hw) nt = hw;
    if (nt < 1) nt = 1;
    x->pool = new OtmbThreadPool(nt);
    ctx->xfer = x;
    return OTMB_OK;
}

static void par_memcpy(OtmbThreadPool *pool, char *dst, const char *src, size_t bytes) {
    const int parts = (bytes >= ((size_t)1 << 20)) ? pool->size() : 1;
    const size_t per = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
    pool->parallel_for(parts, [&](int p) {
        const size_t a = (size_t)p * per;
        if (a < bytes) memcpy(dst + a, src + a, (a + per <= bytes) ? per : bytes - a);
    });
}

int32_t otmb_xfer(otmb_ctx *ctx, bool to_device, const OtmbXferItem *items, int n) {
    int32_t rc;
    if (to_device) {  // fault injection for the tests of the residency keys (a batch that never reaches the device must not be remembered)
        const char *e = getenv("OTMB_TEST_FAIL_UPLOAD");
        if (e && e[0] == '1') return otmb_fail(ctx, OTMB_ERR_HIP, "upload failure injected (OTMB_TEST_FAIL_UPLOAD)");
    }
    if ((rc = xfer_init(ctx))) return rc;
    OtmbXfer &x = *ctx->xfer;
    struct Piece { char *dev, *host; size_t bytes; };
    std::vector<Piece> pieces;
    for (int q = 0; q < n; ++q) {
        if (!items[q].bytes) continue;
        // small arrays: the runtime's own pageable path is fine; arrays inside pinned memory of otmb_host_alloc are the DMA's
        // own source / target: no staging, no host copy
        if (items[q].bytes < ((size_t)256 << 10) || otmb_host_is_pinned(ctx, items[q].host, items[q].bytes)) {
            HIP_TRY(ctx, to_device ? hipMemcpyAsync(items[q].dev, items[q].host, items[q].bytes, hipMemcpyHostToDevice, ctx->stream)
                                   : hipMemcpyAsync(items[q].host, items[q].dev, items[q].bytes, hipMemcpyDeviceToHost, ctx->stream));
            continue;
        }
        for (size_t off = 0; off < items[q].bytes; off += x.chunk)
            pieces.push_back({(char *)items[q].dev + off, (char *)items[q].host + off,
                              (off + x.chunk <= items[q].bytes) ? x.chunk : items[q].bytes - off});
    }
    const int np = (int)pieces.size(), NS = OtmbXfer::NSLOT;
    if (to_device) {
        for (int p = 0; p < np; ++p) {
            const int s = p % NS;
            if (p >= NS) HIP_TRY(ctx, hipEventSynchronize(x.ev[s]));  // the DMA that read this slot has finished
            par_memcpy(x.pool, x.pin + (size_t)s * x.chunk, pieces[p].host, pieces[p].bytes);
            HIP_TRY(ctx, hipMemcpyAsync(pieces[p].dev, x.pin + (size_t)s * x.chunk, pieces[p].bytes, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(x.ev[s], ctx->stream));
        }
        // the ring is reused by the next call: its last DMAs must have read their slots
        for (int s = 0; s < NS && s < np; ++s) HIP_TRY(ctx, hipEventSynchronize(x.ev[s]));
    } else {
        int issued = 0;
        auto issue = [&](int p) -> int32_t {
            const int s = p % NS;
            HIP_TRY(ctx, hipMemcpyAsync(x.pin + (size_t)s * x.chunk, pieces[p].dev, pieces[p].bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(x.ev[s], ctx->stream));
            return OTMB_OK;
        };
        for (; issued < np && issued < NS; ++issued)
            if ((rc = issue(issued))) return rc;
        for (int p = 0; p < np; ++p) {
            const int s = p % NS;
            HIP_TRY(ctx, hipEventSynchronize(x.ev[s]));
            par_memcpy(x.pool, pieces[p].host, x.pin + (size_t)s * x.chunk, pieces[p].bytes);
            if (issued < np) {
                if ((rc = issue(issued))) return rc;
                ++issued;
            }
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the small direct copies
    }
    return OTMB_OK;
}
