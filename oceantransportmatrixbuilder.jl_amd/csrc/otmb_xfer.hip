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
    if (const char *e = getenv("OTMB_XFER_NARROW")) x->narrow_ok = atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));
    if (const char *e = getenv("OTMB_XFER_NARROW_MIN_KB")) x->narrow_min = (size_t)(atoi(e) > 0 ? atoi(e) : 1) << 10;  // (tests: small arrays too)
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

static void par_widen(OtmbThreadPool *pool, int64_t *dst, const int32_t *src, size_t count) {
    const int parts = (count >= ((size_t)1 << 18)) ? pool->size() : 1;
    const size_t per = ((count + parts - 1) / parts + 1023) & ~(size_t)1023;
    pool->parallel_for(parts, [&](int p) {
        const size_t a = (size_t)p * per, b = (a + per <= count) ? a + per : count;
        for (size_t i = a; i < b; ++i) dst[i] = (int64_t)src[i];
    });
}

// dst[0 .. count - 2] = src[i + 1] - src[i] as bytes; the host rebuilds out[i + 1] = out[i] + dst[i] from out[0] = first
static void par_offsets(OtmbThreadPool *pool, int64_t *dst, const uint8_t *diffs, size_t count, int64_t first) {
    if (count == 0) return;
    const size_t nd = count - 1;
    const int parts = (nd >= ((size_t)1 << 18)) ? pool->size() : 1;
    const size_t per = ((nd + parts - 1) / parts + 1023) & ~(size_t)1023;
    std::vector<int64_t> sums(parts + 1, 0);
    pool->parallel_for(parts, [&](int p) {  // pass 1: every slice's total
        const size_t a = (size_t)p * per, b = (a + per <= nd) ? a + per : nd;
        int64_t t = 0;
        for (size_t i = a; i < b; ++i) t += diffs[i];
        sums[p + 1] = t;
    });
    for (int p = 0; p < parts; ++p) sums[p + 1] += sums[p];
    dst[0] = first;
    pool->parallel_for(parts, [&](int p) {  // pass 2: the offsets
        const size_t a = (size_t)p * per, b = (a + per <= nd) ? a + per : nd;
        int64_t run = first + sums[p];
        for (size_t i = a; i < b; ++i) { run += diffs[i]; dst[i + 1] = run; }
    });
}
// n offsets -> pieces of `per` differences each: [the piece's first offset, 8 bytes][its differences, one byte each], `stride` bytes apart
__global__ __launch_bounds__(256) void offsets_to_bytes_kernel(const int64_t *__restrict__ src, uint8_t *__restrict__ dst, size_t n, size_t per, size_t stride) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t p = i / per, r = i - p * per;
    if (r == 0) *(int64_t *)(dst + p * stride) = src[i];
    if (i + 1 < n) dst[p * stride + 8 + r] = (uint8_t)(src[i + 1] - src[i]);
}

__global__ __launch_bounds__(256) void narrow_i64_kernel(const int64_t *__restrict__ src, int32_t *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (int32_t)src[i];
}

int32_t otmb_xfer(otmb_ctx *ctx, bool to_device, const OtmbXferItem *items, int n, const std::function<void()> *link_free) {
    int32_t rc;
    if (to_device) {  // fault injection for the tests of the residency keys (a batch that never reaches the device must not be remembered)
        const char *e = getenv("OTMB_TEST_FAIL_UPLOAD");
        if (e && e[0] == '1') return otmb_fail(ctx, OTMB_ERR_HIP, "upload failure injected (OTMB_TEST_FAIL_UPLOAD)");
    }
    if ((rc = xfer_init(ctx))) return rc;
    OtmbXfer &x = *ctx->xfer;
    struct Piece { char *dev, *host; size_t bytes; int widen; size_t count; };  // widen: 0 copy, 1 Int32 -> Int64, 2 first offset + byte differences -> Int64 offsets (ONE piece)
    std::vector<Piece> pieces;
    std::vector<int> direct;  // (device -> host) copies that need no staging: issued BEHIND the first ring pieces, see below
    // narrow items: one scratch buffer of Int32 for all of them, filled by one kernel each on the stream
    size_t narrow_entries = 0;
    // offsets items (narrow == 2): pieces of `per` byte differences behind their own first offset, one ring slot each.  (With OTMB_XFER_NARROW=1
    // they travel as they are: Int32 would need a bound on the offsets themselves, which a 0.1 degree matrix passes.)
    const size_t per = x.chunk - 4096;
    auto stride_of = [&](size_t cnt) { return (cnt - 1 > per) ? x.chunk : ((8 + (cnt > 1 ? cnt - 1 : 0) + 4095) & ~(size_t)4095); };
    auto as_bytes = [&](const OtmbXferItem &it) { return it.narrow == 2 && x.narrow_ok >= 2; };
    auto is_narrow = [&](const OtmbXferItem &it) { return it.bytes >= x.narrow_min && (it.narrow == 1 || as_bytes(it)); };
    auto byte_pieces = [&](size_t cnt) { return cnt > 1 ? (cnt - 1 + per - 1) / per : (size_t)1; };
    if (!to_device && x.narrow_ok)
        for (int q = 0; q < n; ++q)
            if (is_narrow(items[q])) narrow_entries += as_bytes(items[q]) ? byte_pieces(items[q].bytes / 8) * (stride_of(items[q].bytes / 8) / 4) + 2 : items[q].bytes / 8;
    if (narrow_entries) {
        if ((rc = otmb_reserve(ctx, ctx->xfer_narrow, narrow_entries * 4))) return rc;
        size_t at = 0;
        for (int q = 0; q < n; ++q) {
            if (!is_narrow(items[q])) continue;
            const size_t cnt = items[q].bytes / 8;
            int32_t *d32 = (int32_t *)ctx->xfer_narrow.p + at;
            if (as_bytes(items[q])) {
                at = (at + 1) & ~(size_t)1;  // (8-byte aligned: every piece starts with its first offset)
                uint8_t *d8 = (uint8_t *)((int32_t *)ctx->xfer_narrow.p + at);
                const size_t stride = stride_of(cnt);
                hipLaunchKernelGGL(offsets_to_bytes_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, (const int64_t *)items[q].dev, d8, cnt, per, stride);
                const size_t np_ = byte_pieces(cnt);
                for (size_t p = 0; p < np_; ++p) {
                    const size_t a = p * per, b = (a + per + 1 <= cnt) ? a + per + 1 : cnt;  // offsets [a, b): b - a - 1 differences
                    pieces.push_back({(char *)d8 + p * stride, (char *)((int64_t *)items[q].host + a), 8 + (b - a - 1), 2, b - a});
                }
                at += np_ * (stride / 4);
                continue;
            }
            hipLaunchKernelGGL(narrow_i64_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, (const int64_t *)items[q].dev, d32, cnt);
            for (size_t off = 0; off < cnt * 4; off += x.chunk)
                pieces.push_back({(char *)d32 + off, (char *)items[q].host + 2 * off, (off + x.chunk <= cnt * 4) ? x.chunk : cnt * 4 - off, 1, 0});
            at += cnt;
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    for (int q = 0; q < n; ++q) {
        if (!items[q].bytes) continue;
        if (narrow_entries && is_narrow(items[q])) continue;  // (queued above)
        // small arrays: the runtime's own pageable path is fine; arrays inside pinned memory of otmb_host_alloc are the DMA's
        // own source / target: no staging, no host copy
        if (items[q].bytes < ((size_t)256 << 10) || otmb_host_is_pinned(ctx, items[q].host, items[q].bytes)) {
            if (to_device) HIP_TRY(ctx, hipMemcpyAsync(items[q].dev, items[q].host, items[q].bytes, hipMemcpyHostToDevice, ctx->stream));
            else direct.push_back(q);
            continue;
        }
        for (size_t off = 0; off < items[q].bytes; off += x.chunk)
            pieces.push_back({(char *)items[q].dev + off, (char *)items[q].host + off,
                              (off + x.chunk <= items[q].bytes) ? x.chunk : items[q].bytes - off, 0, 0});
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
        // The copies that go straight into pinned host memory queue up BEHIND the first ring pieces: the host threads unpack those
        // (copy, or widen Int32 -> Int64) while the DMA engine works through these.  With more pieces than slots the direct copies are
        // dealt out one per freed slot, so that the LAST things on the link are copies nobody has to unpack.
        size_t next_direct = 0;
        auto issue_direct = [&](size_t upto) -> int32_t {
            for (; next_direct < upto && next_direct < direct.size(); ++next_direct) {
                const int q = direct[next_direct];
                HIP_TRY(ctx, hipMemcpyAsync(items[q].host, items[q].dev, items[q].bytes, hipMemcpyDeviceToHost, ctx->stream));
            }
            return OTMB_OK;
        };
        const size_t per_slot = (np > NS) ? (direct.size() + (size_t)(np - NS)) / (size_t)(np - NS + 1) : direct.size();
        if ((rc = issue_direct(np > NS ? per_slot : direct.size()))) return rc;
        if (link_free && np <= NS) {
            // every DMA of this call is in the queue: wait for the last one, hand the link on, unpack afterwards
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (*link_free)();
            for (int p = 0; p < np; ++p) {
                const char *slot = x.pin + (size_t)(p % NS) * x.chunk;
                if (pieces[p].widen == 2) par_offsets(x.pool, (int64_t *)pieces[p].host, (const uint8_t *)slot + 8, pieces[p].count, *(const int64_t *)slot);
                else if (pieces[p].widen) par_widen(x.pool, (int64_t *)pieces[p].host, (const int32_t *)slot, pieces[p].bytes / 4);
                else par_memcpy(x.pool, pieces[p].host, slot, pieces[p].bytes);
            }
            return OTMB_OK;
        }
        for (int p = 0; p < np; ++p) {
            const int s = p % NS;
            HIP_TRY(ctx, hipEventSynchronize(x.ev[s]));
            const char *slot = x.pin + (size_t)s * x.chunk;
            if (pieces[p].widen == 2) par_offsets(x.pool, (int64_t *)pieces[p].host, (const uint8_t *)slot + 8, pieces[p].count, *(const int64_t *)slot);
            else if (pieces[p].widen) par_widen(x.pool, (int64_t *)pieces[p].host, (const int32_t *)slot, pieces[p].bytes / 4);
            else par_memcpy(x.pool, pieces[p].host, slot, pieces[p].bytes);
            if (issued < np) {
                if ((rc = issue(issued))) return rc;
                ++issued;
                if ((rc = issue_direct(next_direct + per_slot))) return rc;
            }
        }
        if ((rc = issue_direct(direct.size()))) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the direct copies
        if (link_free) (*link_free)();
    }
    return OTMB_OK;
}
