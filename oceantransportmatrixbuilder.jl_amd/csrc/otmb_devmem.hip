// otmb_devmem.hip -- device memory laid out for this chip's HBM (otmb_dev_alloc / otmb_dev_free).
//
// On MI355X the way a buffer is BACKED decides how fast a bandwidth-bound kernel streams through it (measured, round 4:
// tools/micro/placement_mix.hip, profiles/r04/README.md section 8).  The fill pass's byte mix (ten streams read, ten written) over
//   one hipMalloc per stream                      5.85 TB/s (1 degree sizes)   6.43 TB/s (0.25 degree sizes)
//   all streams carved out of ONE hipMalloc       5.61 (5.1 behind other data)   5.30
//   virtual ranges mapped from 2 MiB handles      6.11-6.28                     6.43-6.58
//   ... from 256 MiB handles                      5.2                           6.45
// and the real fill pass follows (all arrays in one allocation: +15-19 % time at both sizes, whatever the strides between the
// arrays; one allocation per array: what torch's allocator happens to do, with a process-to-process spread of 10-20 %).
// So the arrays the hot kernels stream through are given their own virtual ranges, mapped from SMALL physical handles
// (hipMemCreate / hipMemMap): 2 MiB handles up to 1 GiB, 32 MiB handles beyond (any size does at those sizes; fewer handles).
// One registry for the process behind a lock: blocks may be freed from any thread (a garbage collector's finalizer), with or
// without the context that allocated them.  Falls back to hipMalloc where virtual memory management is not available.
#include <mutex>
#include <unordered_map>

#include "otmb_common.h"

namespace {
struct DevBlock {
    size_t size = 0;  // mapped bytes (0: a plain hipMalloc block)
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
struct DevRegistry {
    std::mutex m;
    std::unordered_map<void *, DevBlock> blocks;
    int vmm = -1;  // -1 unknown, 0 unavailable, 1 available
};
DevRegistry &registry() {
    static DevRegistry *r = new DevRegistry();  // (leaked on purpose: finalizers may run after static destructors)
    return *r;
}
void release(void *p, DevBlock &b) {
    if (b.size == 0) {
        (void)hipFree(p);
        return;
    }
    (void)hipMemUnmap(p, b.size);
    (void)hipMemAddressFree(p, b.size);
    for (auto h : b.handles) (void)hipMemRelease(h);
}
}  // namespace

extern "C" {

int32_t otmb_dev_alloc(otmb_ctx *ctx, int64_t bytes, void **out) {
    if (!ctx || !out || bytes < 0) return otmb_fail(ctx, OTMB_ERR_INVALID_ARG, "otmb_dev_alloc");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevRegistry &reg = registry();
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    {
        std::lock_guard<std::mutex> l(reg.m);
        if (reg.vmm < 0) {
            size_t gran = 0;
            const char *e = getenv("OTMB_DEV_ALLOC");  // "malloc": plain hipMalloc blocks (experiments)
            reg.vmm = (!(e && std::string(e) == "malloc") &&
                       hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) == hipSuccess && gran > 0 &&
                       gran <= ((size_t)2 << 20)) ? 1 : 0;
            (void)hipGetLastError();
        }
    }
    const size_t want = (size_t)(bytes > 0 ? bytes : 1);
    DevBlock blk;
    blk.device = ctx->device;
    void *p = nullptr;
    if (reg.vmm == 1) {
        size_t H = want <= ((size_t)1 << 30) ? ((size_t)2 << 20) : ((size_t)32 << 20);
        if (const char *e = getenv("OTMB_DEV_ALLOC_HANDLE_MB")) { const long v = atol(e); if (v >= 2) H = (size_t)v << 20; }
        const size_t size = (want + H - 1) / H * H;
        bool ok = hipMemAddressReserve(&p, size, H, nullptr, 0) == hipSuccess;
        size_t mapped = 0;
        for (size_t off = 0; ok && off < size; off += H) {
            hipMemGenericAllocationHandle_t h;
            ok = hipMemCreate(&h, H, &prop, 0) == hipSuccess;
            if (!ok) break;
            blk.handles.push_back(h);
            ok = hipMemMap((char *)p + off, H, 0, h, 0) == hipSuccess;
            if (ok) mapped = off + H;
        }
        if (ok) {
            hipMemAccessDesc acc = {};
            acc.location.type = hipMemLocationTypeDevice;
            acc.location.id = ctx->device;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            ok = hipMemSetAccess(p, size, &acc, 1) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            if (p) {
                if (mapped) (void)hipMemUnmap(p, mapped);
                (void)hipMemAddressFree(p, size);
            }
            for (auto h : blk.handles) (void)hipMemRelease(h);
            return otmb_fail(ctx, OTMB_ERR_ALLOC, "otmb_dev_alloc: hipMemCreate / hipMemMap");
        }
        blk.size = size;
    } else {
        if (hipMalloc(&p, want) != hipSuccess) return otmb_fail(ctx, OTMB_ERR_ALLOC, "otmb_dev_alloc: hipMalloc");
        blk.size = 0;
    }
    {
        std::lock_guard<std::mutex> l(reg.m);
        reg.blocks[p] = std::move(blk);
    }
    *out = p;
    return OTMB_OK;
}

// The context argument is IGNORED (NULL and a destroyed context are fine); any thread.  The caller makes sure no kernel still uses the block.
int32_t otmb_dev_free(otmb_ctx *, void *p) {
    if (!p) return OTMB_OK;
    DevRegistry &reg = registry();
    DevBlock blk;
    {
        std::lock_guard<std::mutex> l(reg.m);
        auto it = reg.blocks.find(p);
        if (it == reg.blocks.end()) return OTMB_ERR_INVALID_ARG;
        blk = std::move(it->second);
        reg.blocks.erase(it);
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(blk.device);
    (void)hipDeviceSynchronize();  // (unmapping memory a running kernel still touches is a fault, not an error code)
    release(p, blk);
    (void)hipSetDevice(cur);
    return OTMB_OK;
}

// 1: blocks come from small physical handles (virtual memory management), 0: plain hipMalloc, -1: nothing allocated yet
int32_t otmb_dev_alloc_mode(void) { return registry().vmm; }

}  // extern "C"
