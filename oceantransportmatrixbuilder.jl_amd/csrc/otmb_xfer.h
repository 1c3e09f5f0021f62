// otmb_xfer.h -- host <-> device transfers of the HOST-pointer entry points (what a Julia ccall hands over is ordinary
// pageable memory).  hipMemcpy from pageable memory stages through a small internal buffer with one host thread and
// reaches ~13 GB/s on this platform; here the staging is explicit: a ring of pinned chunks, a few host threads that
// copy pageable <-> pinned in parallel, and the DMA of one chunk overlapped with the host copy of the next, which
// brings the transfers close to the PCIe rate.  Only data movement: no compute ever happens on the host.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "otmb_common.h"

struct OtmbXferItem {
    void *dev;
    void *host;
    size_t bytes;
    // (device -> host only) the array holds bytes / 8 Int64 values that the caller KNOWS to lie in [0, 2^31) -- row indices, column
    // offsets: it crosses the link as Int32 (a narrowing kernel into a scratch buffer, half the bytes) and the host threads widen it
    // into the Int64 array while the DMA engine is busy with the next pieces.  The host can widen 16 G entries/s with 8 threads
    // (tools/micro/host_widen.cpp), the link moves 7 G Int64 entries/s.  Bit-identical by construction.
    // narrow = 2 (device -> host only): the array holds COLUMN OFFSETS whose consecutive differences the caller KNOWS to lie in [0, 255] (a column
    // of the five matrices holds at most 7 entries): the first offset and one BYTE per column cross the link (an eighth of the Int32 form) and the
    // host threads rebuild the Int64 offsets by a prefix sum.  Bit-identical by construction; OTMB_XFER_NARROW=0 / =1 switch it off.
    int narrow = 0;
};

class OtmbThreadPool {
   public:
    explicit OtmbThreadPool(int n);
    ~OtmbThreadPool();
    int size() const { return (int)workers_.size() + 1; }
    // run fn(part) for part = 0..parts-1 on the pool (the caller takes part in the work); returns when all are done
    void parallel_for(int parts, const std::function<void(int)> &fn);

   private:
    void loop();
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    int next_ = 0, parts_ = 0, pending_ = 0;
    unsigned long generation_ = 0;
    bool stop_ = false;
};

struct OtmbXfer {
    static const int NSLOT = 4;
    size_t chunk = (size_t)32 << 20;  // measured at 1 degree (1.06 GB down): 8 MiB chunks 41 GB/s, 32 MiB chunks 51 GB/s
    char *pin = nullptr;  // NSLOT * chunk bytes of pinned host memory
    hipEvent_t ev[NSLOT] = {};
    OtmbThreadPool *pool = nullptr;
    int narrow_ok = 2;  // OTMB_XFER_NARROW=0: `narrow` items travel as they are; =1: column offsets as Int32 like row indices (A/B); 2: offsets as byte differences
    size_t narrow_min = (size_t)256 << 10;  // smaller arrays are not worth a kernel and a ring piece
    ~OtmbXfer();
};

// to_device: enqueue on ctx->stream; returns once every byte has been handed to the DMA engine (the caller's buffers may
// be reused).  !to_device: returns when the host buffers are filled (the stream's earlier work is waited for).
// link_free (device -> host, optional): called once when this call's last DMA has finished -- possibly before the host threads have
// unpacked the last ring pieces -- so that a caller that takes turns on the link can hand it on early.
int32_t otmb_xfer(otmb_ctx *ctx, bool to_device, const OtmbXferItem *items, int n, const std::function<void()> *link_free = nullptr);
