// otmb_mgpu.hip -- depth slabs over several GPUs of ONE process, behind the C ABI (HOST pointers: what Julia's ccall hands over).
//
// Nothing distributed exists in the reference; this is SURVEY.md section 8(e) for a caller of the boundary: `transportmatrix(...;
// devices = 0:7)`.  The wet index is k-slowest (src/matrixbuilding.jl:14-15), so a slab of consecutive levels owns a contiguous
// column range of every matrix and a contiguous range of every (nx,ny,nz) host array:
//   * the library cuts the levels with otmb_balanced_partition (wet counts as even as a sweep gets them), owns one context per
//     device and one host thread per slab, and moves every slab over ITS device's PCIe link -- the host-pointer path is PCIe-bound
//     (1.06 GB down at 1 degree), so N links are the one lever left on it;
//   * facefluxes: the bottom-up continuity recurrence (src/velocities.jl:236-243) is a chain in k with a fixed association; the
//     slabs run it deepest first and hand ONE (nx,ny) plane of ϕtop per boundary to the slab above -- never re-associated, so the
//     fluxes stay bit-identical to a single-device run.  The hand-off is a grouped ncclSend / ncclRecv on a single-process
//     communicator (ncclCommInitAll; RCCL over xGMI) when the two slabs live on different devices, and a device-to-device
//     hipMemcpyAsync when they share one (tests on a one-GPU box; never selected between different devices);
//   * transportmatrix: with the whole grid in host memory a slab's halo levels (one above, one below: they act as neighbours only)
//     are just the adjacent levels of the same host arrays, so every slab uploads levels [k0 - 1, k1 + 1) and nothing crosses
//     devices; lwet3d carries GLOBAL wet ranks already.  Each slab writes its column range with global colptr offsets
//     (otmb_transportmatrix_set_slab / _set_nnz_base) straight into its range of the caller's five CSC matrices.
// No COO triplet or matrix entry ever crosses devices (gather formulation: a column is built by the owner of its cell).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "otmb_common.h"
#include "otmb_xfer.h"

namespace {

// ---- RCCL, loaded on demand (a one-device run never needs it; libotmb_hip.so does not link against it) ----
struct Rccl {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string &err) {
        if (h) return true;
        h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) { err = std::string("librccl.so: ") + dlerror(); return false; }
#define OTMB_SYM(field, name)                                                         \
    field = (decltype(field))dlsym(h, name);                                          \
    if (!field) { err = std::string("librccl.so lacks ") + name; dlclose(h); h = nullptr; return false; }
        OTMB_SYM(CommInitAll, "ncclCommInitAll")
        OTMB_SYM(CommDestroy, "ncclCommDestroy")
        OTMB_SYM(GroupStart, "ncclGroupStart")
        OTMB_SYM(GroupEnd, "ncclGroupEnd")
        OTMB_SYM(Send, "ncclSend")
        OTMB_SYM(Recv, "ncclRecv")
        OTMB_SYM(GetErrorString, "ncclGetErrorString")
#undef OTMB_SYM
        return true;
    }
};

enum { B_PHI0 = 0, B_V = 6, B_THK, B_RHO, B_LW, B_LWET, B_EDGE0, B_DIST0 = B_EDGE0 + 4, B_AREA = B_DIST0 + 4, B_ZT, B_ML, B_UMO, B_VMO,
       B_WET, B_PLANE, B_COLPTR0, B_ROWVAL0 = B_COLPTR0 + 5, B_NZVAL0 = B_ROWVAL0 + 5,
       B_GIVEN0 = B_NZVAL0 + 5,  // the slab's columns of the operators the caller passes (otmb_tm_args.given): colptr, rowval, nzval of m = 1 .. 4
       B_COUNT = B_GIVEN0 + 12 };

struct Slab {
    otmb_ctx *ctx = nullptr;
    int device = 0;
    i64 k0 = 0, k1 = 0;      // owned levels
    int ha = 0, hb = 0;      // halo level above / below
    i64 wet_base = 0, n_own = 0;
    DevBuf buf[B_COUNT];
    i64 nnz[5] = {0, 0, 0, 0, 0}, base[5] = {0, 0, 0, 0, 0};
    int32_t status = OTMB_OK;
    std::string msg;
    int32_t u_valid = 0, v_valid = 0;
    // what the staging buffers hold (otmb_mgpu_set_reuse): the host array a buffer was uploaded from, its bytes and the first level
    struct Key { const void *host = nullptr; size_t bytes = 0; i64 e0 = -1; };
    Key key[B_COUNT];
    bool phi_resident = false;       // B_PHI0.. hold what otmb_mgpu_facefluxes computed for levels [k0 - ha, k1 + hb) ...
    const void *phi_host[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // ... and copied to these host arrays
    i64 uploaded = 0;                // bytes copied to this slab's device so far
    // hand-off of the chain's plane to this slab (from the slab below it), piece by piece: pieces_ready of the C row bands have been
    // enqueued (RCCL: the receive sits on THIS slab's stream) or copied behind piece_ev[c] (copies: recorded on the producer's stream)
    std::mutex m;
    std::condition_variable cv;
    int pieces_ready = 0;
    bool plane_failed = false;
    std::vector<hipEvent_t> piece_ev;  // events of THIS slab as a producer: its c-th piece has left for the slab above
};

__global__ void shift_i64_kernel(i64 *p, i64 n, i64 delta) {
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) p[q] -= delta;
}

// the reference's order of the checks inside one transportmatrix call (src/matrixbuilding.jl:233, the generators' loops, :39, :61,
// :90, :114), behind the library's own input checks -- as oceantransportmatrixbuilder.jl_amd/dist.py ranks them
int status_rank(int32_t st) {
    static const int order[] = {OTMB_ERR_NONCANONICAL_INDICES, OTMB_ERR_HIP, OTMB_ERR_PUSH_MASK, OTMB_ERR_RHO_NAN, OTMB_ERR_FLUX_INTO_LAND,
                                OTMB_ERR_TADV_NAN, OTMB_ERR_TKH_NAN, OTMB_ERR_TKVML_NAN, OTMB_ERR_TKVDEEP_NAN, OTMB_ERR_CAPACITY, OTMB_ERR_GIVEN_FOREIGN};
    for (int q = 0; q < (int)(sizeof order / sizeof order[0]); ++q)
        if (order[q] == st) return q;
    return 99;
}

}  // namespace

struct otmb_mgpu {
    std::vector<Slab *> slabs;
    std::string err;
    int transport = 0;  // 0: every slab on one device (device-to-device copy); 1: RCCL send / recv; 2: hipMemcpyPeerAsync
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    // the grid the current partition was made for
    i64 nx = 0, ny = 0, nz = 0;
    std::vector<i64> bounds;
    bool reuse_grid = false, reuse_fluxes = false;  // otmb_mgpu_set_reuse
    int chain_pieces = 0;  // row bands the facefluxes chain is handed over in (otmb_mgpu_set_chain_pieces); 0 = by grid size
    // pending plan
    bool planned = false;
    otmb_tm_args args;  // host pointers of the plan
    i64 N = 0;
    i64 nnz[5] = {0, 0, 0, 0, 0};
};

namespace {

int32_t mg_fail(otmb_mgpu *mg, int32_t st, const std::string &detail = std::string()) {
    if (mg) {
        mg->err = otmb_status_string(st);
        if (!detail.empty()) mg->err += ": " + detail;
    }
    return st;
}

template <typename F>
void run_slabs(otmb_mgpu *mg, F &&fn) {  // one host thread per slab (the calling thread takes slab 0)
    const int n = (int)mg->slabs.size();
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);  // slab 0 switches the CALLER's thread to its device: put the caller's current device back afterwards
    std::vector<std::thread> th;
    for (int s = 1; s < n; ++s) th.emplace_back([&, s] { fn(s); });
    fn(0);
    for (auto &t : th) t.join();
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
}

int32_t reserve(Slab &sl, int b, size_t bytes, void **out) {
    const int32_t rc = otmb_reserve(sl.ctx, sl.buf[b], bytes ? bytes : 8);
    *out = sl.buf[b].p;
    return rc;
}

// first failing slab in the reference's order; copies its message
int32_t collect_status(otmb_mgpu *mg) {
    int best = -1;
    for (int s = 0; s < (int)mg->slabs.size(); ++s) {
        const Slab &sl = *mg->slabs[s];
        if (sl.status == OTMB_OK) continue;
        if (best < 0 || status_rank(sl.status) < status_rank(mg->slabs[best]->status)) best = s;
    }
    if (best < 0) return OTMB_OK;
    char where[64];
    snprintf(where, sizeof where, " [slab %d of %d, levels %lld-%lld]", best + 1, (int)mg->slabs.size(), (long long)mg->slabs[best]->k0 + 1,
             (long long)mg->slabs[best]->k1);
    mg->err = mg->slabs[best]->msg + (mg->slabs.size() > 1 ? where : "");
    return mg->slabs[best]->status;
}

int32_t set_partition(otmb_mgpu *mg, const std::vector<i64> &counts, i64 nx, i64 ny) {
    const int n = (int)mg->slabs.size();
    const i64 nz = (i64)counts.size();
    if (n > nz) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "more devices than levels");
    std::vector<i64> nb(n + 1, 0);
    int32_t rc = otmb_balanced_partition(counts.data(), nz, n, nb.data());
    if (rc) return mg_fail(mg, rc, "partition");
    if (nb != mg->bounds || nx != mg->nx || ny != mg->ny || nz != mg->nz)  // another grid or another cut: nothing staged is still valid
        for (Slab *sl : mg->slabs) {
            for (auto &k : sl->key) k = Slab::Key();
            sl->phi_resident = false;
        }
    mg->bounds = nb;
    std::vector<i64> cum(nz + 1, 0);
    for (i64 k = 0; k < nz; ++k) cum[k + 1] = cum[k] + counts[k];
    for (int s = 0; s < n; ++s) {
        Slab &sl = *mg->slabs[s];
        sl.k0 = mg->bounds[s]; sl.k1 = mg->bounds[s + 1];
        sl.ha = s > 0; sl.hb = s + 1 < n;
        sl.wet_base = cum[sl.k0];
        sl.n_own = cum[sl.k1] - cum[sl.k0];
        sl.status = OTMB_OK; sl.msg.clear();
    }
    mg->nx = nx; mg->ny = ny; mg->nz = nz;
    return OTMB_OK;
}

// Hand piece c of the plane -- elements [off, off + len) of src (device memory of slab `from`, produced on its stream) -- to slab
// `to`'s receive buffer and wake the thread of `to`.  Called by the thread of `from`.  No host wait anywhere: RCCL puts the receive on
// the consumer's stream (in front of the kernel its thread enqueues after being woken); the copy transports record an event behind the
// copy on the producer's stream, which the consumer's stream waits for (hipStreamWaitEvent) before its kernel of that piece.
int32_t send_piece(otmb_mgpu *mg, int from, int to, const double *src, i64 off, i64 len, int c) {
    Slab &a = *mg->slabs[from], &b = *mg->slabs[to];
    double *dst = (double *)b.buf[B_PLANE].p + off;
    src += off;
    int32_t rc = OTMB_OK;
    if (const char *e = getenv("OTMB_TEST_FAIL_PIECE")) {  // fault injection (tests): "<slab>:<piece>" fails that hand-off
        int fs = -1, fc = -1;
        if (sscanf(e, "%d:%d", &fs, &fc) == 2 && fs == from && fc == c) { a.msg = "plane hand-off failure injected (OTMB_TEST_FAIL_PIECE)"; rc = OTMB_ERR_HIP; }
    }
    if (rc != OTMB_OK) {
        // (nothing is sent: the consumer is released below with the failure)
    } else if (mg->transport == 1 && a.device != b.device) {
        // one grouped send / receive pair on the single-process communicator per piece: the send is ordered behind the producing
        // kernel on the producer's stream, the receive goes onto the CONSUMER's stream -- the piece crosses xGMI without a host round trip
        ncclResult_t r = mg->rccl.GroupStart();
        if (r == ncclSuccess) r = mg->rccl.Send(src, (size_t)len, ncclDouble, to, mg->comms[from], a.ctx->stream);
        if (r == ncclSuccess) r = mg->rccl.Recv(dst, (size_t)len, ncclDouble, from, mg->comms[to], b.ctx->stream);
        const ncclResult_t r2 = mg->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { a.msg = std::string("RCCL: ") + mg->rccl.GetErrorString(r); rc = OTMB_ERR_HIP; }
    } else {
        // same device (tests on a one-GPU box) or peer copy: behind the producing kernel on the producer's stream, then the event
        const hipError_t e = (a.device == b.device) ? hipMemcpyAsync(dst, src, (size_t)len * 8, hipMemcpyDeviceToDevice, a.ctx->stream)
                                                    : hipMemcpyPeerAsync(dst, b.device, src, a.device, (size_t)len * 8, a.ctx->stream);
        if (e != hipSuccess || hipEventRecord(a.piece_ev[c], a.ctx->stream) != hipSuccess) rc = OTMB_ERR_HIP;
    }
    {
        std::lock_guard<std::mutex> l(b.m);
        b.pieces_ready = c + 1;
        b.plane_failed |= rc != OTMB_OK;
    }
    b.cv.notify_all();
    return rc;
}

void fail_plane(otmb_mgpu *mg, int to) {  // a slab that cannot produce its plane must still release the slab above it
    Slab &b = *mg->slabs[to];
    {
        std::lock_guard<std::mutex> l(b.m);
        b.pieces_ready = 1 << 30;
        b.plane_failed = true;
    }
    b.cv.notify_all();
}

// One slab's share of a plan: stage its levels [k0 - ha, k1 + hb) of the host arrays on its device (what is resident under the reuse
// promises is not uploaded again), then count.  Leaves sl.nnz / sl.status.  before_upload / after_upload bracket the upload batch.
void plan_slab(otmb_mgpu *mg, int s, const otmb_tm_args *a, const std::function<void()> &before_upload, const std::function<void()> &after_upload,
               bool one_phase = false) {
    const i64 P = a->nx * a->ny;
    Slab &sl = *mg->slabs[s];
    // The residency keys are written when an upload is QUEUED: a slab that fails before its batch has reached the device (a later
    // reserve, the transfer, the index shift) must forget them, or a retry with reuse on would take arrays that were never copied
    // for resident and build matrices from uninitialised memory.
    bool on_device = false;
    auto fail = [&](int32_t st) {
        sl.status = st; sl.msg = otmb_last_error(sl.ctx);
        if (!on_device) {
            for (int b = 0; b < B_COUNT; ++b) sl.key[b] = Slab::Key();
            sl.phi_resident = false;
        }
    };
    if (hipSetDevice(sl.device) != hipSuccess) { otmb_fail(sl.ctx, OTMB_ERR_HIP, "hipSetDevice"); return fail(OTMB_ERR_HIP); }
    const i64 e0 = sl.k0 - sl.ha, e1 = sl.k1 + sl.hb, nze = e1 - e0;
    const size_t Ge = (size_t)(nze * P), off = (size_t)e0 * P;
    otmb_tm_args d = *a;
    d.nz = nze;
    d.n_wet = sl.n_own;
    d.push_mask = nullptr;
    std::vector<OtmbXferItem> up;
    int32_t r;
    void *p;
    // One staged array: uploaded unless the slot still holds this very host array (pointer, bytes, first level) and the caller has
    // promised not to have modified it (otmb_mgpu_set_reuse).  KIND 1: grid constant; 2: a face-flux array; 0: every call.
    auto put = [&](int B, const void *host, size_t bytes, int kind, void **out) -> int32_t {
        const void *before = sl.buf[B].p;
        int32_t rr = reserve(sl, B, bytes, out);
        if (rr) return rr;
        Slab::Key &k = sl.key[B];
        if (sl.buf[B].p != before) k = Slab::Key();  // the buffer was (re)allocated: whatever it held is gone
        const bool promised = (kind == 1 && mg->reuse_grid) || (kind == 2 && mg->reuse_fluxes);
        const bool resident = promised && k.host == host && k.bytes == bytes && k.e0 == e0 && bytes > 0;
        if (!resident && bytes) {
            up.push_back({*out, const_cast<void *>(host), bytes});
            sl.uploaded += (i64)bytes;
            if (kind == 1) otmb_ctx_forget_given(sl.ctx);  // (verdicts on given operators are keyed to addresses whose content has just changed)
        }
        k.host = (promised || resident) ? host : nullptr;
        k.bytes = bytes;
        k.e0 = e0;
        return resident ? -1 : OTMB_OK;  // (-1: nothing was queued)
    };
    // ϕ: what otmb_mgpu_facefluxes left on this device (already in the extended layout, halo fluxes filled in) is used when the
    // caller hands back the host arrays facefluxes wrote (reuse_fluxes); otherwise the levels [e0, e1) of the host arrays go up
    bool phi_here = mg->reuse_fluxes && sl.phi_resident && sl.buf[B_PHI0].cap >= Ge * 8;
    for (int f = 0; f < 6 && phi_here; ++f) phi_here = sl.phi_host[f] == (const void *)a->phi[f] && sl.buf[B_PHI0 + f].p;
    for (int f = 0; f < 6; ++f) {
        if (phi_here) {
            d.phi[f] = (const double *)sl.buf[B_PHI0 + f].p;
        } else {
            if ((r = put(B_PHI0 + f, (const char *)a->phi[f] + off * 8, Ge * 8, 0, &p)) > 0) return fail(r);
            d.phi[f] = (const double *)p;
        }
    }
    if (!phi_here) sl.phi_resident = false;
    if ((r = put(B_V, (const char *)a->v3d + off * 8, Ge * 8, 1, &p)) > 0) return fail(r);
    d.v3d = (const double *)p;
    if ((r = put(B_THK, (const char *)a->thkcello + off * 8, Ge * 8, 1, &p)) > 0) return fail(r);
    d.thkcello = (const double *)p;
    if (a->rho) {
        if ((r = put(B_RHO, (const char *)a->rho + off * 8, Ge * 8, 0, &p)) > 0) return fail(r);
        d.rho = (const double *)p;
    }
    if ((r = put(B_LW, (const char *)a->lwet3d + off * 8, Ge * 8, 1, &p)) > 0) return fail(r);
    d.lwet3d = (const int64_t *)p;
    const int32_t lw_rc = put(B_LWET, (const char *)((const i64 *)a->lwet + sl.wet_base), (size_t)sl.n_own * 8, 1, &p);
    if (lw_rc > 0) return fail(lw_rc);
    d.lwet = (const int64_t *)p;
    i64 *dlwet = (i64 *)p;
    const bool lwet_fresh = lw_rc == OTMB_OK;  // (a resident Lwet has been shifted to local indices already)
    for (int k = 0; k < 4; ++k) {
        if ((r = put(B_EDGE0 + k, a->edge_length[k], (size_t)P * 8, 1, &p)) > 0) return fail(r);
        d.edge_length[k] = (const double *)p;
        if ((r = put(B_DIST0 + k, a->dist_nbr[k], (size_t)P * 8, 1, &p)) > 0) return fail(r);
        d.dist_nbr[k] = (const double *)p;
    }
    if ((r = put(B_AREA, a->area2d, (size_t)P * 8, 1, &p)) > 0) return fail(r);
    d.area2d = (const double *)p;
    if ((r = put(B_ML, a->mlotst, (size_t)P * 8, 0, &p)) > 0) return fail(r);
    d.mlotst = (const double *)p;
    if ((r = put(B_ZT, a->zt + e0, (size_t)nze * 8, 1, &p)) > 0) return fail(r);
    d.zt = (const double *)p;
    // operators the caller passes (otmb_tm_args.given): this slab's columns -- colptr entries [wet_base, wet_base + n_own] (global numbering)
    // and the entries they delimit; grid constants of a time loop, so they fall under the reuse_grid promise
    for (int m = 1; m < 5; ++m) {
        const otmb_csc &g = a->given[m];
        if (!g.colptr) continue;
        const i64 lo = g.colptr[sl.wet_base] - 1, hi = g.colptr[sl.wet_base + sl.n_own] - 1;
        if (g.nnz < 0 || lo < 0 || hi < lo || hi > g.nnz || (g.nnz > 0 && (!g.rowval || !g.nzval))) {
            otmb_fail(sl.ctx, OTMB_ERR_INVALID_ARG, "given operator: colptr / nnz (is it N x N?)");
            return fail(OTMB_ERR_INVALID_ARG);
        }
        const int b0 = B_GIVEN0 + 3 * (m - 1);
        if ((r = put(b0, g.colptr + sl.wet_base, (size_t)(sl.n_own + 1) * 8, 1, &p)) > 0) return fail(r);
        d.given[m].colptr = (const int64_t *)p;
        if ((r = put(b0 + 1, g.rowval + lo, (size_t)(hi - lo) * 8, 1, &p)) > 0) return fail(r);
        d.given[m].rowval = (const int64_t *)p;
        if ((r = put(b0 + 2, g.nzval + lo, (size_t)(hi - lo) * 8, 1, &p)) > 0) return fail(r);
        d.given[m].nzval = (const double *)p;
        d.given[m].nnz = hi - lo;
    }
    if (before_upload) before_upload();  // (the pipelined build lets the slabs of one device take the link in turn)
    r = up.empty() ? OTMB_OK : otmb_xfer(sl.ctx, true, up.data(), (int)up.size());
    if (after_upload) after_upload();
    if (r) return fail(r);
    // Lwet of the owned cells as LOCAL linear indices of the extended grid (levels [e0, e1))
    if (lwet_fresh && sl.n_own > 0 && off > 0) {
        hipLaunchKernelGGL(shift_i64_kernel, dim3((unsigned)((sl.n_own + 255) / 256)), dim3(256), 0, sl.ctx->stream, dlwet, sl.n_own, (i64)off);
        if (hipGetLastError() != hipSuccess) { otmb_fail(sl.ctx, OTMB_ERR_HIP, "shift_i64_kernel"); return fail(OTMB_ERR_HIP); }
    }
    on_device = true;  // (errors from here on are the reference's own: the arrays ARE where the keys say)
    if ((r = otmb_transportmatrix_set_slab(sl.ctx, sl.wet_base))) return fail(r);
    if ((r = otmb_transportmatrix_plan_dev(sl.ctx, &d, sl.nnz))) return fail(r);
    // a given operator that is not what the library derives makes T a sparse add over whole columns of four matrices, sized by the sum of
    // their counts: with more than one slab, or in the one-phase build (whose T array is sized 7N), that is not built here -- the
    // two-phase call on one context does it (the host layers fall back to it)
    if (mg->slabs.size() > 1 || one_phase)
        for (int m = 1; m < 5; ++m)
            if (otmb_ctx_given_state(sl.ctx, m) == 2) { otmb_fail(sl.ctx, OTMB_ERR_GIVEN_FOREIGN); return fail(OTMB_ERR_GIVEN_FOREIGN); }
}

// matrices (bit m) a build with these arguments hands out: not T's operators when only T is wanted, never an operator the caller passed
unsigned wanted_mask(const otmb_tm_args &a) {
    unsigned w = (a.only_t ? 1u : 0x1fu) & ~((unsigned)a.skip_ops & 0x1fu);
    for (int m = 1; m < 5; ++m)
        if (a.given[m].colptr) w &= ~(1u << m);
    return w;
}

}  // namespace

extern "C" {

// Split levels 0..nz-1 into `nslabs` consecutive slabs with >= 1 level each, wet counts as even as a greedy sweep gets them
// (upper levels are wetter, so equal level counts would not balance).  bounds: nslabs + 1 entries, slab s = [bounds[s], bounds[s+1]).
// The ONE partition rule of the package: otmb_mgpu_* use it, and so does the process-per-GPU orchestration (dist.py).
int32_t otmb_balanced_partition(const int64_t *level_counts, int64_t nz, int32_t nslabs, int64_t *bounds) {
    if (!level_counts || !bounds || nz < 1 || nslabs < 1 || nslabs > nz) return OTMB_ERR_INVALID_ARG;
    std::vector<i64> cum(nz + 1, 0);
    for (i64 k = 0; k < nz; ++k) {
        if (level_counts[k] < 0) return OTMB_ERR_INVALID_ARG;
        cum[k + 1] = cum[k] + level_counts[k];
    }
    const i64 total = cum[nz];
    bounds[0] = 0;
    for (int r = 1; r < nslabs; ++r) {
        // first k with cum[k] >= total * r / nslabs, compared exactly: cum[k] * nslabs >= total * r
        const __int128 target = (__int128)total * r;
        i64 k = (i64)(std::lower_bound(cum.begin(), cum.end(), target, [&](i64 c, __int128 t) { return (__int128)c * nslabs < t; }) - cum.begin());
        if (k > nz) k = nz;
        // the closer of k-1 and k
        if (k > 0) {
            const __int128 dl = target - (__int128)cum[k - 1] * nslabs, dr = (__int128)cum[k < nz ? k : nz] * nslabs - target;
            if ((dl < 0 ? -dl : dl) <= (dr < 0 ? -dr : dr)) k -= 1;
        }
        if (k < bounds[r - 1] + 1) k = bounds[r - 1] + 1;  // >= 1 level per slab on both sides
        if (k > nz - (nslabs - r)) k = nz - (nslabs - r);
        bounds[r] = k;
    }
    bounds[nslabs] = nz;
    return OTMB_OK;
}

int32_t otmb_mgpu_create(int32_t ndev, const int32_t *device_ids, otmb_mgpu **out) {
    if (!out) return OTMB_ERR_INVALID_ARG;
    *out = nullptr;
    if (ndev < 1 || ndev > 64 || !device_ids) return OTMB_ERR_INVALID_ARG;
    otmb_mgpu *mg = new otmb_mgpu();
    bool distinct = true, same = true;
    for (int s = 0; s < ndev; ++s) {
        for (int q = 0; q < s; ++q) distinct &= device_ids[q] != device_ids[s];
        same &= device_ids[s] == device_ids[0];
    }
    for (int s = 0; s < ndev; ++s) {
        Slab *sl = new Slab();
        sl->device = device_ids[s];
        mg->slabs.push_back(sl);
        const int32_t rc = otmb_ctx_create(device_ids[s], &sl->ctx);
        if (rc) { otmb_mgpu_destroy(mg); return rc; }
        // every slab has its own transfer engine (pinned ring + host copy threads); together they get the machine's cores, not eight each
        const int hw = (int)std::thread::hardware_concurrency();
        sl->ctx->xfer_threads = std::max(1, std::min(8, (hw > 0 ? hw : 8) / ndev));
    }
    const char *tenv = getenv("OTMB_MGPU_TRANSPORT");
    if (same && ndev == 1 && tenv && std::string(tenv) == "rccl") {
        // (tests on a one-GPU box: load librccl.so and bring up a one-rank communicator, so that at least the loading and the
        // initialisation of the RCCL transport have run somewhere; a single slab never hands a plane over)
        std::string why;
        if (mg->rccl.load(why)) {
            mg->comms.assign(1, nullptr);
            if (mg->rccl.CommInitAll(mg->comms.data(), 1, device_ids) == ncclSuccess) mg->transport = 1;
            else mg->comms.clear();
        }
        if (mg->transport != 1) { mg->err = "RCCL transport not available: " + (why.empty() ? std::string("ncclCommInitAll failed") : why); }
    } else if (same) {
        mg->transport = 0;
    } else if (!distinct) {
        otmb_mgpu_destroy(mg);  // a device twice among others: no communicator has such a shape
        return OTMB_ERR_INVALID_ARG;
    } else {
        // different devices: the single-process RCCL communicator (one rank per slab); peer copies only if RCCL cannot be loaded
        // or is refused by OTMB_MGPU_TRANSPORT=peer
        const char *e = getenv("OTMB_MGPU_TRANSPORT");
        const bool want_peer = e && std::string(e) == "peer";
        mg->transport = 2;
        std::string why;
        if (!want_peer && mg->rccl.load(why)) {
            mg->comms.assign(ndev, nullptr);
            if (mg->rccl.CommInitAll(mg->comms.data(), ndev, device_ids) == ncclSuccess) mg->transport = 1;
            else { mg->comms.clear(); why = "ncclCommInitAll failed"; }
        }
        if (mg->transport == 2) {
            // which transport runs and why is the caller's to see (otmb_mgpu_last_error right after create, otmb_mgpu_transport; bench.py prints both)
            mg->err = want_peer ? "peer copies (OTMB_MGPU_TRANSPORT=peer)" : "peer copies, RCCL not used: " + why;
            for (int s = 0; s + 1 < ndev; ++s) {  // neighbours hand planes upwards: enable access both ways
                for (int dir = 0; dir < 2; ++dir) {
                    const int from = device_ids[s + dir], to = device_ids[s + 1 - dir];
                    (void)hipSetDevice(from);
                    const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                        char msg[128];
                        snprintf(msg, sizeof msg, "; hipDeviceEnablePeerAccess(%d -> %d): %s", from, to, hipGetErrorString(e));
                        mg->err += msg;
                    }
                    (void)hipGetLastError();
                }
            }
        }
    }
    *out = mg;
    return OTMB_OK;
}

void otmb_mgpu_destroy(otmb_mgpu *mg) {
    if (!mg) return;
    for (Slab *sl : mg->slabs) {
        if (sl->ctx) {
            (void)hipSetDevice(sl->device);
            (void)hipStreamSynchronize(sl->ctx->stream);
        }
    }
    for (ncclComm_t c : mg->comms)
        if (c) (void)mg->rccl.CommDestroy(c);
    for (Slab *sl : mg->slabs) {
        if (sl->ctx) {
            (void)hipSetDevice(sl->device);
            for (DevBuf &b : sl->buf)
                if (b.p) (void)hipFree(b.p);
            for (hipEvent_t e : sl->piece_ev) (void)hipEventDestroy(e);
            otmb_ctx_destroy(sl->ctx);
        }
        delete sl;
    }
    delete mg;
}

const char *otmb_mgpu_last_error(const otmb_mgpu *mg) { return mg ? mg->err.c_str() : "null otmb_mgpu"; }
int32_t otmb_mgpu_ndev(const otmb_mgpu *mg) { return mg ? (int32_t)mg->slabs.size() : 0; }
int32_t otmb_mgpu_transport(const otmb_mgpu *mg) { return mg ? mg->transport : -1; }

// The two promises of otmb_ctx_set_reuse_grid / _set_reuse_fluxes for the slabs (independent; both off by default): grid-constant host
// arrays that are the very arrays of the previous plan are not uploaded again; ϕ that otmb_mgpu_facefluxes computed and copied to
// these very host arrays is used where it is, on the devices.
int32_t otmb_mgpu_set_reuse(otmb_mgpu *mg, int32_t grid, int32_t fluxes) {
    if (!mg) return OTMB_ERR_INVALID_ARG;
    mg->reuse_grid = grid != 0;
    mg->reuse_fluxes = fluxes != 0;
    for (Slab *sl : mg->slabs) {
        if (!mg->reuse_grid)
            for (int b = B_V; b < B_UMO; ++b) sl->key[b] = Slab::Key();
    }
    return OTMB_OK;
}
// Speed only: in how many row bands the facefluxes chain hands its plane from slab to slab (0 = by grid size: 4 on grids of half a
// million columns and more, else 1).  Any number gives the same six arrays.
int32_t otmb_mgpu_set_chain_pieces(otmb_mgpu *mg, int32_t pieces) {
    if (!mg || pieces < 0) return OTMB_ERR_INVALID_ARG;
    mg->chain_pieces = pieces;
    return OTMB_OK;
}
// bytes copied host -> device over all slabs since the object was created (diagnostics / tests)
int64_t otmb_mgpu_uploaded_bytes(const otmb_mgpu *mg) {
    if (!mg) return -1;
    i64 t = 0;
    for (const Slab *sl : mg->slabs) t += sl->uploaded;
    return t;
}

// the partition of the last facefluxes / plan: ndev + 1 level bounds
int32_t otmb_mgpu_partition(const otmb_mgpu *mg, int64_t *bounds) {
    if (!mg || !bounds || mg->bounds.empty()) return OTMB_ERR_INVALID_ARG;
    for (size_t q = 0; q < mg->bounds.size(); ++q) bounds[q] = mg->bounds[q];
    return OTMB_OK;
}

// facefluxes over the slabs: same arguments and results as otmb_facefluxes (src/velocities.jl:118-130, :154-255).
int32_t otmb_mgpu_facefluxes(otmb_mgpu *mg, const void *umo, const void *vmo, int32_t src_is_f32, const uint8_t *wet3d, double fill,
                             int64_t nx, int64_t ny, int64_t nz, int32_t topology, double *const phi[6]) {
    if (!mg || !umo || !vmo || !wet3d || !phi) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null argument");
    for (int f = 0; f < 6; ++f)
        if (!phi[f]) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null output");
    if (nx < 1 || ny < 1 || nz < 1) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "grid size");
    const i64 P = nx * ny;
    const size_t es = src_is_f32 ? 4 : 8;
    const int n = (int)mg->slabs.size();
    mg->planned = false;
    // wet cells per level (bytes of indices.wet3D): what the partition balances
    std::vector<i64> counts(nz, 0);
    {
        OtmbThreadPool pool(std::min<int>(8, std::max<int>(1, (int)std::thread::hardware_concurrency())));
        pool.parallel_for((int)nz, [&](int k) {
            const uint8_t *w = wet3d + (size_t)k * P;
            i64 c = 0;
            for (i64 q = 0; q < P; ++q) c += w[q] != 0;
            counts[k] = c;
        });
    }
    int32_t rc;
    if ((rc = set_partition(mg, counts, nx, ny))) return rc;
    for (Slab *sl : mg->slabs) { sl->pieces_ready = 0; sl->plane_failed = false; }
    // The chain in C row bands: slab s piece c needs only piece c of the plane of slab s + 1, so the slabs of ONE field
    // overlap: critical path t_ff / W x (1 + (W - 1) / C) instead of t_ff.  Whole rows per piece: the south-row dependence of a cell is on
    // INPUTS of its own slab (src/velocities.jl:219-224), never on another piece's results.
    int C = mg->chain_pieces > 0 ? mg->chain_pieces : ((P >= (1ll << 19) && n > 1) ? 4 : 1);
    if (C > ny) C = (int)ny;
    std::vector<i64> jb(C + 1);
    for (int c = 0; c <= C; ++c) jb[c] = ny * c / C;
    // The plane buffer of slab s is written by the thread of slab s + 1 (send_plane): every slab's buffer is (re)allocated HERE, before any
    // slab thread runs -- reserved inside its owner's thread it could still be missing, or be freed for a larger one (an otmb_mgpu reused
    // on a larger grid), while the slab below was already copying into it (an intermittent "plane hand-off" error in
    // tests/test_mgpu.py::test_mgpu_reports_the_reference_error_of_the_failing_slab, round 4 call 42).
    {
        int dev0 = 0;
        (void)hipGetDevice(&dev0);
        for (Slab *sl : mg->slabs) {
            void *q;
            if (hipSetDevice(sl->device) != hipSuccess) { (void)hipSetDevice(dev0); return mg_fail(mg, OTMB_ERR_HIP, "hipSetDevice"); }
            if ((rc = reserve(*sl, B_PLANE, (size_t)P * 8, &q))) { (void)hipSetDevice(dev0); return mg_fail(mg, rc, "plane buffer"); }
            while ((int)sl->piece_ev.size() < C) {  // (created on the producer's device, before any slab thread runs)
                hipEvent_t e;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipSetDevice(dev0); return mg_fail(mg, OTMB_ERR_HIP, "hipEventCreate"); }
                sl->piece_ev.push_back(e);
            }
        }
        (void)hipSetDevice(dev0);
    }
    // Slabs that share a device take its link in turn: uploads deepest slab first (the chain's order), one slab downloading at a time -- the
    // deepest slab's six arrays go home while the slab above it uploads (the link is full duplex; section 9 of profiles/r05/README.md).
    std::mutex turn_m;
    std::condition_variable turn_cv;
    std::vector<char> ff_uploaded(n, 0);
    std::mutex ff_down_turn[16];
    run_slabs(mg, [&](int s) {
        Slab &sl = *mg->slabs[s];
        int succ = -1;  // the nearest slab BELOW me on my device: I upload when it has
        for (int q = s + 1; q < n && succ < 0; ++q)
            if (mg->slabs[q]->device == sl.device) succ = q;
        bool left = false;
        auto leave_link = [&] {
            if (left) return;
            left = true;
            { std::lock_guard<std::mutex> l(turn_m); ff_uploaded[s] = 1; }
            turn_cv.notify_all();
        };
        struct Leave { std::function<void()> f; ~Leave() { f(); } } leave_guard{leave_link};  // (whatever path this slab returns by)
        auto fail = [&](int32_t st) {
            sl.status = st;
            sl.msg = otmb_last_error(sl.ctx);
            if (s > 0) fail_plane(mg, s - 1);
        };
        if (hipSetDevice(sl.device) != hipSuccess) { otmb_fail(sl.ctx, OTMB_ERR_HIP, "hipSetDevice"); return fail(OTMB_ERR_HIP); }
        const i64 nl = sl.k1 - sl.k0;
        const size_t Gl = (size_t)(nl * P), Ge = (size_t)((nl + sl.ha + sl.hb) * P);
        void *du, *dv, *dw, *dplane, *dphi[6];
        int32_t r;
        sl.phi_resident = false;
        if ((r = reserve(sl, B_UMO, Gl * es, &du)) || (r = reserve(sl, B_VMO, Gl * es, &dv)) || (r = reserve(sl, B_WET, Gl, &dw)) ||
            (r = reserve(sl, B_PLANE, (size_t)P * 8, &dplane)))
            return fail(r);
        // the six ϕ arrays are kept in the layout otmb_mgpu_transportmatrix_plan wants (levels [k0 - ha, k1 + hb)): the owned levels are
        // computed in place behind the halo plane above; with otmb_mgpu_set_reuse(…, fluxes) the plan then uploads no ϕ at all
        for (int f = 0; f < 6; ++f) {
            void *q;
            if ((r = reserve(sl, B_PHI0 + f, Ge * 8, &q))) return fail(r);
            sl.key[B_PHI0 + f] = Slab::Key();
            if (sl.ha && hipMemsetAsync(q, 0, (size_t)P * 8, sl.ctx->stream) != hipSuccess) return fail(OTMB_ERR_HIP);
            if (sl.hb && hipMemsetAsync((char *)q + (Ge - (size_t)P) * 8, 0, (size_t)P * 8, sl.ctx->stream) != hipSuccess) return fail(OTMB_ERR_HIP);
            dphi[f] = (char *)q + (size_t)sl.ha * P * 8;
        }
        OtmbXferItem up[3] = {{du, (char *)umo + (size_t)sl.k0 * P * es, Gl * es}, {dv, (char *)vmo + (size_t)sl.k0 * P * es, Gl * es},
                              {dw, (char *)wet3d + (size_t)sl.k0 * P, Gl}};
        if (succ >= 0) {
            std::unique_lock<std::mutex> l(turn_m);
            turn_cv.wait(l, [&] { return ff_uploaded[succ] != 0; });
        }
        r = otmb_xfer(sl.ctx, true, up, 3);
        leave_link();
        if (r) return fail(r);
        sl.uploaded += (i64)(2 * Gl * es + Gl);
        double *dp[6];
        for (int f = 0; f < 6; ++f) dp[f] = (double *)dphi[f];
        const bool by_event = !(mg->transport == 1 && s + 1 < n && mg->slabs[s + 1]->device != sl.device);
        for (int c = 0; c < C; ++c) {
            if (sl.hb) {  // the chain: piece c of ϕtop of the level below this slab
                std::unique_lock<std::mutex> l(sl.m);
                sl.cv.wait(l, [&] { return sl.pieces_ready > c; });
                if (sl.plane_failed) {  // the slab below failed: its status is the one reported; release the slab above
                    l.unlock();
                    if (s > 0) fail_plane(mg, s - 1);
                    return;
                }
                l.unlock();
                // (copy transports: this slab's stream waits for the copy of piece c -- an event, no host wait)
                if (by_event && hipStreamWaitEvent(sl.ctx->stream, mg->slabs[s + 1]->piece_ev[c], 0) != hipSuccess) {
                    otmb_fail(sl.ctx, OTMB_ERR_HIP, "hipStreamWaitEvent");
                    return fail(OTMB_ERR_HIP);
                }
            }
            if ((r = otmb_facefluxes_rows_dev(sl.ctx, du, dv, src_is_f32, (const uint8_t *)dw, fill, nx, ny, nl, topology, dp,
                                              sl.hb ? (const double *)dplane : nullptr, nullptr, jb[c], jb[c + 1], c == 0)))
                return fail(r);
            if (s > 0 && (r = send_piece(mg, s, s - 1, dp[OTMB_TOP], jb[c] * nx, (jb[c + 1] - jb[c]) * nx, c))) {  // ϕtop of this slab's first level goes up
                otmb_fail(sl.ctx, r, sl.msg.empty() ? "plane hand-off" : sl.msg.c_str());
                sl.status = r;
                sl.msg = otmb_last_error(sl.ctx);
                fail_plane(mg, s - 1);
                return;
            }
        }
        // the halo levels act as neighbours only; the one flux each of them pushes into an owned cell: the halo above pushes its ϕbottom =
        // this slab's first ϕtop (src/velocities.jl:240), the halo below its ϕtop = the plane received from the slab below
        bool halo_ok = true;
        if (sl.ha) halo_ok &= hipMemcpyAsync((char *)dp[OTMB_BOTTOM] - (size_t)P * 8, dp[OTMB_TOP], (size_t)P * 8, hipMemcpyDeviceToDevice, sl.ctx->stream) == hipSuccess;
        if (sl.hb) halo_ok &= hipMemcpyAsync((char *)dp[OTMB_TOP] + Gl * 8, dplane, (size_t)P * 8, hipMemcpyDeviceToDevice, sl.ctx->stream) == hipSuccess;
        std::vector<OtmbXferItem> down;
        for (int f = 0; f < 6; ++f) down.push_back({dp[f], phi[f] + (size_t)sl.k0 * P, Gl * 8});
        {
            std::unique_lock<std::mutex> turn(ff_down_turn[sl.device % 16]);
            const std::function<void()> hand_on = [&] { if (turn.owns_lock()) turn.unlock(); };
            if ((r = otmb_xfer(sl.ctx, false, down.data(), (int)down.size(), &hand_on))) { sl.status = r; sl.msg = otmb_last_error(sl.ctx); return; }
        }
        if ((r = otmb_facefluxes_slab_flags(sl.ctx, &sl.u_valid, &sl.v_valid))) { sl.status = r; sl.msg = otmb_last_error(sl.ctx); return; }
        if (halo_ok) {
            sl.phi_resident = true;
            for (int f = 0; f < 6; ++f) sl.phi_host[f] = phi[f];
        }
    });
    if ((rc = collect_status(mg))) return rc;
    bool u = false, v = false;
    for (Slab *sl : mg->slabs) { u |= sl->u_valid != 0; v |= sl->v_valid != 0; }
    if (!u || !v) return mg_fail(mg, OTMB_ERR_ALL_MISSING);  // @assert over the WHOLE grid (:199-200)
    (void)n;
    return OTMB_OK;
}

// transportmatrix over the slabs, two-phase like otmb_transportmatrix_plan / _fetch: HOST pointers of the whole grid.
int32_t otmb_mgpu_transportmatrix_plan(otmb_mgpu *mg, const otmb_tm_args *a, int64_t nnz[5]) {
    if (!mg || !a || !nnz) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null argument");
    if (a->nx < 1 || a->ny < 1 || a->nz < 1) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "grid size");
    for (int f = 0; f < 6; ++f)
        if (!a->phi[f]) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "phi");
    for (int d = 0; d < 4; ++d)
        if (!a->edge_length[d] || !a->dist_nbr[d]) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "metrics");
    if (!a->v3d || !a->thkcello || !a->lwet3d || !a->area2d || !a->zt || !a->mlotst) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null input array");
    const i64 P = a->nx * a->ny, G = P * a->nz, N = a->n_wet;
    if (N < 0 || N > G || (N > 0 && !a->lwet)) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "lwet / n_wet");
    mg->planned = false;
    // wet cells per level from Lwet (ascending 1-based linear indices, k slowest): positions of the level boundaries
    std::vector<i64> counts(a->nz, 0);
    {
        const i64 *lw = (const i64 *)a->lwet;
        i64 prev = 0;
        for (i64 k = 0; k < a->nz; ++k) {
            const i64 pos = (i64)(std::upper_bound(lw, lw + N, (k + 1) * P) - lw);  // entries with L <= (k+1) P
            counts[k] = pos - prev;
            prev = pos;
        }
        if (prev != N) return mg_fail(mg, OTMB_ERR_NONCANONICAL_INDICES);
    }
    int32_t rc;
    if ((rc = set_partition(mg, counts, a->nx, a->ny))) return rc;
    mg->args = *a;
    mg->N = N;
    run_slabs(mg, [&](int s) { plan_slab(mg, s, a, nullptr, nullptr); });
    if ((rc = collect_status(mg))) return rc;
    for (int m = 0; m < 5; ++m) {
        i64 run = 0;
        for (Slab *sl : mg->slabs) { sl->base[m] = run; run += sl->nnz[m]; }
        nnz[m] = mg->nnz[m] = run;
    }
    mg->planned = true;
    return OTMB_OK;
}

int32_t otmb_mgpu_transportmatrix_fetch(otmb_mgpu *mg, int64_t *const colptr[5], int64_t *const rowval[5], double *const nzval[5],
                                        int64_t nnz_out[5]) {
    if (!mg || !colptr || !rowval || !nzval || !nnz_out) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null argument");
    if (!mg->planned) return mg_fail(mg, OTMB_ERR_NO_PLAN);
    const unsigned want = wanted_mask(mg->args);
    const int n = (int)mg->slabs.size();
    for (int m = 0; m < 5; ++m)
        if (((want >> m) & 1u) && (!colptr[m] || (mg->nnz[m] > 0 && (!rowval[m] || !nzval[m])))) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null output");
    mg->planned = false;  // a plan is consumed by its fetch (not by a call that is refused for its arguments)
    // phase 1: every slab fills its column range on its device and copies the four operators home (their counts are exact
    // since the plan); T's final count is only known now (entries that summed to exactly zero, src/matrixbuilding.jl:147)
    run_slabs(mg, [&](int s) {
        Slab &sl = *mg->slabs[s];
        auto fail = [&](int32_t st) { sl.status = st; sl.msg = otmb_last_error(sl.ctx); };
        if (hipSetDevice(sl.device) != hipSuccess) { otmb_fail(sl.ctx, OTMB_ERR_HIP, "hipSetDevice"); return fail(OTMB_ERR_HIP); }
        int32_t r;
        int64_t *dcp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}, *drv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        double *dnz[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        void *p;
        for (int m = 0; m < 5; ++m) {
            if (!((want >> m) & 1u)) continue;
            if ((r = reserve(sl, B_COLPTR0 + m, (size_t)(sl.n_own + 1) * 8, &p))) return fail(r);
            dcp[m] = (int64_t *)p;
            if ((r = reserve(sl, B_ROWVAL0 + m, (size_t)sl.nnz[m] * 8, &p))) return fail(r);
            drv[m] = (int64_t *)p;
            if ((r = reserve(sl, B_NZVAL0 + m, (size_t)sl.nnz[m] * 8, &p))) return fail(r);
            dnz[m] = (double *)p;
        }
        if ((r = otmb_transportmatrix_set_nnz_base(sl.ctx, sl.base))) return fail(r);
        if ((r = otmb_transportmatrix_fill_dev(sl.ctx, dcp, drv, dnz))) return fail(r);
        if ((r = otmb_transportmatrix_nnz(sl.ctx, sl.nnz))) return fail(r);
        std::vector<OtmbXferItem> down;
        const bool last = s + 1 == n;
        const int small_rows = mg->N < ((i64)1 << 31);  // (otmb_xfer.h: `narrow`)
        for (int m = 1; m < 5; ++m) {
            if (!((want >> m) & 1u)) continue;
            down.push_back({dcp[m], colptr[m] + sl.wet_base, (size_t)(sl.n_own + (last ? 1 : 0)) * 8, 2});  // (<= 7 entries per column: one byte each)
            if (sl.nnz[m] > 0) {
                down.push_back({drv[m], rowval[m] + sl.base[m], (size_t)sl.nnz[m] * 8, small_rows});
                down.push_back({dnz[m], nzval[m] + sl.base[m], (size_t)sl.nnz[m] * 8});
            }
        }
        if ((r = otmb_xfer(sl.ctx, false, down.data(), (int)down.size()))) return fail(r);
    });
    int32_t rc;
    if ((rc = collect_status(mg))) return rc;
    // phase 2: T.  A slab whose T lost entries shifts the offsets of every slab below it.
    std::vector<i64> tbase(n, 0);
    i64 run = 0;
    for (int s = 0; s < n; ++s) { tbase[s] = run; run += mg->slabs[s]->nnz[0]; }
    run_slabs(mg, [&](int s) {
        Slab &sl = *mg->slabs[s];
        if (hipSetDevice(sl.device) != hipSuccess) { sl.status = OTMB_ERR_HIP; sl.msg = "hipSetDevice"; return; }
        const bool last = s + 1 == n;
        std::vector<OtmbXferItem> down;
        // (T of a foreign build -- one slab only -- comes out of the sparse adds: no bound on a column's entries, Int32 offsets where they fit)
        down.push_back({sl.buf[B_COLPTR0].p, colptr[0] + sl.wet_base, (size_t)(sl.n_own + (last ? 1 : 0)) * 8,
                        otmb_tm_plan_foreign(sl.ctx) ? (mg->nnz[0] + 1 < ((i64)1 << 31) ? 1 : 0) : 2});
        if (sl.nnz[0] > 0) {
            down.push_back({sl.buf[B_ROWVAL0].p, rowval[0] + tbase[s], (size_t)sl.nnz[0] * 8, mg->N < ((i64)1 << 31)});
            down.push_back({sl.buf[B_NZVAL0].p, nzval[0] + tbase[s], (size_t)sl.nnz[0] * 8});
        }
        const int32_t r = otmb_xfer(sl.ctx, false, down.data(), (int)down.size());
        if (r) { sl.status = r; sl.msg = otmb_last_error(sl.ctx); return; }
        const i64 delta = sl.base[0] - tbase[s];  // entries that the slabs above reserved and did not store
        if (delta > 0) {
            i64 *cp = colptr[0] + sl.wet_base;
            const i64 cnt = sl.n_own + (last ? 1 : 0);
            for (i64 q = 0; q < cnt; ++q) cp[q] -= delta;
        }
    });
    if ((rc = collect_status(mg))) return rc;
    for (int m = 0; m < 5; ++m) {
        i64 tot = 0;
        for (Slab *sl : mg->slabs) tot += sl->nnz[m];
        nnz_out[m] = mg->nnz[m] = ((want >> m) & 1u) ? tot : 0;
    }
    return OTMB_OK;
}

// One-phase transportmatrix over the slabs, pipelined: HOST pointers of the whole grid in, the five matrices out into arrays of
// `capacity` entries (7N, 7N, 5N, 3N, 3N always suffice: src/matrixbuilding.jl:244-296, :348-415, :450-477), no nnz round trip to the
// caller in between.  Slab s uploads its levels while slab s - 1 counts, fills and copies its columns home: the link carries both
// directions at once (tools/micro/pcie_duplex.py: 57 GB/s one way, 2 x 49 GB/s both ways on this pool's boxes), which the two-phase
// protocol cannot use -- every upload there precedes the count, every download follows it.  Slabs that share a device take the link
// in turn (slab order) for their uploads; a slab's column offsets are the running sums of the FINAL counts of the slabs above it
// (T's entries that cancelled exactly included), so nothing is re-based afterwards.  Same matrices bit for bit.
int32_t otmb_mgpu_transportmatrix_onepass(otmb_mgpu *mg, const otmb_tm_args *a, int64_t *const colptr[5], int64_t *const rowval[5],
                                          double *const nzval[5], const int64_t capacity[5], int64_t nnz_out[5]) {
    if (!mg || !a || !colptr || !rowval || !nzval || !capacity || !nnz_out) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null argument");
    if (a->nx < 1 || a->ny < 1 || a->nz < 1) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "grid size");
    for (int f = 0; f < 6; ++f)
        if (!a->phi[f]) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "phi");
    for (int d = 0; d < 4; ++d)
        if (!a->edge_length[d] || !a->dist_nbr[d]) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "metrics");
    if (!a->v3d || !a->thkcello || !a->lwet3d || !a->area2d || !a->zt || !a->mlotst) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null input array");
    const i64 P = a->nx * a->ny, G = P * a->nz, N = a->n_wet;
    if (N < 0 || N > G || (N > 0 && !a->lwet)) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "lwet / n_wet");
    const unsigned want = wanted_mask(*a);
    const int n = (int)mg->slabs.size();
    for (int m = 0; m < 5; ++m)
        if (((want >> m) & 1u) && (!colptr[m] || capacity[m] < 0 || (capacity[m] > 0 && (!rowval[m] || !nzval[m])))) return mg_fail(mg, OTMB_ERR_INVALID_ARG, "null output");
    mg->planned = false;
    std::vector<i64> counts(a->nz, 0);
    {
        const i64 *lw = (const i64 *)a->lwet;
        i64 prev = 0;
        for (i64 k = 0; k < a->nz; ++k) {
            const i64 pos = (i64)(std::upper_bound(lw, lw + N, (k + 1) * P) - lw);
            counts[k] = pos - prev;
            prev = pos;
        }
        if (prev != N) return mg_fail(mg, OTMB_ERR_NONCANONICAL_INDICES);
    }
    int32_t rc;
    if ((rc = set_partition(mg, counts, a->nx, a->ny))) return rc;
    mg->args = *a;
    mg->N = N;
    // the pipeline's state: which slabs have left the link (uploads), which have published their final counts
    std::mutex m;
    std::condition_variable cv;
    std::vector<char> uploaded(n, 0);
    int published = 0;
    bool broken = false;  // a slab above failed: nobody below it stores anything
    i64 next_base[5] = {0, 0, 0, 0, 0};
    // OTMB_ONEPASS_TRACE=1: every slab's phase stamps (ms since the call began) on stderr -- where a time slice goes
    const bool trace = getenv("OTMB_ONEPASS_TRACE") && getenv("OTMB_ONEPASS_TRACE")[0] == '1';
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::vector<std::array<double, 6>> stamps(n, std::array<double, 6>{{0, 0, 0, 0, 0, 0}});
    std::mutex down_turn[16];
    const bool down_turns = !(getenv("OTMB_ONEPASS_DOWN_TURNS") && getenv("OTMB_ONEPASS_DOWN_TURNS")[0] == '0');
    run_slabs(mg, [&](int s) {
        Slab &sl = *mg->slabs[s];
        int pred = -1;  // the nearest slab above me on MY device: I upload when it has
        for (int q = s - 1; q >= 0 && pred < 0; --q)
            if (mg->slabs[q]->device == sl.device) pred = q;
        bool left_link = false;
        auto leave_link = [&] {
            if (left_link) return;
            left_link = true;
            { std::lock_guard<std::mutex> l(m); uploaded[s] = 1; }
            cv.notify_all();
        };
        bool have_published = false;
        auto publish = [&](bool ok) {  // my final counts are known (or never will be): release the slab below me
            if (have_published) return;
            have_published = true;
            {
                std::lock_guard<std::mutex> l(m);
                if (ok) for (int q = 0; q < 5; ++q) next_base[q] += sl.nnz[q];
                else broken = true;
                published = s + 1;
            }
            cv.notify_all();
        };
        auto fail = [&](int32_t st) { sl.status = st; sl.msg = otmb_last_error(sl.ctx); };
        plan_slab(mg, s, a,
                  [&] {
                      if (pred >= 0) {
                          std::unique_lock<std::mutex> l(m);
                          cv.wait(l, [&] { return uploaded[pred] != 0; });
                      }
                      stamps[s][0] = now_ms();
                  },
                  [&] { stamps[s][1] = now_ms(); leave_link(); }, true);
        leave_link();  // (a slab that failed before its upload must not hold the link)
        stamps[s][2] = now_ms();
        {   // my turn to place my columns: every slab above me has published
            std::unique_lock<std::mutex> l(m);
            cv.wait(l, [&] { return published >= s; });
            if (broken || sl.status != OTMB_OK) {
                l.unlock();
                return publish(false);
            }
            for (int q = 0; q < 5; ++q) sl.base[q] = next_base[q];
        }
        stamps[s][3] = now_ms();
        if (hipSetDevice(sl.device) != hipSuccess) { otmb_fail(sl.ctx, OTMB_ERR_HIP, "hipSetDevice"); fail(OTMB_ERR_HIP); return publish(false); }
        for (int q = 0; q < 5; ++q)
            if (((want >> q) & 1u) && sl.base[q] + sl.nnz[q] > capacity[q]) {
                otmb_fail(sl.ctx, OTMB_ERR_CAPACITY);
                fail(OTMB_ERR_CAPACITY);
                return publish(false);
            }
        int32_t r;
        int64_t *dcp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}, *drv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        double *dnz[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        void *p;
        for (int q = 0; q < 5; ++q) {
            if (!((want >> q) & 1u)) continue;
            if ((r = reserve(sl, B_COLPTR0 + q, (size_t)(sl.n_own + 1) * 8, &p))) { fail(r); return publish(false); }
            dcp[q] = (int64_t *)p;
            if ((r = reserve(sl, B_ROWVAL0 + q, (size_t)sl.nnz[q] * 8, &p))) { fail(r); return publish(false); }
            drv[q] = (int64_t *)p;
            if ((r = reserve(sl, B_NZVAL0 + q, (size_t)sl.nnz[q] * 8, &p))) { fail(r); return publish(false); }
            dnz[q] = (double *)p;
        }
        if ((r = otmb_transportmatrix_set_nnz_base(sl.ctx, sl.base)) || (r = otmb_transportmatrix_fill_dev(sl.ctx, dcp, drv, dnz)) ||
            (r = otmb_transportmatrix_nnz(sl.ctx, sl.nnz))) {
            fail(r);
            return publish(false);
        }
        publish(true);  // (T's final count included: the slab below starts exactly where my entries end)
        stamps[s][4] = now_ms();
        std::vector<OtmbXferItem> down;
        const bool last = s + 1 == n;
        // row indices (<= N) cross the link as Int32 where they provably fit, column offsets as one byte per column (otmb_xfer.h: `narrow`)
        const int small_rows = N < ((i64)1 << 31);
        for (int q = 0; q < 5; ++q) {
            if (!((want >> q) & 1u)) continue;
            down.push_back({dcp[q], colptr[q] + sl.wet_base, (size_t)(sl.n_own + (last ? 1 : 0)) * 8, 2});
            if (sl.nnz[q] > 0) {
                down.push_back({drv[q], rowval[q] + sl.base[q], (size_t)sl.nnz[q] * 8, small_rows});
                down.push_back({dnz[q], nzval[q] + sl.base[q], (size_t)sl.nnz[q] * 8});
            }
        }
        {   // One slab of a device copies home at a time: two download streams side by side slowed the NEXT slab's upload to 22 GB/s
            // (profiles/r05/README.md section 9).  The turn ends with the slab's last DMA; its host threads unpack the ring afterwards.
            // (experiment: OTMB_ONEPASS_DOWN_TURNS=0 lets the downloads overlap)
            std::unique_lock<std::mutex> turn(down_turn[sl.device % 16], std::defer_lock);
            if (down_turns) turn.lock();
            const std::function<void()> hand_on = [&] { if (turn.owns_lock()) turn.unlock(); };
            if ((r = otmb_xfer(sl.ctx, false, down.data(), (int)down.size(), &hand_on))) fail(r);
        }
        stamps[s][5] = now_ms();
    });
    if (trace)
        for (int s = 0; s < n; ++s)
            fprintf(stderr, "onepass slab %d: upload %.2f-%.2f  planned %.2f  my turn %.2f  filled %.2f  home %.2f ms\n", s, stamps[s][0], stamps[s][1],
                    stamps[s][2], stamps[s][3], stamps[s][4], stamps[s][5]);
    if ((rc = collect_status(mg))) return rc;
    for (int q = 0; q < 5; ++q) nnz_out[q] = mg->nnz[q] = ((want >> q) & 1u) ? next_base[q] : 0;
    return OTMB_OK;
}

}  // extern "C"
