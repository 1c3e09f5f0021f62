"""Multi-GPU: depth-slab partition of the transport-matrix assembly (one process per GPU).

Nothing distributed exists in the reference; this is the MI355X-side design of
SURVEY.md section 8e.  The wet index is k-slowest (src/matrixbuilding.jl:14-15), so a slab of
consecutive levels owns a contiguous column range of every matrix and the per-rank CSC pieces
concatenate into the global CSC.  What crosses ranks:

  setup (once per grid)   per-level wet counts (all_reduce, nz Int64)  -> global wet rank of any cell;
                          one boundary level of {v3D, rho, wet mask} to the slab above and below
                          (point-to-point over xGMI, nx*ny*8 B per field)
  per (umo,vmo) field     the continuity recurrence of facefluxes (src/velocities.jl:236-243) is a chain
                          in k with a fixed association: each slab receives ϕtop of the level below it
                          from the next rank, continues the chain, and hands its own first-level ϕtop up
                          (one nx*ny plane per boundary; never re-associated, so results stay bit-exact);
                          then one all_gather of the five per-slab nnz (+2 validity flags) -> colptr bases.

No COO triplet or matrix entry ever crosses ranks: every column is built entirely by the rank that
owns its cell (gather formulation, csrc/otmb_transportmatrix.hip), using the halo levels as neighbours.

`SlabRunner` is backend-agnostic: the product backend is `HipSlabBackend` (HIP kernels through the C
ABI); the CPU tests plug in a checker backend to exercise the orchestration with gloo.
"""
import ctypes as C

import os

import numpy as np
import torch
import torch.distributed as dist

from .capi import HDIRS, MATS, PHI_ORDER


def balanced_partition(level_counts, world):
    """Split levels 0..nz-1 into `world` consecutive slabs with >= 1 level each, wet counts as even as a
    greedy sweep gets them (upper levels are wetter, so equal level counts would not balance).  ONE rule for the
    package: the library's otmb_balanced_partition (pure host arithmetic, needs no GPU), which the single-process
    multi-GPU entry points (otmb_mgpu_*, include/otmb.h) cut their slabs with as well."""
    nz = len(level_counts)
    if world > nz:
        raise ValueError(f"{world} ranks but only {nz} levels")
    from . import capi

    return capi.balanced_partition([int(x) for x in np.asarray(level_counts, dtype=np.int64)], world)


class Comm:
    """torch.distributed plumbing.  backend "nccl" == RCCL over xGMI; with "gloo" (CPU tests, or several
    ranks sharing one GPU) device tensors are staged through host memory."""

    def __init__(self):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.backend = dist.get_backend() if dist.is_initialized() else "none"

    def _stage(self, t):
        return t.cpu() if (self.backend == "gloo" and t.is_cuda) else t

    def exchange(self, sends, recvs):
        """sends: [(tensor, dst)], recvs: [(tensor, src)] -- one batched point-to-point round."""
        if self.world == 1 or (not sends and not recvs):
            return
        ops, staged = [], []
        for t, dst in sends:
            ops.append(dist.P2POp(dist.isend, self._stage(t).contiguous(), dst))
        for t, src in recvs:
            b = self._stage(t) if self.backend == "gloo" and t.is_cuda else t
            if b is not t:
                b = torch.empty_like(b)
                staged.append((t, b))
            ops.append(dist.P2POp(dist.irecv, b, src))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for t, b in staged:
            t.copy_(b)

    def send(self, t, dst):
        dist.send(self._stage(t).contiguous(), dst)

    def isend(self, t, dst):
        """Non-blocking send; returns a handle for wait_send.  The tensor must stay unmodified until then (with gloo it
        is snapshotted to the host right here)."""
        b = self._stage(t).contiguous()
        return (dist.isend(b, dst), b)

    @staticmethod
    def wait_send(handle):
        if handle is not None:
            handle[0].wait()

    def irecv(self, t, src):
        """Non-blocking receive into t; returns a handle for wait_recv.  With RCCL the transfer runs on the communicator's
        own stream, concurrently with whatever the caller enqueues next on the compute stream."""
        if self.backend == "gloo" and t.is_cuda:
            b = torch.empty(t.shape, dtype=t.dtype)
            return (dist.irecv(b, src), b, t)
        return (dist.irecv(t, src), None, t)

    @staticmethod
    def wait_recv(handle):
        """The current stream (RCCL) or the host (gloo) waits for the receive posted by irecv."""
        if handle is None:
            return
        work, staged, t = handle
        work.wait()
        if staged is not None:
            t.copy_(staged)

    def recv(self, t, src):
        if self.backend == "gloo" and t.is_cuda:
            b = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(b, src)
            t.copy_(b)
        else:
            dist.recv(t, src)

    def allreduce_sum_i64(self, arr, device):
        t = torch.as_tensor(np.asarray(arr, dtype=np.int64))
        if self.world > 1:
            t = t.to(device) if self.backend != "gloo" else t
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather_i64(self, vals, device):
        t = torch.as_tensor(np.asarray(vals, dtype=np.int64))
        if self.world == 1:
            return t.numpy()[None, :]
        t = t.to(device) if self.backend != "gloo" else t
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return torch.stack(out).cpu().numpy()


class SlabRunner:
    """Orchestrates one rank's slab.  `grid` holds this rank's OWN levels [k0,k1) of the global grid (host
    numpy, Julia shapes) plus the replicated 2-D/1-D data; see make_local_grid()."""

    def __init__(self, backend, comm, grid, chain_pieces=None):
        self.be, self.comm, self.g = backend, comm, grid
        self.rank, self.world = comm.rank, comm.world
        self.device = backend.device
        # The facefluxes chain (src/velocities.jl:236-243) is handed from slab to slab in C row bands (whole rows): slab s starts piece c as
        # soon as piece c of slab s + 1 has arrived -- SURVEY 8e -- so that the slabs of ONE field overlap: critical path
        # t_ff / W x (1 + (W - 1) / C) instead of t_ff.  A cell's fluxes depend on its own column's plane value and on INPUTS of its west /
        # south neighbours only, so any C gives the same arrays.  Default (the rule of otmb_mgpu_set_chain_pieces): 4 on grids of 2^19
        # columns and more, else 1; OTMB_CHAIN_PIECES overrides.  The same number on every rank.
        P = int(grid["nx"]) * int(grid["ny"])
        if chain_pieces is None:
            chain_pieces = int(os.environ.get("OTMB_CHAIN_PIECES", "0")) or (4 if (P >= (1 << 19) and comm.world > 1) else 1)
        self.chain_pieces = max(1, min(int(chain_pieces), int(grid["ny"])))
        ny, nx = int(grid["ny"]), int(grid["nx"])
        self._rows = [ny * c // self.chain_pieces for c in range(self.chain_pieces + 1)]   # row bounds of the pieces
        self._cuts = [(self._rows[c] * nx, self._rows[c + 1] * nx) for c in range(self.chain_pieces)]  # the same as plane offsets
        self.setup()

    def setup(self):
        """Everything here runs on the backend's device with torch (the CPU for the gloo tests): the slab's own levels
        arrive as host arrays or as device tensors (make_local_grid), the halo planes arrive over the wire, and the
        extended local grid (halo + owned + halo), the wet mask and the GLOBAL wet ranks are built where they will be
        used -- nothing of size G passes through host numpy."""
        g, cm = self.g, self.comm
        dev = self.device
        nx, ny, nzg = g["nx"], g["ny"], g["nz_global"]
        k0, k1 = g["k0"], g["k1"]
        P, nl = nx * ny, k1 - k0
        self.P, self.k0, self.k1 = P, k0, k1
        self.has_above, self.has_below = self.rank > 0, self.rank < self.world - 1

        def flat(a, dtype=torch.float64):  # (nx,ny,nl) host array or flat tensor -> flat tensor on the device (Julia order)
            if torch.is_tensor(a):
                return a.to(device=dev, dtype=dtype).reshape(-1)
            return torch.from_numpy(np.asfortranarray(a, dtype=np.float64).ravel(order="F")).to(device=dev, dtype=dtype)

        v_own, thk_own = flat(g["v3D"]), flat(g["thkcello"])
        rho3d = None if (not torch.is_tensor(g["rho"]) and np.ndim(g["rho"]) == 0) else flat(g["rho"])
        wet_own = ~torch.isnan(v_own)  # makeindices: wet = !isnan(v3D)  (matrixbuilding.jl:15)
        counts = np.zeros(nzg, dtype=np.int64)
        counts[k0:k1] = wet_own.view(nl, P).sum(dim=1).cpu().numpy()
        counts = cm.allreduce_sum_i64(counts, dev)  # every rank learns every level's wet count
        self.level_offset = np.concatenate([[0], np.cumsum(counts)])
        self.n_global = int(self.level_offset[-1])
        self.n_own = int(self.level_offset[k1] - self.level_offset[k0])
        self.wet_base = int(self.level_offset[k0])

        # ---- static halo exchange: one boundary level of v3D, rho, wet mask each way ----
        nf = 2 + (rho3d is not None)

        def pack(q):
            sl = slice(q * P, (q + 1) * P)
            fields = [v_own[sl], wet_own[sl].to(torch.float64)]
            if rho3d is not None:
                fields.append(rho3d[sl])
            return torch.stack(fields).contiguous()

        up_recv = torch.empty((nf, P), dtype=torch.float64, device=dev) if self.has_above else None
        dn_recv = torch.empty((nf, P), dtype=torch.float64, device=dev) if self.has_below else None
        sends, recvs = [], []
        if self.has_above:
            sends.append((pack(0), self.rank - 1))
            recvs.append((up_recv, self.rank - 1))
        if self.has_below:
            sends.append((pack(nl - 1), self.rank + 1))
            recvs.append((dn_recv, self.rank + 1))
        cm.exchange(sends, recvs)

        # ---- extended local grid: [halo above] + owned + [halo below] ----
        ha, hb = int(self.has_above), int(self.has_below)
        nze = nl + ha + hb
        nanp = torch.full((P,), float("nan"), dtype=torch.float64, device=dev)

        def ext(own, up, dn):
            return torch.cat(([up] if ha else []) + [own] + ([dn] if hb else []))

        v_ext = ext(v_own, up_recv[0] if ha else None, dn_recv[0] if hb else None)
        wet_ext = ext(wet_own, (up_recv[1] > 0.5) if ha else None, (dn_recv[1] > 0.5) if hb else None)
        rho_ext = ext(rho3d, up_recv[2] if ha else None, dn_recv[2] if hb else None) if rho3d is not None else float(g["rho"])
        thk_ext = ext(thk_own, nanp, nanp)  # halos' thickness is never read
        # GLOBAL wet ranks of the extended grid: level offset + rank inside the level (+1), 0 = missing
        offs = torch.from_numpy(self.level_offset[k0 - ha:k0 - ha + nze].astype(np.int64)).to(dev)
        w2 = wet_ext.view(nze, P)
        lw_ext = ((torch.cumsum(w2.to(torch.int64), dim=1) + offs[:, None]) * w2).reshape(-1)
        zt_ext = np.asarray(g["zt_global"], dtype=np.float64)[k0 - ha:k1 + hb]
        self.nze, self.ha, self.hb = nze, ha, hb
        self.be.setup(dict(nx=nx, ny=ny, nz=nze, topology=g["topology"], k_own0=ha, k_own1=ha + nl,
                           wet_base=self.wet_base, n_own=self.n_own, v3D=v_ext, thkcello=thk_ext, rho=rho_ext,
                           lwet3d=lw_ext, wet_own=wet_own.to(torch.uint8), zt=zt_ext, edge_length_2D=g["edge_length_2D"],
                           distance_to_neighbour_2D=g["distance_to_neighbour_2D"], area2D=g["area2D"],
                           mlotst=g["mlotst"], kappa=g["kappa"], upwind=g["upwind"]))
        self.top_below = torch.empty(P, dtype=torch.float64, device=dev) if self.has_below else None
        self.n_wet_total = self.n_global

    def step(self, umo, vmo, fill):
        """One (umo, vmo) field of this rank's own levels -> this rank's columns of the five matrices."""
        cm = self.comm
        top_first = None
        for c, (a, b) in enumerate(self._cuts):
            if self.has_below:  # chain: wait for piece c of ϕtop of the level below my slab
                cm.recv(self.top_below[a:b], self.rank + 1)
            top_first = self.be.facefluxes_piece(umo, vmo, fill, self.top_below, self._rows[c], self._rows[c + 1], c == 0)
            if self.has_above:
                cm.send(top_first[a:b], self.rank - 1)
        self.be.facefluxes_finish(self.top_below)
        nnz, uv = self.be.plan()
        allv = cm.allgather_i64(list(nnz) + [int(uv[0]), int(uv[1])], self.device)
        if not (allv[:, 5].any() and allv[:, 6].any()):
            raise AssertionError("all umo or vmo values are NaN or _FillValue")  # velocities.jl:199-200
        self.nnz_base = allv[: self.rank, :5].sum(axis=0) if self.rank > 0 else np.zeros(5, dtype=np.int64)
        out = self.be.fill(self.nnz_base)
        # plan's count for T is the union-pattern bound; a slab whose T lost entries (exact-zero sums,
        # matrixbuilding.jl:147) shifts the T offsets of every slab below it
        t_all = cm.allgather_i64([int(self.be.nnz[0])], self.device)[:, 0]
        t_base = int(t_all[: self.rank].sum())
        if t_base != int(self.nnz_base[0]):
            self.be.shift_T_colptr(t_base - int(self.nnz_base[0]))
            self.nnz_base[0] = t_base
        self.nnz_global = allv[:, :5].sum(axis=0)
        self.nnz_global[0] = int(t_all.sum())
        return out

    # ---- asynchronous pipeline ----------------------------------------------------------------------------
    # step() above synchronises all ranks twice per field (the two all_gathers), so every rank waits for the
    # whole facefluxes chain: (W-1) x (one slab's facefluxes + one hop) per field.  step_async() keeps only the
    # chain's point-to-point planes: each rank's columns are written with LOCAL colptr offsets, nothing is
    # gathered, no host waits for the device.  Ranks then run skewed by one chain hop but all stay busy, and a
    # stream of fields (the TMIP workflow: one matrix per month/year) flows at the speed of the slowest slab.
    # finish() synchronises, combines the validity flags and the nnz of the LAST field and turns the local
    # colptrs into global ones (+ per-matrix base), after which the pieces concatenate as usual.
    def step_async(self, umo, vmo, fill):
        cm = self.comm
        if getattr(self, "_n_async", 0) >= self.PIPELINE_DEPTH:  # same count on every rank: the drain is collective
            self.finish()
        self._n_async = getattr(self, "_n_async", 0) + 1
        for h in getattr(self, "_pending_send", None) or ():  # the plane handed up by the previous field has left
            cm.wait_send(h)
        self._pending_send = None
        if self.has_below:
            # The plane from below was posted as non-blocking receives (one per piece) BEFORE the previous field's count/fill kernels
            # were enqueued (see below), so the transfer ran beside them; here the compute stream only waits for piece after piece.
            if getattr(self, "_recv", None) is None:  # first field since setup / finish
                self._recv_bufs = getattr(self, "_recv_bufs", None) or [self.top_below, torch.empty_like(self.top_below)]
                self._recv_cur = 0
                self._recv = [cm.irecv(self._recv_bufs[0][a:b], self.rank + 1) for a, b in self._cuts]
            plane = self._recv_bufs[self._recv_cur]
        else:
            plane = None
        if self.has_above and getattr(self, "_send_buf", None) is None:
            self._send_buf = torch.empty(self.be.P, dtype=torch.float64, device=self.device)
        sends = []
        for c, (a, b) in enumerate(self._cuts):
            if self.has_below:
                cm.wait_recv(self._recv[c])
            top_first = self.be.facefluxes_piece(umo, vmo, fill, plane, self._rows[c], self._rows[c + 1], c == 0)
            if self.has_above:
                # the piece goes up from its own buffer without holding back this rank's later kernels: nothing waits for the send
                # until the next field is about to reuse the buffer
                self._send_buf[a:b].copy_(top_first[a:b])
                sends.append(cm.isend(self._send_buf[a:b], self.rank - 1))
        self.be.facefluxes_finish(plane)
        self._pending_send = sends or None
        if self.has_below:  # the NEXT field's plane (or finish()'s closing message) lands in the other buffer meanwhile
            self._recv_cur ^= 1
            self._recv = [cm.irecv(self._recv_bufs[self._recv_cur][a:b], self.rank + 1) for a, b in self._cuts]
        self.be.assemble_async()

    PIPELINE_DEPTH = 60  # the library keeps the verdicts of its 64 most recent asynchronous calls

    # reference order of the checks inside one transportmatrix call (src/matrixbuilding.jl:233, the loop, :39, :61, :90, :114)
    _STATUS_ORDER = (13, 10, 15, 1, 6, 2, 3, 4, 5, 14)

    def finish(self):
        """Drain the pipeline on every rank.  Each rank reports, per pending step, its slab's two facefluxes validity flags
        and the first step its transportmatrix failed in; ONE all_gather later every rank knows the same thing and raises
        the same error -- that of the earliest failing step (facefluxes' assertion first within a step, as in the
        reference), so no rank is left waiting in a collective for one that has raised."""
        cm = self.comm
        for h in getattr(self, "_pending_send", None) or ():
            cm.wait_send(h)
        self._pending_send = None
        # every rank with a slab below keeps one plane of receives posted ahead: close them with one last (unused) plane
        if self.has_above and getattr(self, "_send_buf", None) is not None and getattr(self, "_n_async", 0) > 0:
            for h in [cm.isend(self._send_buf[a:b], self.rank - 1) for a, b in self._cuts]:
                cm.wait_send(h)
        if self.has_below and getattr(self, "_recv", None) is not None:
            for h in self._recv:
                cm.wait_recv(h)
            self._recv = None
        r = self.be.result()  # never raises for the reference's own errors: dict(nnz, u, v, status, step, message)
        D = self.PIPELINE_DEPTH + 4
        n_steps = len(r["u"])
        vec = list(r["nnz"]) + [int(r["status"]), int(r["step"]), n_steps]
        vec += [int(x) for x in r["u"]] + [0] * (D - n_steps) + [int(x) for x in r["v"]] + [0] * (D - n_steps)
        allv = cm.allgather_i64(vec, self.device)
        self._n_async = 0
        n_steps = int(allv[:, 7].min())
        u_any, v_any = allv[:, 8:8 + D].any(axis=0), allv[:, 8 + D:8 + 2 * D].any(axis=0)
        missing = [q for q in range(n_steps) if not (u_any[q] and v_any[q])]  # velocities.jl:199-200, over the whole grid
        failed = [(int(row[6]), self._STATUS_ORDER.index(int(row[5])) if int(row[5]) in self._STATUS_ORDER else 99, int(row[5]), rk)
                  for rk, row in enumerate(allv) if int(row[5]) != 0]
        first_tm = min(failed) if failed else None
        where = lambda q: f" (asynchronous step {q + 1} of {n_steps})" if n_steps > 1 else ""
        if missing and (first_tm is None or missing[0] <= first_tm[0]):
            raise AssertionError("all umo or vmo values are NaN or _FillValue" + where(missing[0]))
        if first_tm is not None:
            from .capi import OtmbError, lib

            step, _, status, rk = first_tm
            raise OtmbError(status, lib().otmb_status_string(status).decode() + where(step) + f" [slab of rank {rk}]", step=step)
        self.nnz_base = allv[: self.rank, :5].sum(axis=0) if self.rank > 0 else np.zeros(5, dtype=np.int64)
        self.nnz_global = allv[:, :5].sum(axis=0)
        self.be.shift_colptr(self.nnz_base)
        return self.be.out

    def sync(self):
        self.be.sync()


class HipSlabBackend:
    """Product backend: HIP kernels through the C ABI on this rank's GPU."""

    def __init__(self, local_rank=0):
        from . import capi

        if not torch.cuda.is_available():
            raise RuntimeError("HipSlabBackend needs a GPU (no CPU fallback)")
        self.capi = capi
        self.device = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.device)
        self.ctx = capi.Context(local_rank)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.lib = capi.lib()
        self.out = None

    def _t(self, a, dtype=np.float64):
        if torch.is_tensor(a):  # already on a device (SlabRunner.setup builds the extended grid there)
            tdt = {np.float64: torch.float64, np.int64: torch.int64, np.uint8: torch.uint8}[dtype]
            return a.to(device=self.device, dtype=tdt).reshape(-1).contiguous()
        return torch.from_numpy(np.asfortranarray(a, dtype=dtype).ravel(order="F")).to(self.device)

    def setup(self, s):
        self.s = s
        self.nx, self.ny, self.nz = s["nx"], s["ny"], s["nz"]
        self.P = self.nx * self.ny
        self.G = self.P * self.nz
        self.n_own = s["n_own"]
        self.v3d, self.thk, self.lw = self._t(s["v3D"]), self._t(s["thkcello"]), self._t(s["lwet3d"], np.int64)
        scalar_rho = not torch.is_tensor(s["rho"]) and np.ndim(s["rho"]) == 0
        self.rho = None if scalar_rho else self._t(s["rho"])
        self.rho_scalar = float(s["rho"]) if scalar_rho else 0.0
        self.edge = [self._t(s["edge_length_2D"][d]) for d in HDIRS]
        self.dist_ = [self._t(s["distance_to_neighbour_2D"][d]) for d in HDIRS]
        self.area, self.zt, self.ml = self._t(s["area2D"]), self._t(s["zt"]), self._t(s["mlotst"])
        self.wet_own = self._t(s["wet_own"], np.uint8)
        self.phi = [torch.zeros(self.G, dtype=torch.float64, device=self.device) for _ in range(6)]
        self.push_mask = torch.zeros(self.G, dtype=torch.int16, device=self.device)
        self.own0 = s["k_own0"] * self.P
        self.nown_lev = s["k_own1"] - s["k_own0"]
        # Lwet of the owned cells: local linear indices (1-based) inside the extended grid, ascending
        own = torch.nonzero(self.lw[self.own0:self.own0 + self.nown_lev * self.P]).flatten() + (self.own0 + 1)
        assert own.numel() == self.n_own
        self.lwet = own.to(torch.int64).contiguous()
        self._setup_counts()

    def _setup_counts(self):
        """Counts in facefluxes for this slab (otmb_facefluxes_slab_counts_dev; OTMB_COUNT_IN_FF=0 switches it off): the five-flag bytes
        of the EXTENDED local grid (the halo levels are neighbours) and the grid's tables.  Once per grid."""
        self.counts = None
        if os.environ.get("OTMB_COUNT_IN_FF", "1") == "0" or self.n_own == 0:
            return
        capi = self.capi
        wet_ext = (self.lw != 0).to(torch.uint8)
        self.wetflags = torch.empty(self.G, dtype=torch.uint8, device=self.device)
        self.ctx.check(self.lib.otmb_wetflags_dev(self.ctx.handle, wet_ext.data_ptr(), self.nx, self.ny, self.nz, self.s["topology"],
                                                  self.wetflags.data_ptr()))
        self.ff_slab = capi.FfSlab(self.s["k_own0"], self.nz, self.s["wet_base"])
        nbytes = int(self.lib.otmb_count_tables_bytes(self.ctx.handle, self.nx, self.ny, self.nown_lev, self.n_own))
        self.count_tables = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=self.device)
        self.ctx.check(self.lib.otmb_count_tables_slab_dev(self.ctx.handle, self.lw.data_ptr(), self.lwet.data_ptr(), self.wetflags.data_ptr(),
                                                           self.n_own, self.nx, self.ny, self.nown_lev, self.s["topology"],
                                                           C.byref(self.ff_slab), self.count_tables.data_ptr()))
        self.ctx.synchronize()  # (wet_ext is released when this returns)
        c = capi.FfCounts()
        c.tables, c.lwet3d, c.mlotst, c.zt = self.count_tables.data_ptr(), self.lw.data_ptr(), self.ml.data_ptr(), self.zt.data_ptr()
        c.n_wet, c.upwind, c.only_t = self.n_own, int(self.s["upwind"]), 0
        self.counts = c

    def facefluxes(self, umo, vmo, fill, top_below):
        """The whole plane at once (one piece)."""
        self.facefluxes_piece(umo, vmo, fill, top_below, 0, self.ny, True)
        return self.facefluxes_finish(top_below)

    def facefluxes_piece(self, umo, vmo, fill, top_below, j0, j1, first):
        """Rows [j0, j1) of the plane (otmb_facefluxes_rows_dev): one piece of the chain.  top_below: the whole-plane buffer, of which
        rows [j0, j1) have arrived.  Returns the (whole-plane) view of this slab's first-level ϕtop, of which rows [j0, j1) are now valid."""
        o, n = self.own0, self.nown_lev * self.P
        views = [p[o:o + n] for p in self.phi]
        ptrs = self.capi.ptr_array(6, [v.data_ptr() for v in views])
        if self.counts is not None:
            # the kernel also accumulates the tile counts of this slab's transportmatrix: assemble_async / plan skip their counting pass
            # (the library falls back to the plain call, mask written, where it cannot count: the caller need not know)
            self.ctx.check(self.lib.otmb_facefluxes_slab_counts_dev(
                self.ctx.handle, umo.data_ptr(), vmo.data_ptr(), int(umo.dtype == torch.float32), self.wetflags.data_ptr(),
                float(fill), self.nx, self.ny, self.nown_lev, self.s["topology"], C.byref(ptrs),
                top_below.data_ptr() if top_below is not None else None, self.push_mask[o:o + n].data_ptr(), C.byref(self.counts),
                C.byref(self.ff_slab), int(j0), int(j1), int(bool(first))))
            return self.phi[4][o:o + self.P]
        self.ctx.check(self.lib.otmb_facefluxes_rows_dev(
            self.ctx.handle, umo.data_ptr(), vmo.data_ptr(), int(umo.dtype == torch.float32), self.wet_own.data_ptr(),
            float(fill), self.nx, self.ny, self.nown_lev, self.s["topology"], C.byref(ptrs),
            top_below.data_ptr() if top_below is not None else None, self.push_mask[o:o + n].data_ptr(), int(j0), int(j1), int(bool(first))))
        return self.phi[4][o:o + self.P]  # OTMB_TOP, first owned level

    def facefluxes_finish(self, top_below):
        """After the last piece: the halo levels act as neighbours only -- the one flux each of them pushes into an owned cell, and its
        push mask."""
        o = self.own0
        top, bottom = self.phi[4], self.phi[5]  # OTMB_TOP, OTMB_BOTTOM
        top_first = top[o:o + self.P]
        allphi = self.capi.ptr_array(6, [p.data_ptr() for p in self.phi])
        # (the counting facefluxes kernel has already looked at the halo cells as neighbours: no push mask at all in that case)
        masks = not (self.counts is not None and self.lib.otmb_facefluxes_counts_pending(self.ctx.handle))
        if self.s["k_own0"] > 0:  # halo above: its ϕbottom is my first level's ϕtop (velocities.jl:240)
            bottom[0:self.P].copy_(top_first)
            if masks:
                self.ctx.check(self.lib.otmb_push_mask_dev(self.ctx.handle, C.byref(allphi), self.lw.data_ptr(), 0, self.P,
                                                           self.push_mask.data_ptr()))
        if top_below is not None:  # halo below: its ϕtop is the plane received from the slab below
            top[self.G - self.P:self.G].copy_(top_below)
            if masks:
                self.ctx.check(self.lib.otmb_push_mask_dev(self.ctx.handle, C.byref(allphi), self.lw.data_ptr(), self.G - self.P,
                                                           self.P, self.push_mask.data_ptr()))
        return top_first

    def _tm_args(self):
        a = self.capi.TmArgs()
        a.nx, a.ny, a.nz = self.nx, self.ny, self.nz
        a.topology, a.upwind, a.n_wet = self.s["topology"], int(self.s["upwind"]), self.n_own
        for k in range(6):
            a.phi[k] = self.phi[k].data_ptr()
        a.v3d, a.thkcello = self.v3d.data_ptr(), self.thk.data_ptr()
        a.rho = self.rho.data_ptr() if self.rho is not None else None
        a.rho_scalar = self.rho_scalar
        a.lwet3d = self.lw.data_ptr()
        a.lwet = self.lwet.data_ptr()
        for k in range(4):
            a.edge_length[k] = self.edge[k].data_ptr()
            a.dist_nbr[k] = self.dist_[k].data_ptr()
        a.area2d, a.zt, a.mlotst = self.area.data_ptr(), self.zt.data_ptr(), self.ml.data_ptr()
        a.kappa_h, a.kappa_vml, a.kappa_vdeep = self.s["kappa"]
        a.push_mask = self.push_mask.data_ptr()  # written by facefluxes() for exactly self.phi
        return a

    def plan(self):
        a = self._tm_args()
        self._cap = None
        u, v = C.c_int32(0), C.c_int32(0)
        self.ctx.check(self.lib.otmb_facefluxes_slab_flags(self.ctx.handle, C.byref(u), C.byref(v)))
        self.ctx.check(self.lib.otmb_transportmatrix_set_slab(self.ctx.handle, self.s["wet_base"]))
        nnz = (C.c_int64 * 5)()
        self.ctx.check(self.lib.otmb_transportmatrix_plan_dev(self.ctx.handle, C.byref(a), C.byref(nnz)))
        self.nnz = [int(x) for x in nnz]
        return self.nnz, (u.value, v.value)

    def fill(self, nnz_base):
        if self.out is None or any(self.out[m][1].numel() < self.nnz[k] for k, m in enumerate(MATS)):
            self.out = {m: (torch.empty(self.n_own + 1, dtype=torch.int64, device=self.device),
                            torch.empty(max(self.nnz[k], 1) + self.nnz[k] // 64, dtype=torch.int64, device=self.device),
                            torch.empty(max(self.nnz[k], 1) + self.nnz[k] // 64, dtype=torch.float64, device=self.device))
                        for k, m in enumerate(MATS)}
        base = (C.c_int64 * 5)(*[int(x) for x in nnz_base])
        self.ctx.check(self.lib.otmb_transportmatrix_set_nnz_base(self.ctx.handle, C.byref(base)))
        cp = self.capi.ptr_array(5, [self.out[m][0].data_ptr() for m in MATS])
        rv = self.capi.ptr_array(5, [self.out[m][1].data_ptr() for m in MATS])
        nz = self.capi.ptr_array(5, [self.out[m][2].data_ptr() for m in MATS])
        self.ctx.check(self.lib.otmb_transportmatrix_fill_dev(self.ctx.handle, C.byref(cp), C.byref(rv), C.byref(nz)))
        final = (C.c_int64 * 5)()
        self.ctx.check(self.lib.otmb_transportmatrix_nnz(self.ctx.handle, C.byref(final)))
        self.nnz = [int(x) for x in final]
        return self.out

    def shift_T_colptr(self, delta):
        self.out[MATS[0]][0].add_(int(delta))

    PER_COLUMN_MAX = (7, 7, 5, 3, 3)

    def assemble_async(self):
        """otmb_transportmatrix_dev on this slab: count -> scan -> fill enqueued with local colptr offsets."""
        if self.out is None or getattr(self, "_cap", None) is None:
            cap = [self.n_own * k + 1 for k in self.PER_COLUMN_MAX]
            self.out = {m: (torch.empty(self.n_own + 1, dtype=torch.int64, device=self.device),
                            torch.empty(cap[k], dtype=torch.int64, device=self.device),
                            torch.empty(cap[k], dtype=torch.float64, device=self.device)) for k, m in enumerate(MATS)}
            self._cap = cap
        a = self._tm_args()
        self.ctx.check(self.lib.otmb_transportmatrix_set_slab(self.ctx.handle, self.s["wet_base"]))
        cp = self.capi.ptr_array(5, [self.out[m][0].data_ptr() for m in MATS])
        rv = self.capi.ptr_array(5, [self.out[m][1].data_ptr() for m in MATS])
        nz = self.capi.ptr_array(5, [self.out[m][2].data_ptr() for m in MATS])
        caps = (C.c_int64 * 5)(*self._cap)
        self.ctx.check(self.lib.otmb_transportmatrix_dev(self.ctx.handle, C.byref(a), C.byref(cp), C.byref(rv), C.byref(nz),
                                                         C.byref(caps)))
        self._n_async = getattr(self, "_n_async", 0) + 1

    def result(self):
        """Verdicts of every asynchronous step since the previous call.  The reference's own failures are RETURNED
        (status, step, message), not raised: SlabRunner.finish lets every rank raise the same one after its all_gather."""
        cap = 64
        u, v, n = (C.c_int32 * cap)(), (C.c_int32 * cap)(), C.c_int32(0)
        self.ctx.check(self.lib.otmb_facefluxes_pending_flags(self.ctx.handle, cap, u, v, C.byref(n)))
        nnz = (C.c_int64 * 5)()
        rc = self.lib.otmb_transportmatrix_result(self.ctx.handle, C.byref(nnz))
        step, msg = C.c_int64(-1), ""
        if rc != self.capi.OK:
            msg = self.lib.otmb_last_error(self.ctx.handle).decode("utf-8")
            if rc in (9, 10, 11, 12):  # allocation / HIP / usage errors are not the reference's: fail here and now
                raise self.capi.OtmbError(rc, msg)
            self.lib.otmb_transportmatrix_failed_step(self.ctx.handle, C.byref(step))
        else:
            self.nnz = [int(x) for x in nnz]
        # the verdicts of the asynchronous steps are the LAST _n_async facefluxes calls (synchronous steps before them
        # were checked when they ran)
        first = max(0, n.value - getattr(self, "_n_async", 0))
        self._n_async = 0
        return dict(nnz=list(self.nnz) if rc == self.capi.OK else [0] * 5, u=[u[q] for q in range(first, n.value)],
                    v=[v[q] for q in range(first, n.value)], status=int(rc), step=int(step.value), message=msg)

    def shift_colptr(self, bases):
        for k, m in enumerate(MATS):
            if int(bases[k]):
                self.out[m][0].add_(int(bases[k]))

    def sync(self):
        self.ctx.synchronize()

    def result_to_host(self):
        self.sync()
        return {m: (self.out[m][0].cpu().numpy(), self.out[m][1][: self.nnz[k]].cpu().numpy(),
                    self.out[m][2][: self.nnz[k]].cpu().numpy()) for k, m in enumerate(MATS)}


def make_local_grid(gridmetrics, mlotst, rho, k0, k1, nz_global, zt_global, kappa=(500.0, 0.1, 1.0e-5), upwind=True):
    """Bundle one rank's own levels.  `gridmetrics` are those of the OWN levels only (makegridmetrics on the
    slab's volcello); the 2-D metrics are the same on every rank."""
    t = gridmetrics["gridtopology"]
    nx, ny, _ = gridmetrics["v3D"].shape
    return dict(nx=nx, ny=ny, nz_global=nz_global, k0=k0, k1=k1, topology=int(t["kind"]) if isinstance(t, dict) else int(t),
                v3D=gridmetrics["v3D"], thkcello=gridmetrics["thkcello"], rho=rho, zt_global=zt_global,
                edge_length_2D=gridmetrics["edge_length_2D"], distance_to_neighbour_2D=gridmetrics["distance_to_neighbour_2D"],
                area2D=gridmetrics["area2D"], mlotst=mlotst, kappa=kappa, upwind=upwind)


def make_local_grid_from_device(dg, upwind=True):
    """The same bundle from a device-generated slab (synthetic_device.make_device_grid(..., k0=, k1=)): the 3-D fields stay
    on the device as flat tensors, the replicated 2-D metrics are the host arrays of its gridmetrics."""
    gm = dg.gm
    return dict(nx=dg.nx, ny=dg.ny, nz_global=dg.nz, k0=dg.k0, k1=dg.k1, topology=dg.topology, v3D=dg.v3d, thkcello=dg.thkcello,
                rho=dg.rho, zt_global=dg.zt_host, edge_length_2D=gm["edge_length_2D"],
                distance_to_neighbour_2D=gm["distance_to_neighbour_2D"], area2D=gm["area2D"], mlotst=dg.mlotst_host,
                kappa=(dg.kappaH, dg.kappaVML, dg.kappaVdeep), upwind=upwind)


def gather_global_csc(comm, local, n_own, nnz, device):
    """Concatenate the ranks' pieces into the global (colptr,rowval,nzval) on every rank (tests / small grids)."""
    out = {}
    for k, m in enumerate(MATS):
        cp, rv, nz = local[m]
        pieces = [None] * comm.world
        obj = (np.asarray(cp[:n_own]), np.asarray(rv[: nnz[k]]), np.asarray(nz[: nnz[k]]), int(cp[n_own]))
        if comm.world > 1:
            dist.all_gather_object(pieces, obj)
        else:
            pieces = [obj]
        out[m] = (np.concatenate([p[0] for p in pieces] + [np.array([pieces[-1][3]], dtype=np.int64)]),
                  np.concatenate([p[1] for p in pieces]), np.concatenate([p[2] for p in pieces]))
    return out
