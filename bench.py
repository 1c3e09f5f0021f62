#!/usr/bin/env python3
"""bench.py -- wet-cells/s assembled into T on the synthetic ACCESS-ESM1-5-like 1° grid.

One "step" = one pass of the hot path over one (umo, vmo) field, all device resident:
    facefluxesfrommasstransport -> transportmatrix (T, Tadv, TκH, TκVML, TκVdeep in CSC)
Inputs are in HBM before the timed region starts and the five CSC matrices are left in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload access1deg|quarterdeg|tenthdeg|small]
                    [--scaling weak|strong] [--repeats R]

N > 1: one rank per GPU.  Launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in the environment) each process
is one rank; launched plainly (`python bench.py --gpus N`) this process only SPAWNS the N ranks -- before anything here
touches a GPU -- relays rank 0's JSON line and exits with the worst return code.  The grid is partitioned in depth
(otmb_amd/dist.py):
    --scaling weak   (default) the global grid is the workload's horizontal grid with nz*N levels: per-GPU work fixed;
    --scaling strong the workload's own grid (e.g. --workload quarterdeg: BASELINE.json configs[3], the fixed
                     1440x1080x75 grid) cut into N depth slabs with balanced wet counts: total work fixed.
value = wet cells of all ranks / max-over-ranks time.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="the K timed steps are repeated R times; ms_per_step is the median")
    ap.add_argument("--workload", default="access1deg")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--rho", default="array", choices=["array", "scalar"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--seed", type=int, default=20260501)
    ap.add_argument("--extra-configs", default="quarterdeg,tenthdeg",
                    help="after the headline workload (N = 1 only): BASELINE.json configs[2] (0.25 degree, the HBM-roofline run) and configs[4]'s "
                         "grid (0.1 degree, on ONE GPU) as extra records of the same JSON line; '' = none")
    ap.add_argument("--placement-candidates", type=int, default=4,
                    help="set-up, N = 1 only: DeviceAssembler.choose_placement -- the flux arrays and the output matrices are allocated this many times, "
                         "facefluxes / the fill pass timed on each, the fastest kept (profiles/r04/README.md section 12); 1 = off")
    ap.add_argument("--protocol", default="async", choices=["async", "twophase"],
                    help="async: otmb_transportmatrix_dev (count -> scan -> fill enqueued back to back, outputs preallocated "
                         "at their upper bound); twophase: plan (host learns nnz) then fill, as a caller that sizes its outputs does")
    args = ap.parse_args()
    args.steps = max(1, args.steps)
    args.warmup = max(0, args.warmup)
    args.repeats = max(1, args.repeats)
    return args


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this parent never
    initialises a GPU and never re-executes itself), relay rank 0's JSON line."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # (the script that was started, not necessarily this file: tests/bench_rehearsal.py wraps main() and must be what the ranks run)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0]), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll every rank: when one dies the others would wait for it in a collective for ever -- stop them (they are this
    # process's own children) and report; an overall limit bounds the run whatever happens
    import threading

    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("OTMB_BENCH_TIMEOUT", "1500"))
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            failed = bad[0] if bad else ("timeout", 124)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            time.sleep(2)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5)
    out = (out_chunks[0] if out_chunks else b"").decode()
    line = next((l for l in out.splitlines() if l.startswith("{")), None)
    if failed is not None:
        print(f"bench.py: rank {failed[0]} ended with status {failed[1]}; the other ranks were stopped", file=sys.stderr)
        return failed[1] if isinstance(failed[1], int) and failed[1] > 0 else 1
    if line:
        print(line, flush=True)
    return max(abs(rc) for rc in rcs) or (0 if line else 1)


def cpu_baseline(g, gm, workload, reps=5):
    """The oracle (single-thread C restatement of the reference algorithm: push COO -> sparse() x4 ->
    3 sparse adds) timed on this box's host cores over the same workload: 1 warm-up + median of `reps`
    passes.  kind = "port": the Julia reference cannot run here (no julia binary; SURVEY.md
    section 8c)."""
    import numpy as np

    from oracle import oracle as orc

    orc.build()
    idx = orc.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    N = idx["N"]
    t_ff, t_tm = [], []
    for rep in range(reps + 1):
        t1 = time.perf_counter()
        phi = orc.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], fill, gm.gridtopology.kind)
        t2 = time.perf_counter()
        orc.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True, tight=True)
        t3 = time.perf_counter()
        if rep:  # the first pass is the warm-up
            t_ff.append(t2 - t1)
            t_tm.append(t3 - t2)
    ff, tm = float(np.median(t_ff)), float(np.median(t_tm))
    # second figure: the same algorithm on several host threads -- the four operator builds run
    # concurrently and the three adds are column-parallel (oracle/otmb_oracle.c, orc_transportmatrix_omp); the
    # scatter-then-sort formulation of the reference offers no more parallelism without being changed
    t_par = []
    for rep in range(4):
        t1 = time.perf_counter()
        orc.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True, tight=True, parallel=True)
        if rep:
            t_par.append(time.perf_counter() - t1)
    tmp = float(np.median(t_par))
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        model = "unknown CPU"
    return {
        "value": N / (ff + tm), "unit": "wet-cells/s", "cores": 1, "kind": "port",
        "sample": f"facefluxes+transportmatrix on the full {workload} grid (N={N}), 1 warm-up + median of {reps} passes; "
                  f"facefluxes {ff:.3f} s, transportmatrix {tm:.3f} s (COO generation + sparse() x4 + 3 adds); "
                  f"host {os.cpu_count()} logical cores ({model}), 1 used",
        "seconds": ff + tm,
        "multithread": {"value": N / (ff + tmp), "unit": "wet-cells/s", "cores": min(orc.omp_threads(), os.cpu_count() or 1),
                        "seconds": ff + tmp,
                        "sample": f"same workload, OpenMP: 4 concurrent operator builds + column-parallel adds "
                                  f"({orc.omp_threads()} threads), transportmatrix {tmp:.3f} s, median of 3; facefluxes as above"},
    }


def kernel_source_hash():
    """profiles/traffic.json holds counter traffic measured for ONE version of the dominant kernel: it is keyed to a hash of
    the kernel's sources and ignored (traffic: null) when they have changed since."""
    h = hashlib.sha256()
    for f in ("otmb_transportmatrix.hip", "otmb_tm_column.h", "otmb_facefluxes.hip"):
        with open(os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def end_to_end(g, gm, asm_N, reps=5, warm=5):
    """What a Julia caller of the host-pointer C ABI sees (PCIe included; never `value`): facefluxesfrommasstransport +
    transportmatrix through otmb_amd.api on host arrays, median of `reps` time slices after `warm` warm-up slices (the first two slices of
    a loop allocate: device buffers, the pinned rings of the slab contexts, the pinned result blocks -- 123 and 31 ms against 23.4 ms from the
    third on, tools/onepass_loop.py; and the default call spends its first five slices measuring which protocol this host is faster
    with, api.Trial)."""
    import numpy as np

    import otmb_amd
    import otmb_amd.api as api

    idx = api.makeindices(gm.v3D)
    res = {}
    # default: what `transportmatrix(; ϕ, ...)` does with no extension keyword -- on a grid of this size the pipelined one-phase build on 4
    # depth slabs of the GPU (api.default_slabs; otmb_mgpu_transportmatrix_onepass); two_phase: slabs=0, the plan -> allocate -> fetch call of
    # rounds 1-4; reuse: both reuse promises (then always two-phase: nothing is left to upload beside the download)
    for name, kw in (("default", {}), ("two_phase", {"slabs": 0}), ("reuse", {"reuse_grid": True, "reuse_fluxes": True})):
        ts = []
        for rep in range(reps + warm):
            t0 = time.perf_counter()
            phi = api.facefluxesfrommasstransport(umo=g.umo, vmo=g.vmo, gridmetrics=gm, indices=idx)
            t1 = time.perf_counter()
            tm = api.transportmatrix(ϕ=phi, mlotst=g.mlotst, gridmetrics=gm, indices=idx, ρ=g.rho, κH=g.kappaH, κVML=g.kappaVML,
                                     κVdeep=g.kappaVdeep, **kw)
            t2 = time.perf_counter()
            if rep >= warm:
                ts.append((t1 - t0, t2 - t1, api.last_call_seconds["plan"] + api.last_call_seconds["fetch"]))
            del tm, phi  # (a time-slice loop drops the previous matrices: their pinned blocks return to the context's pool)
        res[name] = tuple(float(np.median([x[q] for x in ts])) for q in range(3))
    ff, tm, cabi = res["default"]
    return {"value": asm_N / (ff + tm), "unit": "wet-cells/s", "facefluxes_ms": 1e3 * ff, "transportmatrix_ms": 1e3 * tm,
            "transportmatrix_c_abi_ms": 1e3 * cabi, "slabs": api.default_slabs(asm_N, int(g.umo.data.shape[2]), False, None),
            "default_protocol": ("pipelined" if api.Trial.of(0, int(asm_N)).now else "two-phase") + " (measured by the first five calls)",
            "transportmatrix_ms_two_phase": 1e3 * res["two_phase"][1], "value_two_phase": asm_N / (res["two_phase"][0] + res["two_phase"][1]),
            "facefluxes_ms_reuse": 1e3 * res["reuse"][0], "transportmatrix_ms_reuse": 1e3 * res["reuse"][1],
            "transportmatrix_c_abi_ms_reuse": 1e3 * res["reuse"][2], "value_reuse": asm_N / (res["reuse"][0] + res["reuse"][1]),
            "note": "host-pointer C ABI (what a Julia ccall passes): pageable host input arrays, five host CSC matrices out in pinned memory "
                    "of the library (otmb_host_alloc: the DMA writes them in place), PCIe both ways.  transportmatrix_ms: the default call -- "
                    "pipelined over `slabs` depth slabs of the GPU, a slab uploading while the one above it copies its columns home, row indices "
                    "and column offsets crossing the link as Int32 (otmb_mgpu_transportmatrix_onepass, OtmbXferItem.narrow); *_two_phase: slabs=0, otmb_transportmatrix_plan + _fetch (every upload before the "
                    "count, every download after it: rounds 1-4); c_abi_ms: inside the C calls; *_reuse: gridmetrics / indices uploaded once "
                    "(reuse_grid) and the face fluxes that facefluxes just computed not uploaded again (reuse_fluxes; two-phase)"}


def extra_configs_in_children(args):
    """The other single-GPU BASELINE configurations, each measured by a FRESH process of this same script (`--workload X`) that has the
    GPU to itself, BEFORE this process touches the GPU: where the allocator places an assembler's arrays moves the fill pass by +-10 %, and
    allocations made after another workload's blocks were freed measure up to 10 % slower than a fresh process's (profiles/r03/README.md,
    section 6) -- so every configuration is reported as what `python bench.py --workload X` prints.  Returns {key: record}."""
    names = {"quarterdeg": "config3", "tenthdeg": "config5"}
    out = {}
    for wl in [w for w in args.extra_configs.split(",") if w and w != args.workload]:
        key = names.get(wl, "config_" + wl)
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--extra-configs", "", "--no-cpu-baseline", "--no-end-to-end",
               "--steps", str(min(args.steps, 10)), "--warmup", "2", "--repeats", "2", "--rho", args.rho, "--seed", str(args.seed),
               "--placement-candidates", str(args.placement_candidates)]
        rec = {"workload": wl}
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=float(os.environ.get("OTMB_BENCH_EXTRA_TIMEOUT", "600")))
            line = next((l for l in r.stdout.decode().splitlines() if l.startswith("{")), None)
            if r.returncode != 0 or line is None:
                tail = r.stderr.decode()[-300:].replace("\n", " | ")
                rec["skipped" if "out of memory" in tail.lower() else "error"] = f"child exited with {r.returncode}: {tail}"
            else:
                d = json.loads(line)
                rec.update({"grid": d["config"]["workload"].split("grid ")[1].split(",")[0], "wet_cells": d["config"]["wet_cells"], "nnz": d["config"]["nnz"],
                            "protocol": d["config"]["protocol"], "steps": d["steps"], "warmup": d["warmup"], "ms_per_step": d["ms_per_step"],
                            "value": d["value"], "unit": d["unit"], "repeats": d["repeats"], "roofline": d["roofline"], "kernels_ms": d["kernels_ms"],
                            "step_gbs": d["step_gbs"], "placement": d.get("placement"), "fused_step": d.get("fused_step"), "measured_by": "a fresh process: " + " ".join(cmd[1:6])})
        except subprocess.TimeoutExpired:
            rec["error"] = "child timed out"
        except Exception as e:  # an extra record must never cost the headline
            rec["error"] = f"{type(e).__name__}: {e}"[:300]
        out[key] = rec
    return out


def slab_config(workload, args, world, rank, dev, local_rank, rehearsal, backend, make_backend):
    """BASELINE.json configs[3] beside a multi-rank headline: the FIXED `workload` grid cut into `world` depth slabs (strong scaling),
    facefluxes chain planes over the communicator, asynchronous pipeline; K' = min(K, 10) steps x 2 repeats between barriers, max over
    ranks.  Every rank takes part; rank 0 returns the record."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from otmb_amd import dist as odist
    from otmb_amd import synthetic, synthetic_device

    rec = {"workload": workload, "scaling": "strong", "n_gpus": world}
    nx, ny, nz, lf = synthetic.PRESETS[workload]
    counts = synthetic.level_wet_counts(nx, ny, nz, seed=args.seed, land_fraction=lf)
    k0, k1 = odist.balanced_partition(counts, world)[rank]
    dg = synthetic_device.make_device_grid((nx, ny, nz), dev, seed=args.seed, land_fraction=lf, rho=args.rho, k0=k0, k1=k1)
    local = odist.make_local_grid_from_device(dg)
    be = make_backend(local_rank)
    srun = odist.SlabRunner(be, odist.Comm(), local)

    def barrier():
        dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    k = min(args.steps, 10)
    for _ in range(2):
        srun.step_async(dg.umo, dg.vmo, dg.fill)
    srun.finish()
    be.sync()
    per = []
    for _ in range(2):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            srun.step_async(dg.umo, dg.vmo, dg.fill)
        srun.finish()
        be.sync()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per.append(float(t.item()) / k)
    ms = 1e3 * float(np.median(per))
    rec.update({"grid": f"{nx}x{ny}x{nz}", "rank0_levels": [int(k0), int(k1)], "wet_cells": int(srun.n_global),
                "nnz": dict(zip(("T", "Tadv", "TkH", "TkVML", "TkVdeep"), (int(x) for x in srun.nnz_global))), "protocol": "async",
                "steps": k, "ms_per_step": ms, "value": srun.n_global / (ms * 1e-3), "unit": "wet-cells/s",
                "ms_per_step_min": 1e3 * min(per), "ms_per_step_max": 1e3 * max(per)})
    del srun, be, dg, local
    import gc

    gc.collect()
    if not rehearsal:
        torch.cuda.empty_cache()
    return rec


def config2_bolus(g, gm, dev, local_rank, steps):
    """BASELINE.json configs[1] ("... incl. Redi/GM triads").  In the reference the triads never enter T (src/RediGM.jl:44; DESIGN.md
    section 0): the headline line IS config 2's transportmatrix (3-D ρ).  What the reference has of Redi/GM is bolus_GM_velocity
    (src/RediGM.jl:46-79): its two kernels on the same grid, device resident, as an extra record (never `value`).  Algorithmic bytes:
    ρ, Z3D (8 B), the wet byte and two (nx,ny) distances in, u and v out = 33 B per cell; the κGM·S intermediates are overhead."""
    import ctypes as C

    import numpy as np
    import torch

    from otmb_amd import capi

    ctx = capi.Context(local_rank)
    nx, ny, nz = gm.v3D.shape
    flat = lambda a: torch.from_numpy(np.asfortranarray(a, dtype=np.float64).ravel(order="F")).to(dev)
    dn = gm.distance_to_neighbour_2D
    rho, z3d, de, dnn = flat(g.rho), flat(gm.Z3D), flat(dn["east"]), flat(dn["north"])
    wet = (~torch.isnan(flat(gm.v3D))).to(torch.uint8)
    u, v = torch.empty_like(rho), torch.empty_like(rho)
    torch.cuda.synchronize(dev)

    def call():
        ctx.check(capi.lib().otmb_bolus_gm_velocity_dev(ctx.handle, rho.data_ptr(), z3d.data_ptr(), wet.data_ptr(), de.data_ptr(), dnn.data_ptr(),
                                                        nx, ny, nz, int(gm.gridtopology.kind), 600.0, 0.01, u.data_ptr(), v.data_ptr()))

    for _ in range(3):
        call()
    ctx.synchronize()
    ctx.timing_enable(True)
    for _ in range(steps):
        call()
    kt = ctx.timing_collect()
    ctx.timing_enable(False)
    ms = kt["gm_slopes+gm_dyad"][0] / kt["gm_slopes+gm_dyad"][1]
    G = nx * ny * nz
    alg = 33 * G + 16 * nx * ny
    ctx.close()
    return {"workload": f"bolus_GM_velocity on the {nx}x{ny}x{nz} grid (src/RediGM.jl:46-79; the reference's Redi/GM code never enters T)",
            "kernels": "gm_fused_kernel (κGM·S through LDS; OTMB_GM_FUSED=0: gm_slopes_kernel + gm_dyad_kernel)", "ms": ms, "cells_per_s": G / (ms * 1e-3), "algorithmic_bytes": alg,
            "achieved_gbs": alg / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "4 arrays in (ρ, Z3D, wet, 2-D distances), 2 out; the two κGM·S arrays between the two steps stay in LDS.  Bound by its "
                    "arithmetic (~25 Float64 divisions and a tanh per cell in dependent chains), not by its bytes.  The transportmatrix of "
                    "config 2 (3-D ρ) is this line's headline."}


def box_probe(dev):
    """How fast is THIS box's memory system for the plainest job there is?  A 2 GiB device-to-device copy (torch), after the timed region: a
    reference point beside the line's own kernel times (the boxes of the pool differ by up to 15 % on the fill pass, profiles/r03/README.md
    section 6).  Not part of the measurement."""
    import torch
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float64, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    return {"copy_2GiB_ms": ms, "copy_gbs_read_plus_write": 2 * n * 8 / (ms * 1e-3) / 1e9,
            "note": "torch device-to-device copy of 2 GiB after the timed region: this box's plain streaming rate.  (Round 3: the pool's boxes differ by up to 15 % on the "
                    "fill pass and this rate does not follow it -- 4.63 TB/s on a box with a 0.316 ms fill, 4.98 TB/s on one with 0.360 ms -- the pass is bound by "
                    "request latency, not by streaming bandwidth: profiles/r03/README.md 5b, 6.)"}


def rocm_smi_probe(index=0):
    """Clocks / power / temperature of the card, from a process that has NOT initialised the GPU (rocm-smi is an `env python3` script: started
    from a process whose GPU a HIP call or a profiler preload has initialised, every hop of that chain is an exec the GPU boxes refuse --
    profiles/r04 call 21, gpurun_out/.graft_exec_refused): only the launcher parent of bench.py calls this, right after the headline child."""
    try:
        r = subprocess.run(["rocm-smi", "-d", str(index), "--showclocks", "--showpower", "--showtemp", "--showmemuse", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
        j = json.loads(r.stdout.decode() or "{}")
        card = next(iter(j.values())) if j else {}
        keep = ("sclk", "mclk", "fclk", "socclk", "Power", "Temperature", "junction", "memory", "GPU Memory Allocated")
        return {k: v for k, v in card.items() if any(t.lower() in k.lower() for t in keep)}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:120]}


def traffic_for(workload, kernel, args, world=1):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json), valid only for the
    kernel sources they were measured on (kernel_source_sha16) and for the workload they were measured at."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if world != 1 or args.rho != "array" or tj.get("kernel_source_sha16") != kernel_source_hash():
            return None
        return tj.get("workloads", {}).get(workload, {}).get(kernel, {}).get("traffic_bytes")
    except (OSError, ValueError):
        return None


def under_profiler():
    """A rocprofv3 / rocprof preload has initialised the GPU before this script started: no child processes then (they would inherit the
    profiler, write their dispatches into the same output directory and mix two grids in one summary)."""
    return any(k.startswith(("ROCPROFILER_", "ROCPROF_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def main(slab_backend_factory=None, cpu_rehearsal=False):
    """slab_backend_factory / cpu_rehearsal: a seam for tests/bench_rehearsal.py ONLY (the multi-rank plumbing of this script on a box
    without GPUs, over gloo, with a backend the TEST side supplies).  Nothing in this file imports from tests/ or oracle/ except the
    cpu_baseline leg, no environment variable or flag selects another backend: run as `python bench.py` the product library computes."""
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    extras = {}
    if args.gpus == 1 and "RANK" not in os.environ and args.extra_configs and not cpu_rehearsal and not under_profiler():
        # Every configuration of the line is measured by a FRESH process that has the GPU to itself, the HEADLINE FIRST: a process that
        # starts after the 0.25 / 0.1 degree runs have allocated and freed 35 / 220 GB measures the same kernels 2-3 % slower
        # (profiles/r04/README.md section 7: 0.3775 against 0.3658 ms on one box; round 3 ran the headline last).  This process only
        # launches the others -- it never touches the GPU -- and prints the headline child's line with the extra records merged in.
        cmd = [sys.executable, os.path.abspath(sys.argv[0]), *[a for a in sys.argv[1:]], "--extra-configs", ""]
        head = None
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=float(os.environ.get("OTMB_BENCH_TIMEOUT", "1500")))
            line = next((l for l in r.stdout.decode().splitlines() if l.startswith("{")), None)
            if r.returncode == 0 and line is not None:
                head = json.loads(line)
        except (subprocess.TimeoutExpired, OSError, ValueError):
            head = None
        if head is not None:
            if isinstance(head.get("box_probe"), dict):
                head["box_probe"]["rocm_smi"] = rocm_smi_probe()  # this process never touches the GPU
            head.update(extra_configs_in_children(args))
            head["measured_by"] = "fresh processes, one per configuration, headline first: " + " ".join(cmd[1:])
            print(json.dumps(head), flush=True)
            return
        # (the headline child failed: measure it here, as rounds 1-3 did, so that the failure is visible in this process's output)
        extras = extra_configs_in_children(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    import otmb_amd
    from otmb_amd import synthetic
    from otmb_amd.device import DeviceAssembler

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # OTMB_FORCE_SLAB=1: take the depth-slab (distributed) code path even with one rank, e.g. under
    # `torchrun --nproc-per-node 1`, to exercise RCCL initialisation and collectives on a one-GPU box
    force_slab = os.environ.get("OTMB_FORCE_SLAB") == "1" and "RANK" in os.environ
    # rehearsal (tests/bench_rehearsal.py only): the multi-rank orchestration of this script on a box without GPUs -- gloo, a
    # backend supplied by the test -- the line it prints is labelled as not being a measurement
    rehearsal = bool(cpu_rehearsal)
    saved_stdout_fd = None
    backend = os.environ.get("OTMB_DIST_BACKEND", "gloo" if rehearsal else "nccl")
    if world > 1 or force_slab:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # OTMB_DIST_BACKEND=gloo + OTMB_SHARE_GPU=1: rehearsal of the multi-rank path on a one-GPU box
        if os.environ.get("OTMB_SHARE_GPU") == "1":
            local_rank = 0
        if not rehearsal:
            torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on the process's stdout when its first communicator comes up; stdout must carry
        # exactly one JSON line, so file descriptor 1 points at stderr until the result is printed
        sys.stdout.flush()
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cpu") if rehearsal else torch.device("cuda", local_rank)

    def make_backend(lr):
        if slab_backend_factory is not None:
            assert rehearsal, "a foreign slab backend is a rehearsal, never a measurement"
            return slab_backend_factory(lr)
        from otmb_amd import dist as odist_

        return odist_.HipSlabBackend(lr)

    # One arena for everything the run allocates: a block of the workload's whole footprint is reserved and handed back to
    # torch's caching allocator BEFORE anything else is allocated, so that every array of the run is carved out of one device
    # allocation (one contiguous range of addresses) instead of one hipMalloc per array (placement: profiles/r04/README.md section 7).
    arena_gb = float(os.environ.get("OTMB_BENCH_ARENA_GB", "0"))
    if arena_gb > 0 and not rehearsal:
        _arena = torch.empty(int(arena_gb * 2 ** 30), dtype=torch.uint8, device=dev)
        del _arena

    nx, ny, nz, lf = synthetic.PRESETS[args.workload]
    host_grid = None  # (g, gm) when the whole grid also exists on the host (cpu_baseline / end_to_end legs)
    placement = None  # DeviceAssembler.choose_placement's record (single-GPU, asynchronous protocol)
    if world > 1 or force_slab:
        from otmb_amd import dist as odist
        from otmb_amd import synthetic_device

        # weak: nz*world levels stretched over the same depth, per-GPU work fixed; strong: the workload's own grid
        nzg = nz * world if args.scaling == "weak" else nz
        counts = synthetic.level_wet_counts(nx, ny, nzg, seed=args.seed, land_fraction=lf)
        k0, k1 = odist.balanced_partition(counts, world)[rank]
        # every rank generates only its own levels, on its own device
        dg = synthetic_device.make_device_grid((nx, ny, nzg), dev, seed=args.seed, land_fraction=lf, rho=args.rho, k0=k0, k1=k1)
        local = odist.make_local_grid_from_device(dg)
        be = make_backend(local_rank)
        srun = odist.SlabRunner(be, odist.Comm(), local)
        umo, vmo, fill = dg.umo, dg.vmo, dg.fill

        class _Slab:
            n_wet_total = srun.n_global
            ctx = getattr(be, "ctx", None)
            pending = False
            slab = (k0, k1, nzg)

            def step(self):
                # always the asynchronous pipeline: only the facefluxes chain's planes couple the ranks (the synchronous
                # SlabRunner.step gathers nnz with two collectives per field: every rank would wait out the chain's skew every step)
                srun.step_async(umo, vmo, fill)
                self.pending = True

            def sync(self):
                if self.pending:
                    srun.finish()
                    self.pending = False
                be.sync()

            @property
            def nnz(self):
                return [int(x) for x in srun.nnz_global]

            def algorithmic_bytes(self):  # this rank's slab (rank 0 reports)
                n3d = 9 + (1 if be.rho is not None else 0)
                return 8 * be.G * n3d + 80 * nx * ny + 8 * be.nz + sum(16 * z + 8 * (be.n_own + 1) for z in be.nnz)

            def facefluxes_bytes(self):
                return be.nown_lev * be.P * (16 + 1 + 48)

        runner = _Slab()
    else:
        if args.workload in ("quarterdeg", "tenthdeg"):  # too large to build on the host and copy: generated on the device
            from otmb_amd import synthetic_device

            dg = synthetic_device.make_device_grid(args.workload, dev, seed=args.seed, rho=args.rho)
            asm = synthetic_device.assembler_for(dg, local_rank)
            umo, vmo, fill = dg.umo, dg.vmo, dg.fill
        else:
            g = synthetic.make_grid(nx, ny, nz, seed=args.seed, land_fraction=lf, rho=args.rho)
            gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                          lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
            host_grid = (g, gm)
            asm = DeviceAssembler(local_rank)
            asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=True)
            umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
            vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
            fill = g.umo.properties["_FillValue"]

        # (not under a profiler: the candidates' launches would enter the per-kernel averages of profiles/*_kernel_stats.csv and traffic.json)
        if args.placement_candidates > 1 and args.protocol == "async" and args.workload != "tenthdeg" and not under_profiler():
            try:  # set-up, outside the timed region; never results (tests/test_device_api.py)
                placement = asm.choose_placement(umo, vmo, fill, candidates=args.placement_candidates)
            except Exception as e:
                placement = {"error": f"{type(e).__name__}: {e}"[:200]}

        class _Single:
            protocol = "twophase" if (args.protocol == "twophase" or args.workload == "tenthdeg") else "async"
            n_wet_total = asm.N
            ctx = asm.ctx
            pending = False
            slab = (0, nz, nz)

            def step(self):
                if args.protocol == "async" and args.workload != "tenthdeg":
                    asm.step_async(umo, vmo, fill)  # no host round trip inside a step; errors surface in sync()
                    self.pending = True
                else:  # (the upper-bound output buffers of the asynchronous protocol do not fit at 0.1 degree)
                    asm.step(umo, vmo, fill, onepass=False)

            def sync(self):
                if self.pending:
                    asm.finish()
                    self.pending = False
                asm.ctx.synchronize()

            @property
            def nnz(self):
                return asm.nnz

            def algorithmic_bytes(self):
                return asm.algorithmic_bytes()

            def facefluxes_bytes(self):
                return asm.facefluxes_bytes()

            def bytes_split(self, fill_pass):  # (read, written) of the dominant kernel
                return asm.algorithmic_bytes_split() if fill_pass else (asm.G * 17, asm.G * 48)

            def stream_mix(self):
                return asm.fill_pass_stream_mix()

            def fused_step(self):  # the extension otmb_step_dev: only ϕtop stored (never `value`: the two-call path is the drop-in)
                asm.step_fused_async(umo, vmo, fill)

            fused_available = args.protocol == "async" and args.workload != "tenthdeg" and nx >= 3

        runner = _Single()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.step()
    runner.sync()
    # EXACTLY K steps between two barrier + synchronize brackets, max over ranks; repeated R times: the pool's boxes
    # differ by ~15 % and a single 10 ms region says little, so the median is reported with the spread beside it
    per_repeat = []
    for _ in range(args.repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            runner.step()
        runner.sync()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist.is_initialized():
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        per_repeat.append(elapsed)
    elapsed = float(np.median(per_repeat))

    # per-kernel durations with HIP events on the launch stream, over one more pass of the same K steps
    ktimes = {}
    if runner.ctx is not None:
        runner.ctx.timing_enable(True)
        for _ in range(args.steps):
            runner.step()
        runner.sync()
        ktimes = runner.ctx.timing_collect()
        runner.ctx.timing_enable(False)

    # The fused device-resident step (otmb_step_dev: facefluxes stores ϕtop only, the fill pass re-derives the other five fluxes from umo /
    # vmo; the same five matrices bit for bit, tests/test_fused_step.py), measured the same way right after the headline and reported as an
    # EXTRA record: `value` stays the two-call path, which is what the reference's API is (facefluxesfrommasstransport returns six arrays).
    fused = None
    if getattr(runner, "fused_available", False) and not rehearsal:
        try:
            for _ in range(args.warmup):
                runner.fused_step()
            runner.pending = True
            runner.sync()
            reps = []
            for _ in range(min(args.repeats, 3)):
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    runner.fused_step()
                runner.pending = True
                runner.sync()
                barrier()
                reps.append(time.perf_counter() - t0)
            runner.ctx.timing_enable(True)
            for _ in range(args.steps):
                runner.fused_step()
            runner.pending = True
            runner.sync()
            fk = runner.ctx.timing_collect()
            runner.ctx.timing_enable(False)
            fms = 1e3 * float(np.median(reps)) / args.steps
            fused = {"what": "otmb_step_dev (extension, NOT the headline): one call from (umo, vmo) to the five matrices, only ϕtop stored -- "
                             "64 bytes per cell less HBM traffic; the same matrices bit for bit",
                     "ms_per_step": fms, "wet_cells_per_s": runner.n_wet_total / (fms * 1e-3),
                     "kernels_ms": {k: round(v[0] / v[1], 5) for k, v in fk.items()}}
        except Exception as e:  # an extra record must never cost the headline
            fused = {"error": f"{type(e).__name__}: {e}"[:300]}

    # N > 1: BASELINE.json configs[3] (the fixed 0.25 degree grid cut into N depth slabs) beside the headline, unless it IS the headline
    config4 = None
    c4_workload = os.environ.get("OTMB_BENCH_CONFIG4_WORKLOAD", "quarterdeg")  # (tests rehearse this leg on a small grid)
    if world > 1 and args.extra_configs and not (args.workload == c4_workload and args.scaling == "strong") and \
            (not rehearsal or "OTMB_BENCH_CONFIG4_WORKLOAD" in os.environ):
        try:
            config4 = slab_config(c4_workload, args, world, rank, dev, local_rank, rehearsal, backend, make_backend)
        except Exception as e:  # every rank raises the same pipeline errors (dist.SlabRunner.finish), so nobody is left in a collective
            config4 = {"workload": c4_workload, "error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        n_total = runner.n_wet_total
        value = n_total * args.steps / elapsed
        kavg = {k: v[0] / v[1] for k, v in ktimes.items()}
        roof = None
        if kavg:
            dom = max(kavg, key=kavg.get)
            bytes_alg = runner.algorithmic_bytes() if dom.startswith(("tm_kernel", "dm_fill")) else runner.facefluxes_bytes()
            achieved = bytes_alg / (kavg[dom] * 1e-3) / 1e9
            traffic = traffic_for(args.workload, dom, args, world)
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": bytes_alg,
                    "avg_kernel_ms": kavg[dom]}
        if roof is not None and world == 1 and not force_slab and not rehearsal and hasattr(runner, "bytes_split"):
            # The same kernel time as a fraction of what THIS box sustains: one plain read stream and one plain non-temporal write stream
            # (otmb_ctx_box_ceilings, measured here, in this process, right after the timed region), combined for the kernel's own read : write
            # mix -- bytes / (reads / read rate + writes / write rate).  The pool's boxes differ by up to 15 % on the same binary.
            try:
                rd, wr = runner.ctx.box_ceilings()
                br, bw = runner.bytes_split(dom.startswith(("tm_kernel", "dm_fill")))
                mix = (br + bw) / (br / rd + bw / wr)
                roof["box"] = {"read_gbs": rd, "write_gbs": wr, "mix_ceiling": mix, "frac_of_box": achieved / mix,
                               "bytes_read": br, "bytes_written": bw}
                if dom.startswith("tm_kernel") and hasattr(runner, "stream_mix"):
                    # ... and of what an ideal streaming kernel reaches over the fill pass's OWN arrays, where they lie (destroys the matrices:
                    # everything that reads them has run).  The two plain streams above are the same on fast and slow boxes (profiles/r05).
                    sm = runner.stream_mix()  # {columns per slice: GB/s}; 256 = the fill pass's own tile
                    roof["box"]["stream_mix_gbs"] = {str(k): v for k, v in sm.items()}
                    roof["box"]["frac_of_mix"] = achieved / max(sm.values())  # against the BEST plain stream over these arrays
            except Exception as e:  # a diagnostic must never cost the line
                roof["box"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        k0, k1, nzg = runner.slab
        if world > 1:
            shape = (f"{nx}x{ny}x{nzg} cut into {world} depth slabs (rank 0: levels [{k0},{k1}))"
                     + (f"; weak scaling: {nz} levels per rank" if args.scaling == "weak" else "; strong scaling: fixed grid"))
        else:
            shape = f"{nx}x{ny}x{nz}"
        out = {
            "metric": "wet-cells/s assembled into T" if not rehearsal else "REHEARSAL of the multi-rank orchestration (CPU checker backend): not a measurement",
            "value": value, "unit": "wet-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: synthetic ACCESS-ESM1-5-like tripolar grid {shape}"
                            f", facefluxes + full transportmatrix (5 CSC matrices), rho={args.rho}, upwind",
                "wet_cells": n_total, "nnz": dict(zip(("T", "Tadv", "TkH", "TkVML", "TkVdeep"), runner.nnz)),
                "seed": args.seed, "protocol": getattr(runner, "protocol", args.protocol),
            },
            "repeats": {"n": args.repeats, "ms_per_step_median": ms_step, "ms_per_step_min": 1e3 * min(per_repeat) / args.steps,
                        "ms_per_step_max": 1e3 * max(per_repeat) / args.steps},
            "roofline": roof,
            "kernels_ms": {k: round(v, 5) for k, v in kavg.items()},
            "step_gbs": None if rehearsal else (runner.algorithmic_bytes() + runner.facefluxes_bytes()) / (ms_step * 1e-3) / 1e9,
        }
        if fused is not None:
            out["fused_step"] = fused
        if world == 1 and not force_slab and not rehearsal:
            out["placement"] = placement
        if world > 1 or force_slab:
            out["config"]["ranks_over"] = backend  # the depth-slab path: "nccl" = RCCL (one rank per GPU)
        if world == 1 and host_grid is not None and not rehearsal:
            try:  # (an extra record must never cost the headline: the host-pointer legs allocate gigabytes of pinned memory and start threads)
                out["end_to_end"] = None if args.no_end_to_end else end_to_end(*host_grid, n_total)
            except Exception as e:
                out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            out["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline(*host_grid, args.workload)
        if world == 1 and host_grid is not None and not rehearsal and dev.type == "cuda":
            try:  # an extra record must never cost the headline
                out["config2"] = config2_bolus(*host_grid, dev, local_rank, min(args.steps, 10))
            except Exception as e:
                out["config2"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not rehearsal and dev.type == "cuda":
            try:  # after the timed region and not part of the measurement: it must never cost the line
                torch.cuda.empty_cache()
                out["box_probe"] = box_probe(dev)
            except Exception as e:
                out["box_probe"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        if config4 is not None:
            out["config4"] = config4
        out.update(extras)  # BASELINE.json configs[2] ("HBM-roofline run") and the grid of configs[4] on this one GPU
        sys.stdout.flush()
        if saved_stdout_fd is not None:
            os.dup2(saved_stdout_fd, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout_fd is not None:
            os.dup2(2, 1)  # anything the communicator says while shutting down goes to stderr too
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
