#!/usr/bin/env python3
"""bench.py -- wet-cells/s assembled into T on the synthetic ACCESS-ESM1-5-like 1° grid.

One "step" = one pass of the hot path over one (umo, vmo) field, all device resident:
    facefluxesfrommasstransport -> transportmatrix (T, Tadv, TκH, TκVML, TκVdeep in CSC)
Inputs are in HBM before the timed region starts and the five CSC matrices are left in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload access1deg|quarterdeg|small]

N > 1 (launched by torch.distributed.run, one rank per GPU): the grid is partitioned in depth,
see otmb_amd/dist.py; value = wet cells of all ranks / max-over-ranks time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="access1deg")
    ap.add_argument("--rho", default="array", choices=["array", "scalar"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=20260501)
    ap.add_argument("--protocol", default="async", choices=["async", "twophase"],
                    help="async: otmb_transportmatrix_dev (count -> scan -> fill enqueued back to back, outputs preallocated "
                         "at their upper bound); twophase: plan (host learns nnz) then fill, as a caller that sizes its outputs does")
    args = ap.parse_args()
    args.steps = max(1, args.steps)
    args.warmup = max(0, args.warmup)
    return args


def cpu_baseline(g, gm, workload, reps=5):
    """The oracle (single-thread C restatement of the reference algorithm: push COO -> sparse() x4 ->
    3 sparse adds) timed on this box's host cores over the same workload: 1 warm-up + median of `reps`
    passes.  kind = "port": the Julia reference cannot run here (no julia binary; SURVEY.md
    section 8c)."""
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


def main():
    args = parse()
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
    saved_stdout_fd = None
    if world > 1 or force_slab:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # OTMB_DIST_BACKEND=gloo + OTMB_SHARE_GPU=1: rehearsal of the multi-rank path on a one-GPU box
        backend = os.environ.get("OTMB_DIST_BACKEND", "nccl")
        if os.environ.get("OTMB_SHARE_GPU") == "1":
            local_rank = 0
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
    dev = torch.device("cuda", local_rank)

    nx, ny, nz, lf = synthetic.PRESETS[args.workload]
    if world > 1 or force_slab:
        # weak scaling: the global grid is nx x ny x (nz*world) (levels stretched over the same depth), cut
        # into `world` depth slabs with balanced wet counts; every rank generates only its own levels
        from otmb_amd import dist as odist

        nzg = nz * world
        counts = synthetic.level_wet_counts(nx, ny, nzg, seed=args.seed, land_fraction=lf)
        k0, k1 = odist.balanced_partition(counts, world)[rank]
        g = synthetic.make_slab(nx, ny, nzg, k0, k1, seed=args.seed, land_fraction=lf, rho=args.rho)
        gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat,
                                      lev=g.lev[k0:k1], lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
        local = odist.make_local_grid(gm, g.mlotst, g.rho, k0, k1, nzg, g.lev,
                                      kappa=(g.kappaH, g.kappaVML, g.kappaVdeep), upwind=True)
        be = odist.HipSlabBackend(local_rank)
        srun = odist.SlabRunner(be, odist.Comm(), local)
        umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
        vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
        fill = g.umo.properties["_FillValue"]

        class _Slab:
            n_wet_total = srun.n_global
            ctx = be.ctx

            pending = False

            def step(self):
                if args.protocol == "async":
                    srun.step_async(umo, vmo, fill)  # only the facefluxes chain's planes couple the ranks
                    self.pending = True
                else:
                    srun.step(umo, vmo, fill)

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
        g = synthetic.make_grid(nx, ny, nz, seed=args.seed, land_fraction=lf, rho=args.rho)
        gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                      lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
        asm = DeviceAssembler(local_rank)
        asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=True)
        umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
        vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
        fill = g.umo.properties["_FillValue"]

        class _Single:
            n_wet_total = asm.N
            ctx = asm.ctx

            pending = False

            def step(self):
                if args.protocol == "async":
                    asm.step_async(umo, vmo, fill)  # no host round trip inside a step; errors surface in sync()
                    self.pending = True
                else:
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

        runner = _Single()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.step()
    runner.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step()
    runner.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations with HIP events on the launch stream, over a second pass of the same K steps
    runner.ctx.timing_enable(True)
    for _ in range(args.steps):
        runner.step()
    runner.sync()
    ktimes = runner.ctx.timing_collect()
    runner.ctx.timing_enable(False)

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        n_total = runner.n_wet_total
        value = n_total * args.steps / elapsed
        kavg = {k: v[0] / v[1] for k, v in ktimes.items()}
        dom = max(kavg, key=kavg.get)
        bytes_alg = runner.algorithmic_bytes() if dom.startswith("tm_kernel") else runner.facefluxes_bytes()
        achieved = bytes_alg / (kavg[dom] * 1e-3) / 1e9
        # HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes of this same
        # command (profiles/traffic.json; bytes do not depend on the box, unlike times)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("workload") == args.workload and world == 1 and args.rho == "array":
                traffic = tj["kernels"].get(dom, {}).get("traffic_bytes")
        except (OSError, ValueError):
            pass
        out = {
            "metric": "wet-cells/s assembled into T", "value": value, "unit": "wet-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: synthetic ACCESS-ESM1-5-like tripolar grid {nx}x{ny}x{nz}"
                            + (f" per rank, stacked in depth x{world}" if world > 1 else "")
                            + f", facefluxes + full transportmatrix (5 CSC matrices), rho={args.rho}, upwind",
                "wet_cells": n_total, "nnz": dict(zip(("T", "Tadv", "TkH", "TkVML", "TkVdeep"), runner.nnz)),
                "seed": args.seed,
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_alg, "avg_kernel_ms": kavg[dom],
            },
            "kernels_ms": {k: round(v, 5) for k, v in kavg.items()},
            "step_gbs": (runner.algorithmic_bytes() + runner.facefluxes_bytes()) / (ms_step * 1e-3) / 1e9,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(g, gm, args.workload)
        elif world == 1:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        if saved_stdout_fd is not None:
            os.dup2(saved_stdout_fd, 1)
        print(json.dumps(out), flush=True)
        if saved_stdout_fd is not None:
            os.dup2(2, 1)  # anything the communicator says while shutting down goes to stderr too
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
