"""Worker for the multi-process slab tests (spawned by test_dist_*.py): one rank of a depth-slab run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(rank, world, port, backend_kind, case, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    rccl = os.environ.get("OTMB_TEST_RCCL") == "1"  # one GPU per rank, planes over RCCL (xGMI): needs world GPUs on the box
    if rccl:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import otmb_amd
    from otmb_amd import dist as od, synthetic

    nx, ny, nz, seed, rho, topo = case
    counts = synthetic.level_wet_counts(nx, ny, nz, seed=seed, topology=topo)
    k0, k1 = od.balanced_partition(counts, world)[rank]
    g = synthetic.make_slab(nx, ny, nz, k0, k1, seed=seed, rho=rho, topology=topo)
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev[k0:k1],
                                  lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    local = od.make_local_grid(gm, g.mlotst, g.rho, k0, k1, nz, g.lev, upwind=os.environ.get("OTMB_TEST_CENTRED") != "1")
    if backend_kind == "hip":
        be = od.HipSlabBackend(rank if rccl else 0)
    else:
        from slab_checker_backend import OracleSlabBackend

        be = OracleSlabBackend()
    record_kernels = backend_kind == "hip" and os.environ.get("OTMB_TEST_KERNELS") == "1"
    if record_kernels:  # which kernels this rank launched (tests of the counts-in-facefluxes slab path)
        be.ctx.timing_enable(True)
    comm = od.Comm()
    runner = od.SlabRunner(be, comm, local)
    dev = be.device
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).to(dev)
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).to(dev)
    fault = os.environ.get("OTMB_TEST_FAULT")  # "rank:step:kind" -- a failure injected into one step of an asynchronous pipeline
    if fault:
        from otmb_amd.capi import OtmbError

        frank, fstep, kind = fault.split(":")
        frank, fstep = int(frank), int(fstep)
        lw3 = be.s["lwet3d"]
        lw = lw3.cpu().numpy().ravel() if torch.is_tensor(lw3) else np.asfortranarray(lw3).ravel(order="F")
        P = nx * ny
        own = np.flatnonzero(lw[be.s["k_own0"] * P:be.s["k_own1"] * P]) + be.s["k_own0"] * P
        L = int(own[len(own) // 2])
        caught = None
        try:
            for sidx in range(5):
                u, restore = umo, None
                if sidx == fstep and kind == "rho" and rank == frank:  # NaN in ρ on ONE wet cell of ONE slab, this step only
                    if backend_kind == "hip":
                        old = be.rho[L].clone()
                        be.rho[L] = float("nan")
                        restore = lambda: be.rho.__setitem__(L, old)
                    else:
                        flat = be.s["rho"].reshape(-1, order="F")
                        assert np.shares_memory(flat, be.s["rho"])
                        old = float(flat[L])
                        flat[L] = np.nan
                        restore = lambda: flat.__setitem__(L, old)
                if sidx == fstep and kind == "missing":  # no valid umo anywhere (every rank): velocities.jl:199
                    u = torch.full_like(umo, float("nan"))
                runner.step_async(u, vmo, 1e20)
                if restore:
                    restore()
            runner.finish()
        except (OtmbError, AssertionError) as e:
            caught = e
        with open(os.path.join(outdir, f"fault_{rank}.txt"), "w") as f:
            f.write("none" if caught is None else f"{type(caught).__name__}|{getattr(caught, 'step', None)}|{caught}")
        # the pipeline is usable again after the failure
        runner.step_async(umo, vmo, 1e20)
        out = runner.finish()
    elif os.environ.get("OTMB_TEST_ASYNC") == "1":
        for _ in range(3):  # a stream of fields with no collective in between
            runner.step_async(umo, vmo, 1e20)
        out = runner.finish()
    else:
        for _ in range(2):  # twice: buffers are reused between fields
            out = runner.step(umo, vmo, 1e20)
    runner.sync()
    if record_kernels:
        import json

        with open(os.path.join(outdir, f"kernels_{rank}.json"), "w") as f:
            json.dump({k: v[1] for k, v in be.ctx.timing_collect().items()}, f)
    host = be.result_to_host() if backend_kind == "hip" else out
    glob = od.gather_global_csc(comm, host, runner.n_own, be.nnz, dev)
    if rank == 0:
        np.savez(os.path.join(outdir, "global.npz"), n=runner.n_global, nnz=runner.nnz_global,
                 **{f"{k}_{q}": glob[m][k] for q, m in enumerate(od.MATS) for k in range(3)})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    rank, world, port, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    nx, ny, nz, seed = (int(x) for x in sys.argv[5:9])
    run(rank, world, port, kind, (nx, ny, nz, seed, sys.argv[9], sys.argv[10]), sys.argv[11])
