"""GPU: the depth-slab runner with all ranks as THREADS of this process (one context each on GPU 0), talking through an in-process
stand-in for otmb_amd.dist.Comm -- the same SlabRunner / HipSlabBackend code as under torchrun, without a process group, so that many
shapes, cuts, row bands and wave geometries fit in one test run.  Every case is compared with the whole-grid oracle bit for bit, and the
kernels every rank launched are recorded: a slab that can count in facefluxes launches no counting pass and no push-mask kernel."""
import queue
import threading

import numpy as np
import pytest

import otmb_amd
from helpers import COUNTS_ON, MATS, assert_csc_equal
from otmb_amd import dist as od, synthetic

pytestmark = pytest.mark.gpu
TIMEOUT = 120


class ThreadWorld:
    def __init__(self, world):
        self.world = world
        self.box = {(s, d): queue.Queue() for s in range(world) for d in range(world)}
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadComm:
    """The interface of otmb_amd.dist.Comm between threads.  Every rank's kernels go to the same (default) stream of the one GPU, so a
    tensor put into a mailbox after its producing kernel was enqueued is complete for whoever copies from it later."""
    backend = "threads"

    def __init__(self, tw, rank):
        self.tw, self.rank, self.world = tw, rank, tw.world

    def send(self, t, dst):
        self.tw.box[(self.rank, dst)].put(t.clone())

    def isend(self, t, dst):
        self.send(t, dst)
        return None

    @staticmethod
    def wait_send(handle):
        pass

    def recv(self, t, src):
        t.copy_(self.tw.box[(src, self.rank)].get(timeout=TIMEOUT))

    def irecv(self, t, src):
        return (src, t)

    def wait_recv(self, handle):
        if handle is not None:
            self.recv(handle[1], handle[0])

    def exchange(self, sends, recvs):
        for t, dst in sends:
            self.send(t, dst)
        for t, src in recvs:
            self.recv(t, src)

    def _gather(self, val):
        self.tw.slots[self.rank] = val
        self.tw.barrier.wait(timeout=TIMEOUT)
        out = list(self.tw.slots)
        self.tw.barrier.wait(timeout=TIMEOUT)
        return out

    def allreduce_sum_i64(self, arr, device):
        return np.sum(self._gather(np.asarray(arr, dtype=np.int64)), axis=0)

    def allgather_i64(self, vals, device):
        return np.stack(self._gather(np.asarray(vals, dtype=np.int64)))


def run_threads(case, world, pieces, async_mode, upwind=True, steps=2, expect_error=None):
    import torch

    nx, ny, nz, seed, rho, topo = case
    tw = ThreadWorld(world)
    results, errors = [None] * world, []
    counts = synthetic.level_wet_counts(nx, ny, nz, seed=seed, topology=topo)
    parts = od.balanced_partition(counts, world)

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            k0, k1 = parts[rank]
            g = synthetic.make_slab(nx, ny, nz, k0, k1, seed=seed, rho=rho, topology=topo)
            gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev[k0:k1],
                                          lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
            local = od.make_local_grid(gm, g.mlotst, g.rho, k0, k1, nz, g.lev, upwind=upwind)
            be = od.HipSlabBackend(0)
            be.ctx.timing_enable(True)
            runner = od.SlabRunner(be, ThreadComm(tw, rank), local, chain_pieces=pieces)
            umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
            vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
            if async_mode:
                for _ in range(steps):
                    runner.step_async(umo, vmo, 1e20)
                runner.finish()
            else:
                for _ in range(steps):
                    runner.step(umo, vmo, 1e20)
            runner.sync()
            kernels = {k: v[1] for k, v in be.ctx.timing_collect().items()}
            host = be.result_to_host()
            results[rank] = (host, runner.n_own, list(be.nnz), kernels, runner.n_global)
        except BaseException as e:  # noqa: BLE001 -- reported by the test's own thread
            errors.append((rank, e))
            tw.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(TIMEOUT)
    if expect_error is not None:  # the reference raises on this grid: so must the slab that holds the offending cells, and nobody may hang
        texts = [str(e) for _, e in errors]
        assert any(expect_error in s for s in texts), (expect_error, texts)
        return None, None, None
    assert not errors, errors
    assert all(r is not None for r in results), "a rank did not finish"
    glob = {}
    for q, m in enumerate(MATS):
        cps = [np.asarray(r[0][m][0][: r[1]]) for r in results]
        last = results[-1]
        glob[m] = (np.concatenate(cps + [np.asarray(last[0][m][0][last[1]:last[1] + 1])]),
                   np.concatenate([r[0][m][1][: r[2][q]] for r in results]), np.concatenate([r[0][m][2][: r[2][q]] for r in results]))
    return glob, [r[3] for r in results if r[1] > 0], results[0][4]  # (kernels: of the ranks that own a wet cell)


def whole_grid(oracle, case, upwind):
    nx, ny, nz, seed, rho, topo = case
    g = synthetic.make_slab(nx, ny, nz, 0, nz, seed=seed, rho=rho, topology=topo)
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                  lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    idx = oracle.makeindices(gm.v3D)
    phi = oracle.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], 1e20, gm.gridtopology.kind)
    return idx, oracle.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)


def _cases():
    rng = np.random.default_rng(20261004)
    out = []
    for n in range(40):
        nx = int(rng.choice([3, 5, 24, 64, 65, 70, 130, 200]))
        ny = int(rng.integers(2, 14))
        nz = int(rng.integers(3, 24))
        world = int(rng.integers(1, min(nz, 6) + 1))
        rows = int(rng.choice([1, 4]))
        pieces = int(rng.choice([1, 1, 2, 3, ny]))
        out.append(pytest.param((nx, ny, nz, 500 + n, str(rng.choice(["array", "scalar"])), str(rng.choice(["tripolar", "bipolar"]))), world,
                                 min(pieces, ny), rows, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)),
                                 id=f"{nx}x{ny}x{nz}-w{world}-p{min(pieces, ny)}-r{rows}"))
    return out


@pytest.mark.parametrize("case,world,pieces,rows,async_mode,upwind", _cases())
def test_random_cuts_of_random_grids(oracle, monkeypatch, case, world, pieces, rows, async_mode, upwind):
    monkeypatch.setenv("OTMB_FF_ROWS", str(rows))
    try:
        idx, ref = whole_grid(oracle, case, upwind)
    except Exception as e:  # e.g. "TκH contains NaNs." on a tripolar grid with an odd nx: the middle cell of the seam row is its own neighbour
        run_threads(case, world, pieces, async_mode, upwind, expect_error=str(e))
        return
    glob, kernels, n_global = run_threads(case, world, pieces, async_mode, upwind)
    assert n_global == idx["N"]
    for m in MATS:
        assert_csc_equal(glob[m], ref[m], m)
    can_count = COUNTS_ON and case[0] >= 3 and (pieces == 1 or rows == 4)
    for k in kernels:
        assert ("tm_count_kernel" not in k) == can_count, (k, can_count)
        if can_count:
            assert "push_mask_kernel" not in k, k


def test_one_level_slabs_with_halos_on_both_sides(oracle):
    """Every rank owns ONE level: its first level is its last, the cell above and the cell below are both halo cells."""
    case = (24, 7, 4, 601, "array", "tripolar")
    glob, kernels, _ = run_threads(case, 4, 1, True)
    _, ref = whole_grid(oracle, case, True)
    for m in MATS:
        assert_csc_equal(glob[m], ref[m], m)
    assert all("tm_count_kernel" not in k for k in kernels) or not COUNTS_ON, kernels


def test_access1deg_in_three_slabs_equals_the_single_gpu_path():
    """BASELINE.json configs[1]'s grid (360x300x50, one-row wave geometry as the library chooses it, tiles of 256 columns crossing wave
    segments and slab boundaries) cut into three depth slabs that count in facefluxes, against the single-GPU path's matrices."""
    import torch

    from otmb_amd.device import DeviceAssembler

    nx, ny, nz, seed = 360, 300, 50, 20260501
    case = (nx, ny, nz, seed, "array", "tripolar")
    glob, kernels, n_global = run_threads(case, 3, 1, True, steps=1)
    assert all("tm_count_kernel" not in k and "push_mask_kernel" not in k for k in kernels) or not COUNTS_ON, kernels
    g = synthetic.make_slab(nx, ny, nz, 0, nz, seed=seed, rho="array", topology="tripolar")
    gm = otmb_amd.makegridmetrics(areacello=g.areacello, volcello=g.volcello, lon=g.lon, lat=g.lat, lev=g.lev,
                                  lon_vertices=g.lon_vertices, lat_vertices=g.lat_vertices)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep, upwind=True)
    assert asm.N == n_global
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    asm.step(umo, vmo, 1e20)
    ref = asm.result_to_host()
    for m in MATS:
        for a, b, what in zip(glob[m], ref[m], ("colptr", "rowval", "nzval")):
            assert np.array_equal(np.asarray(a), np.asarray(b)), (m, what)


def test_quarterdeg_in_two_slabs_equals_the_single_gpu_path():
    """BASELINE.json configs[3]'s grid (1440x1080x75) as `bench.py --gpus 2 --scaling strong` runs it: the four-row wave geometry and the
    chain in four row bands as the library chooses them, both slabs counting in facefluxes -- against the single-GPU path's matrices,
    compared on the device."""
    import torch

    from otmb_amd import synthetic_device

    free_b, _ = torch.cuda.mem_get_info(0)
    if free_b < 150 * 2 ** 30:
        pytest.skip("needs ~130 GB of device memory for the whole grid and its two slabs side by side")
    dev = torch.device("cuda", 0)
    nx, ny, nz, lf = synthetic.PRESETS["quarterdeg"]
    counts = synthetic.level_wet_counts(nx, ny, nz, seed=20260501, land_fraction=lf)
    parts = od.balanced_partition(counts, 2)
    tw = ThreadWorld(2)
    res, errors = [None, None], []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            k0, k1 = parts[rank]
            dg = synthetic_device.make_device_grid("quarterdeg", dev, k0=k0, k1=k1)
            be = od.HipSlabBackend(0)
            be.ctx.timing_enable(True)
            runner = od.SlabRunner(be, ThreadComm(tw, rank), od.make_local_grid_from_device(dg))
            assert runner.chain_pieces == 4
            for _ in range(2):
                runner.step_async(dg.umo, dg.vmo, dg.fill)
            runner.finish()
            runner.sync()
            res[rank] = (be, runner, {k: v[1] for k, v in be.ctx.timing_collect().items()})
        except BaseException as e:  # noqa: BLE001
            errors.append((rank, e))
            tw.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors, errors
    for _, _, k in res:
        assert (("tm_count_kernel" not in k and "push_mask_kernel" not in k) or not COUNTS_ON) and k.get("facefluxes_kernel", 0) == 8, k  # 2 fields x 4 row bands
    dg = synthetic_device.make_device_grid("quarterdeg", dev)
    asm = synthetic_device.assembler_for(dg, 0)
    asm.step_async(dg.umo, dg.vmo, dg.fill)
    asm.finish()
    asm.ctx.synchronize()
    assert asm.N == res[0][1].n_global == res[0][1].n_own + res[1][1].n_own
    col0 = 0
    for be, runner, _ in res:
        n = runner.n_own
        for q, m in enumerate(MATS):
            cp, rv, nzv = be.out[m]
            rcp, rrv, rnz = asm.out[m]
            lo, hi = int(rcp[col0]) - 1, int(rcp[col0 + n]) - 1
            assert hi - lo == be.nnz[q], (m, hi - lo, be.nnz[q])
            assert torch.equal(cp[: n + 1], rcp[col0:col0 + n + 1]), (m, "colptr")
            assert torch.equal(rv[: be.nnz[q]], rrv[lo:hi]), (m, "rowval")
            assert torch.equal(nzv[: be.nnz[q]], rnz[lo:hi]), (m, "nzval")
        col0 += n
