"""GPU: bench.py itself, as the driver starts it (run with -m gpu).  (1) `python bench.py` on the small workload prints exactly one JSON
line with the contract's keys, a live roofline record and the product library's kernels; (2) the launcher form -- `python -m
torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1` with OTMB_FORCE_SLAB=1 -- takes the depth-slab path with ONE rank over
the "nccl" backend (= RCCL): communicator bring-up with device_id, barrier and the max-over-ranks all_reduce of the timing on this image
and this GPU.  A box with one GPU cannot run two RCCL ranks (tests/test_dist_gpu.py skips those); this is the part of that path it can run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--workload", "small", "--steps", "3", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline", "--no-end-to-end", "--extra-configs="]
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline")


def _one_json_line(r):
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OTMB_FORCE_SLAB")}
    env.update(extra)
    return env


def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON], env=_clean_env(), capture_output=True, text=True, timeout=600)
    d = _one_json_line(r)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "wet-cells/s" and d["dtype"] == "f64"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and "synthetic" in d["data"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert abs(d["value"] - d["config"]["wet_cells"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # the kernels of the product library ran (HIP events on the launch stream), the fill pass among them
    assert d["kernels_ms"]["tm_kernel<fill>"] > 0 and d["kernels_ms"]["facefluxes_kernel"] > 0


def test_bench_as_one_rank_over_rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", *COMMON]
    r = subprocess.run(cmd, env=_clean_env(OTMB_FORCE_SLAB="1", HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=900)
    d = _one_json_line(r)
    assert d["n_gpus"] == 1 and d["unit"] == "wet-cells/s" and d["value"] > 0
    assert d["config"].get("ranks_over") == "nccl", d["config"]  # the depth-slab path, its rank over RCCL
    assert d["roofline"] is not None and 0 < d["roofline"]["frac"] < 1
