"""bench.py's launcher and multi-rank plumbing on a box without GPUs: `python bench.py --gpus N` (no torchrun) must spawn
its N ranks itself and print exactly ONE JSON line.  The ranks run the depth-slab orchestration over gloo with the CPU
checker backend of tests/, injected by tests/bench_rehearsal.py (bench.py itself cannot select it) -- the line says that it is a
rehearsal, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gpus,scaling", [(2, "weak"), (3, "strong")])
def test_bench_spawns_its_ranks_and_prints_one_json_line(gpus, scaling):
    env = dict(os.environ, OTMB_BENCH_CONFIG4_WORKLOAD="small")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--gpus", str(gpus), "--workload", "small", "--scaling", scaling,
                        "--steps", "2", "--warmup", "1", "--repeats", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["scaling"] == scaling and d["steps"] == 2 and d["unit"] == "wet-cells/s"
    assert "REHEARSAL" in d["metric"] and d["roofline"] is None
    assert f"cut into {gpus} depth slabs" in d["config"]["workload"]
    nz = 10 * gpus if scaling == "weak" else 10
    assert f"36x30x{nz}" in d["config"]["workload"]
    # the line's own evidence about its communicator: every rank answered an all_reduce, who carried it, and what the one
    # exchange of the path -- a plane of ϕtop to the slab above -- costs here, whole and in row bands
    assert d["config"]["ranks_seen"] == gpus and d["config"]["ranks_over"] == "gloo"
    cm = d["comm"]
    assert cm["backend"] == "gloo" and cm["world"] == gpus and cm["library"].startswith("gloo")
    ho = cm["plane_hand_off"]
    assert ho["bytes"] == 8 * 36 * 30 and ho["payload_intact"] is True and ho["ms_by_pieces"]["1"] > 0 and ho["ms_per_piece"] > 0
    assert isinstance(d["kernels_ms_over_ranks"], dict)  # (the CPU checker backend launches no kernels: empty here, min / max per kernel on GPUs)
    # the strong-scaled sub-record (BASELINE.json configs[3]; here on the small grid) rides along unless it is the headline itself
    if scaling == "weak":
        c4 = d["config4"]
        assert c4.get("error") is None and c4["scaling"] == "strong" and c4["n_gpus"] == gpus and c4["grid"] == "36x30x10", c4
        assert c4["wet_cells"] > 0 and c4["ms_per_step"] > 0
    else:
        assert "config4" not in d


def test_a_rank_that_never_arrives_ends_the_run_with_its_name():
    """First contact with a node must end with a verdict, not a hang: a rank that never joins the first collective trips the other
    ranks' watchdog (OTMB_BENCH_PREFLIGHT_S), which names the absent rank and exits non-zero; the launcher stops everybody."""
    import time

    env = dict(os.environ, OTMB_BENCH_TEST_STALL_RANK="1", OTMB_BENCH_PREFLIGHT_S="6")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_rehearsal.py"), "--gpus", "2", "--workload", "small", "--steps", "2",
                        "--warmup", "1", "--repeats", "1", "--extra-configs", ""], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t0 < 120
    assert "did not finish 'preflight: all_reduce of ones' within 6 s" in r.stderr and "ranks that never reached this phase: [1]" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_more_ranks_than_gpus_is_said_in_one_line_and_ends_the_run():
    """First contact with a node that shows fewer GPUs than `--gpus N` asks for (here: none) -- the likeliest first failure of a multi-GPU
    run: every rank that has no device says so in one line and exits 3 before any GPU or communicator call, the launcher names the rank and
    returns 3 within seconds, nothing is printed on stdout (profiles/r06/call19_*: the same on a one-GPU box)."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ""
    assert "rank 1 (local rank 1): this node shows" in r.stderr and "--gpus 2 needs 2 or more" in r.stderr
    assert "ended with status 3; the other ranks were stopped" in r.stderr


def test_traffic_json_is_keyed_to_the_kernel_sources():
    """roofline.traffic comes from committed PMC passes: it must name the kernel sources it was measured on."""
    sys.path.insert(0, ROOT)
    import importlib

    bench = importlib.import_module("bench")
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert "kernel_source_sha16" in tj and len(tj["kernel_source_sha16"]) == 16
    assert len(bench.kernel_source_hash()) == 16


def test_bench_py_cannot_reach_the_checker():
    """nothing in bench.py imports from tests/, and the oracle is imported by the cpu_baseline leg alone."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "slab_checker_backend" not in src and "OTMB_BENCH_CHECKER_BACKEND" not in src
    assert '"tests"' not in src and "'tests'" not in src
    head, tail = src.split("def cpu_baseline", 1)
    rest = tail.split("\ndef ", 1)[1]  # everything after the cpu_baseline leg
    for part in (head, rest):
        assert "from oracle" not in part and "import oracle" not in part
