"""GPU: the depth-slab path with the HIP backend.  Two ranks share the box's single GPU and talk over
gloo (RCCL needs one GPU per rank; the transport layer is the only difference -- see otmb_amd.dist.Comm),
and the concatenated result must equal the whole-grid oracle bit for bit."""
import pytest

from helpers import COUNTS_ON
from test_dist_cpu import check_against_whole_grid, run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,rho", [(1, "array"), (2, "array"), (3, "scalar")])
def test_hip_slabs_match_whole_grid_oracle(oracle, tmp_path, world, rho):
    case = (24, 18, 11, 33, rho, "tripolar")
    z = run_ranks(world, "hip", case, tmp_path)
    check_against_whole_grid(oracle, z, case)


def test_hip_slabs_async_pipeline(oracle, tmp_path):
    case = (24, 18, 11, 34, "array", "tripolar")
    z = run_ranks(3, "hip", case, tmp_path, async_mode=True)
    check_against_whole_grid(oracle, z, case)


@pytest.mark.parametrize("pieces,world,async_mode", [(2, 2, False), (5, 3, True), (18, 3, False)])
def test_hip_slabs_chain_in_row_bands(oracle, tmp_path, pieces, world, async_mode):
    """The facefluxes chain handed over in row bands (otmb_facefluxes_rows_dev per piece): bit-identical to the whole-grid oracle."""
    case = (24, 18, 11, 35, "array", "tripolar")
    z = run_ranks(world, "hip", case, tmp_path, async_mode=async_mode, pieces=pieces)
    check_against_whole_grid(oracle, z, case)


def _kernels(outdir, world):
    import json
    import os

    return [json.load(open(os.path.join(outdir, f"kernels_{r}.json"))) for r in range(world)]


@pytest.mark.parametrize("world,async_mode,case", [(1, True, (24, 18, 11, 37, "array", "tripolar")), (2, False, (70, 9, 12, 38, "array", "tripolar")),
                                                   (3, True, (24, 18, 11, 39, "scalar", "tripolar")), (4, True, (24, 18, 11, 40, "array", "bipolar"))])
def test_hip_slabs_count_in_facefluxes(oracle, tmp_path, world, async_mode, case):
    """otmb_facefluxes_slab_counts_dev: every slab's facefluxes kernel counts for its own transportmatrix -- halo levels as neighbours,
    wet ranks from the slab's base -- so no rank launches a counting pass or a push-mask kernel, and the matrices are the whole-grid
    oracle's bit for bit."""
    z = run_ranks(world, "hip", case, tmp_path, async_mode=async_mode, extra_env={"OTMB_TEST_KERNELS": "1"})
    check_against_whole_grid(oracle, z, case)
    for k in _kernels(tmp_path, world):
        assert k.get("facefluxes_kernel", 0) > 0 and (("tm_count_kernel" not in k and "push_mask_kernel" not in k) or not COUNTS_ON), k


def test_hip_slabs_count_in_facefluxes_centred_weighting(oracle, tmp_path):
    case = (24, 18, 11, 42, "array", "tripolar")
    z = run_ranks(3, "hip", case, tmp_path, async_mode=True, extra_env={"OTMB_TEST_KERNELS": "1", "OTMB_TEST_CENTRED": "1"})
    check_against_whole_grid(oracle, z, case, upwind=False)
    for k in _kernels(tmp_path, 3):
        assert "tm_count_kernel" not in k or not COUNTS_ON, k


def test_hip_slabs_without_counts_still_count_for_themselves(oracle, tmp_path):
    case = (24, 18, 11, 37, "array", "tripolar")
    z = run_ranks(2, "hip", case, tmp_path, async_mode=True, extra_env={"OTMB_TEST_KERNELS": "1", "OTMB_COUNT_IN_FF": "0"})
    check_against_whole_grid(oracle, z, case)
    for k in _kernels(tmp_path, 2):
        assert k.get("tm_count_kernel", 0) > 0, k


@pytest.mark.parametrize("pieces,rows", [(3, "4"), (3, "1")])
def test_hip_slabs_count_in_row_bands(oracle, tmp_path, pieces, rows):
    """The chain in row bands: under the four-row wave geometry (large planes; forced here) every band adds its counts to the same
    buffer; under the one-row geometry a band is not a whole number of wave segments and the slab counts for itself."""
    case = (70, 9, 12, 41, "array", "tripolar")
    z = run_ranks(3, "hip", case, tmp_path, async_mode=True, pieces=pieces, extra_env={"OTMB_TEST_KERNELS": "1", "OTMB_FF_ROWS": rows})
    check_against_whole_grid(oracle, z, case)
    for k in _kernels(tmp_path, 3):
        assert ("tm_count_kernel" not in k) == (rows == "4" and COUNTS_ON), k


def _gpus():
    import torch

    return torch.cuda.device_count()  # (does not initialise the GPU in this process)


@pytest.mark.parametrize("async_mode", [False, True])
def test_hip_slabs_over_rccl_one_gpu_per_rank(oracle, tmp_path, async_mode):
    """BASELINE.json configs[3] in small, as it runs on a multi-GPU node: every rank its own GPU, the facefluxes chain's planes
    and the setup halos over RCCL (backend "nccl") point-to-point -- skipped on a one-GPU box.  Strong-scaled small grid,
    compared with the whole-grid oracle bit for bit."""
    n = _gpus()
    if n < 2:
        pytest.skip(f"needs at least 2 GPUs for one rank per GPU over RCCL ({n} visible)")
    world = min(n, 4)
    case = (24, 18, 12, 36, "array", "tripolar")
    z = run_ranks(world, "hip", case, tmp_path, async_mode=async_mode, rccl=True, timeout=600)
    check_against_whole_grid(oracle, z, case)
