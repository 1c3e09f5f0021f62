"""GPU: the depth-slab path with the HIP backend.  Two ranks share the box's single GPU and talk over
gloo (RCCL needs one GPU per rank; the transport layer is the only difference -- see otmb_amd.dist.Comm),
and the concatenated result must equal the whole-grid oracle bit for bit."""
import pytest

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
