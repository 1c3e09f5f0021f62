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
