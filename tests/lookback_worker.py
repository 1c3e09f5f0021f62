"""Child process of test_gpu_parity.test_single_kernel_lookback_variant: the library reads OTMB_LOOKBACK once per process."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
assert os.environ.get("OTMB_LOOKBACK") == "1"
import torch  # noqa: E402

from helpers import MATS, assert_csc_equal, gridmetrics_of, make_case  # noqa: E402
from oracle import oracle  # noqa: E402
from otmb_amd import synthetic  # noqa: E402
from otmb_amd.device import DeviceAssembler  # noqa: E402

oracle.build()
cases = [make_case(n) for n in ("tiny_tripolar", "odd_nx_fold", "small_rho3d")]
g = synthetic.make_grid(90, 80, 20, seed=77, rho="array")  # several hundred tiles: the look-back spans many windows
cases.append((g, gridmetrics_of(g)))
for g, gm in cases:
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True, tight=True)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    asm.ctx.timing_enable(True)
    for _ in range(3):
        asm.step(umo, vmo, fill, onepass=True)
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)
    names = set(asm.ctx.timing_collect())
    assert "tm_kernel<onepass>" in names and "tm_kernel<fill>" not in names, names
print("LOOKBACK_OK")
