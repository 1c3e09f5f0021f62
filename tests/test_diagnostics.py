"""The two diagnostics behind bench.py's `roofline.box` (include/otmb.h): otmb_ctx_box_ceilings (plain HBM read / write streams of this box)
and otmb_ctx_stream_mix (an ideal streaming kernel over the fill pass's own arrays, destructive for the outputs).  They never touch a
result; what is checked here is that they run, report sane rates, stay inside the arrays they are given, and leave the assembler usable."""
import numpy as np
import pytest

from helpers import MATS, assert_csc_equal, make_case

pytestmark = pytest.mark.gpu


def test_box_ceilings_and_stream_mix(oracle):
    import torch

    from otmb_amd.device import DeviceAssembler

    g, gm = make_case("small_rho3d")
    ref = oracle.makeindices(gm.v3D)
    fill = g.umo.properties["_FillValue"]
    rphi = oracle.facefluxes(g.umo.data, g.vmo.data, ref["wet3D"], fill, gm.gridtopology.kind)
    rtm = oracle.transportmatrix(rphi, gm, ref, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
    asm = DeviceAssembler(0)
    asm.set_grid(gm, g.mlotst, g.rho, g.kappaH, g.kappaVML, g.kappaVdeep)
    umo = torch.from_numpy(np.asfortranarray(g.umo.data).ravel(order="F")).cuda()
    vmo = torch.from_numpy(np.asfortranarray(g.vmo.data).ravel(order="F")).cuda()
    rd, wr = asm.ctx.box_ceilings()
    assert 1000.0 < rd < 9000.0 and 1000.0 < wr < 9000.0, (rd, wr)  # GB/s: an MI355X streams a few TB/s, never more than its 8 TB/s
    asm.step(umo, vmo, fill)
    # guard words behind every output array: the stream must stay inside the lengths it is told
    guards = {}
    for k, m in enumerate(MATS):
        for q, t in enumerate(asm.out[m]):
            n = (asm.N + 1) if q == 0 else asm.nnz[k]
            if t.numel() > n + 2:
                t[n:n + 2] = 7 if t.dtype == torch.int64 else 7.0
                guards[(m, q)] = (t, n)
    inputs_before = [p.clone() for p in asm.phi] + [asm.v3d.clone(), asm.lwet3d.clone()]
    gbs = asm.fill_pass_stream_mix()
    assert set(gbs) == {256, 512, 1024, 2048} and all(v > 0.0 for v in gbs.values())
    for (m, q), (t, n) in guards.items():
        assert (t[n:n + 2] == 7).all(), (m, q)
    for a, b in zip(inputs_before, [*asm.phi, asm.v3d, asm.lwet3d]):
        assert torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all()  # inputs are only read
    asm.step(umo, vmo, fill)  # the matrices were overwritten by the diagnostic: the next step rebuilds them
    got = asm.result_to_host()
    for m in MATS:
        assert_csc_equal(got[m], rtm[m], m)
