#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  The reference is Julia and cannot run in the build image, and its tests
hold no golden vectors for this path, so these fixtures are produced by the CPU
oracle (oracle/otmb_oracle.c, itself pinned by the independent pure-Python transliteration): they are
REGRESSION vectors for the oracle and the HIP path, not reference outputs -- parity with the Julia
reference stays unpinned.  Each file holds the full inputs of one small case (so it does not depend on
the RNG) and every output of the path: Lwet, the six ϕ arrays, and the five CSC matrices.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from helpers import MATS, make_case  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = ["tiny_tripolar", "tiny_bipolar", "odd_nx_fold", "nx2", "even_fold_open", "float32_flux", "tiny_rho3d"]
HD = ("west", "east", "south", "north")


def main():
    for name in CASES:
        g, gm = make_case(name)
        idx = orc.makeindices(gm.v3D)
        fill = g.umo.properties["_FillValue"]
        phi = orc.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], fill, gm.gridtopology.kind)
        out = dict(topology=gm.gridtopology.kind, fill=fill, umo=g.umo.data, vmo=g.vmo.data, v3D=gm.v3D,
                   thkcello=gm.thkcello, area2D=gm.area2D, zt=gm.zt, mlotst=g.mlotst, rho=np.asarray(g.rho),
                   kappa=np.array([g.kappaH, g.kappaVML, g.kappaVdeep]), Lwet=idx["Lwet"])
        for d in HD:
            out[f"edge_{d}"] = gm.edge_length_2D[d]
            out[f"dist_{d}"] = gm.distance_to_neighbour_2D[d]
        for k in orc.PHI_ORDER:
            out[f"phi_{k}"] = phi[k]
        for upwind in (True, False):
            tm = orc.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, upwind)
            for q, m in enumerate(MATS):
                for part, arr in zip(("colptr", "rowval", "nzval"), tm[m]):
                    out[f"{'up' if upwind else 'ce'}_{q}_{part}"] = arr
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {m: len(tm[m][1]) for m in MATS})
    make_lump()


LUMP_CASES = [("tiny_tripolar", 2, 2, 1, False), ("tiny_tripolar", 3, 2, 2, True), ("odd_nx_fold", 2, 3, 1, True), ("tiny_bipolar", 4, 4, 1, False)]


def make_lump():
    """lump_and_spray (src/extratools.jl:38-119) regression vectors, tests/golden/lump/*.npz: inputs (wet3D, vol, T's
    pattern, mask, block size) and outputs (LUMP rows/values, SPRAY structure, vol_c) of the oracle."""
    os.makedirs(os.path.join(HERE, "lump"), exist_ok=True)
    for q, (name, di, dj, dk, usemask) in enumerate(LUMP_CASES):
        g, gm = make_case(name)
        idx = orc.makeindices(gm.v3D)
        phi = orc.facefluxes(g.umo.data, g.vmo.data, idx["wet3D"], g.umo.properties["_FillValue"], gm.gridtopology.kind)
        tm = orc.transportmatrix(phi, gm, idx, g.rho, g.mlotst, g.kappaH, g.kappaVML, g.kappaVdeep, True)
        wet = idx["wet3D"].astype(bool)
        vol = gm.v3D.reshape(-1, order="F")[wet.reshape(-1, order="F")]
        mask = None
        if usemask:
            rng = np.random.default_rng(100 + q)
            mask = rng.random(wet.shape) < 0.6
            mask[:, : wet.shape[1] // 3, :] = False
        LUMP, SPRAY, vol_c = orc.lump_and_spray(wet, vol, tm["T"], mask, di, dj, dk)
        np.savez_compressed(os.path.join(HERE, "lump", f"{name}_{di}x{dj}x{dk}{'_mask' if usemask else ''}.npz"), wet3D=wet, vol=vol,
                            T_colptr=tm["T"][0], T_rowval=tm["T"][1], mask=np.ones(wet.shape, bool) if mask is None else mask,
                            block=np.array([di, dj, dk]), lump_rowval=LUMP[1], lump_nzval=LUMP[2], spray_colptr=SPRAY[0],
                            spray_rowval=SPRAY[1], vol_c=vol_c)
        print("lump", name, (di, dj, dk), len(vol), "->", len(vol_c))


if __name__ == "__main__":
    main()
