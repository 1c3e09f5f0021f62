#!/bin/bash
# Ablation of the fill pass with the timing-only debug macros (wrong results, same launch geometry), each variant over
# several array placements (tools/ab_variants.py).  Run on the GPU box:  gpurun -- bash tools/dbg_variants.sh
#   OTMB_DBG_NOSTORE    no global stores in the write phase
#   OTMB_DBG_NOLDS      no LDS staging either (with NOSTORE)
#   OTMB_DBG_NOVALLOAD  the 44 value-only loads of a column are replaced by arithmetic (pattern inputs still loaded)
#   OTMB_DBG_NOEW       east/west values copied from the centre value (no second load into a line that is in flight)
#   OTMB_DBG_MULDIV     the 24-28 divisions of a column become multiplications
# Round-1 result at 1 degree (ms): base 0.391 | NOSTORE 0.241 | NOVALLOAD 0.297 | NOSTORE+NOVALLOAD 0.168 | MULDIV 0.408 |
# NOEW 0.391 -- the three parts (arithmetic + 13 pattern loads + LDS, value loads, stores) ADD UP; fewer VALU instructions
# or fewer loads into busy lines change nothing.
cd ${GRAFT_REPO_ROOT:-/root/repo}
# (round 5: the macros live in tools/experiments/timing_ablations.patch, not in the product sources -- apply it for this run, revert after)
(cd oceantransportmatrixbuilder.jl_amd/csrc && patch -p0 < ../../tools/experiments/timing_ablations.patch) || exit 1
trap '(cd oceantransportmatrixbuilder.jl_amd/csrc && patch -R -p0 < ../../tools/experiments/timing_ablations.patch)' EXIT
REPS=${REPS:-2} ROUNDS=${ROUNDS:-3} python tools/ab_variants.py base="" nostore="-DOTMB_DBG_NOSTORE" \
    novalload="-DOTMB_DBG_NOVALLOAD" neither="-DOTMB_DBG_NOSTORE -DOTMB_DBG_NOVALLOAD" muldiv="-DOTMB_DBG_MULDIV" \
    noew="-DOTMB_DBG_NOEW" 2>&1 | tail -6
