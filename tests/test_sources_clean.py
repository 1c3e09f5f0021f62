"""The shipped kernel sources hold no timing ablation: a stray -D must never build a library that stores other values than the
reference's.  Round 4's review found seven such macros (OTMB_DBG_MULDIV, _NOVALLOAD, _NOEW, _NOGENERIC, _NOSTORE, _NOLDS,
OTMB_STAGGER_UNITS) inside the product kernels; they live in tools/experiments/timing_ablations.patch now."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd", "csrc")
# preprocessor switches the product sources may test.  Every one of them leaves every stored value what the reference computes:
# diagnostics (time stamps) and tunables.  (Round 6: the never-default alternatives -- the dense-tile march formulation, the look-back
# one-pass mode, OTMB_ALIGNED16 / _CHECKS_IN_FILL / _PLAIN_STORES / _PLAIN_PHI_LOADS / _MARCH_SOUTH_FIRST -- left the product for
# tools/experiments/r06_removed_formulations.patch.)
ALLOWED = {
    "OTMB_DBG_STAMPS", "OTMB_DBG_STAMPS_ORDER",  # s_memtime stamps of the fill pass's phases (tools/stamps.py): extra output, same matrices
    "OTMB_MARCH_AUTO_ROWS", "OTMB_MARCH_AUTO_COLS", "TM_THREADS", "TM_WAVES_PER_SIMD", "TM_COUNT_TPB", "FF_KB", "FF_COUNTS_WAVES",  # tunables
}


def test_no_wrong_value_switch_in_the_product_sources():
    seen = set()
    for f in sorted(x for x in os.listdir(CSRC) if x.endswith((".hip", ".h"))):
        text = open(os.path.join(CSRC, f), encoding="utf-8").read()
        for m in re.finditer(r"^\s*#\s*(?:ifdef|ifndef|if|elif)\b(.*)$", text, re.M):
            for name in re.findall(r"\b[A-Z][A-Z0-9_]{3,}\b", m.group(1)):
                if name.startswith(("OTMB_", "TM_", "FF_", "DM_")) and not name.endswith("_H"):
                    seen.add((name, f))
    bad = sorted((n, f) for n, f in seen if n not in ALLOWED)
    assert not bad, f"switches that are not on the list of value-preserving ones: {bad}"
    assert not any("DBG" in n and "STAMPS" not in n for n, _ in seen)


def test_the_ablation_patch_still_applies():
    """tools/experiments/timing_ablations.patch must stay usable (tools/dbg_variants.sh applies it): dry run against the shipped sources."""
    patch = os.path.join(ROOT, "tools", "experiments", "timing_ablations.patch")
    r = subprocess.run(["patch", "--dry-run", "-p0", "-i", patch], cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
