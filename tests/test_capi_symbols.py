"""CPU: the C-ABI library loads and exports every symbol include/otmb.h declares (no compute calls)."""
import os
import re

import otmb_amd
from otmb_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "otmb.h"), encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(otmb_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/otmb.h but not exported"
    assert set(names) == set(capi.SYMBOLS), set(names) ^ set(capi.SYMBOLS)
    assert lib.otmb_version().startswith(b"otmb_hip")
    assert lib.otmb_status_string(1).decode() == "ρ contains NaNs"
    assert lib.otmb_status_string(2).decode() == "Tadv contains NaNs."
    assert lib.otmb_status_string(3).decode() == "TκH contains NaNs."
    assert lib.otmb_status_string(7).decode() == "Unknown grid type"


def test_no_cpu_fallback_in_product_path():
    """The product package must never import, load or link the oracle (comments may mention it)."""
    pkg = os.path.join(ROOT, "oceantransportmatrixbuilder.jl_amd")
    banned = re.compile(r"(import\s+oracle|from\s+oracle|libotmb_oracle|oracle[/\\.]o|otmb_oracle\.c|orc_[a-z_]+\()")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".jl")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                m = banned.search(src)
                assert m is None, f"{f} references the oracle: {m.group(0)}"
    for f in ("julia/OceanTransportMatrixBuilderAMD.jl", "include/otmb.h"):
        assert banned.search(open(os.path.join(ROOT, f), encoding="utf-8").read()) is None
