"""Static check of julia/OceanTransportMatrixBuilderAMD.jl (no Julia toolchain exists in the build image, so the shim cannot
be executed): its `struct TmArgs` must have the fields of `otmb_tm_args` (include/otmb.h) in the same order with
layout-compatible types, and every `ccall` must name a symbol the header declares, with the return type and the argument
types of the C prototype -- checked against oceantransportmatrixbuilder.jl_amd/capi.py, whose bindings ARE executed by the
GPU test-suite, and against the header text itself."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = open(os.path.join(ROOT, "julia", "OceanTransportMatrixBuilderAMD.jl"), encoding="utf-8").read()
HEADER = open(os.path.join(ROOT, "include", "otmb.h"), encoding="utf-8").read()


def split_top(s):
    """Split a comma-separated list, ignoring commas inside {} or ()."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "{(":
            depth += 1
        elif ch in "})":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_kind(t):
    """Julia type -> (size in bytes, kind) with kind in {"ptr", "i32", "i64", "f64", "void"}; NTuple{n,T} -> n copies."""
    t = t.strip()
    m = re.fullmatch(r"NTuple\{(\d+),\s*(.+)\}", t)
    if m:
        return [julia_kind(m.group(2))[0]] * int(m.group(1))
    if t.startswith("Ptr{") or t in ("Cstring",):
        return ["ptr"]
    return [{"Int32": "i32", "Int64": "i64", "Float64": "f64", "Cvoid": "void", "UInt8": "u8"}[t]]


def ctypes_kind(t):
    if t is None:
        return ["void"]
    if isinstance(t, type) and issubclass(t, C.Array):
        return ctypes_kind(t._type_) * t._length_
    if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
        return ["ptr"]
    return [{C.c_int32: "i32", C.c_int64: "i64", C.c_double: "f64"}[t]]


def c_kind(t):
    """C parameter / return type text -> kind."""
    t = t.strip()
    if "*" in t or "[" in t:
        return "ptr"
    t = t.replace("const", "").strip()
    return {"int32_t": "i32", "int64_t": "i64", "double": "f64", "void": "void", "uint64_t": "i64"}[t.split()[0]]


def header_prototypes():
    protos = {}
    text = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(otmb_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else split_top(args)
        protos[name] = (c_kind(ret), [c_kind(re.sub(r"\b\w+(\[\d*\])?$", lambda mm: mm.group(1) or "", p.strip()) if "*" not in p and "[" not in p else p)
                                     for p in params])
    return protos


def test_tmargs_struct_matches_header_and_ctypes_mirror():
    from otmb_amd import capi

    body = re.search(r"struct TmArgs\n(.*?)\nend", SHIM, re.S).group(1)
    fields = []
    for line in body.splitlines():
        line = line.split("#")[0]
        for decl in line.split(";"):
            decl = decl.strip()
            if decl:
                name, typ = decl.split("::")
                fields.append((name.strip(), typ.strip()))
    mirror = capi.TmArgs._fields_
    assert [n for n, _ in fields] == [n for n, _ in mirror]
    for (name, jt), (_, ct) in zip(fields, mirror):
        assert julia_kind(jt) == ctypes_kind(ct), name
    # and the header's own struct, field by field
    hbody = re.search(r"typedef struct \{(.*?)\} otmb_tm_args;", HEADER, re.S).group(1)
    hbody = re.sub(r"/\*.*?\*/", "", hbody, flags=re.S)
    hfields = []
    for decl in hbody.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.fullmatch(r"((?:const )?\w+) (.+)", decl)
        base, names = m.group(1), m.group(2)
        for n in names.split(","):
            mm = re.fullmatch(r"\s*(\*?)\s*(\w+)(?:\[(\d+)\])?\s*", n)
            kind = "ptr" if mm.group(1) else c_kind(base)
            hfields.append((mm.group(2), [kind] * int(mm.group(3) or 1)))
    assert [n for n, _ in hfields] == [n for n, _ in fields]
    for (name, hk), (_, jt) in zip(hfields, fields):
        assert hk == julia_kind(jt), name


def test_every_ccall_matches_its_c_prototype():
    from otmb_amd import capi

    protos = header_prototypes()
    calls = re.findall(r"ccall\(\s*(?:sym|Libdl\.dlsym)\((?:lib\[\],\s*)?:(\w+)\)\s*,\s*(\w+)\s*,\s*\((.*?)\)\s*,", SHIM, re.S)
    assert len(calls) >= 9
    seen = set()
    for name, ret, args in calls:
        seen.add(name)
        assert name in protos, f"{name} is not declared in include/otmb.h"
        jargs = [k for a in split_top(args.replace("\n", " ")) for k in julia_kind(a)]
        cret, cargs = protos[name]
        assert julia_kind(ret)[0] == cret, name
        assert jargs == cargs, (name, jargs, cargs)
        # the ctypes mirror that the GPU tests execute agrees as well
        res, argtypes = capi.SYMBOLS[name]
        assert ctypes_kind(res) == [cret], name
        assert [k for t in argtypes for k in ctypes_kind(t)[:1]] == cargs, name
    for must in ("otmb_ctx_create", "otmb_makeindices", "otmb_facefluxes", "otmb_transportmatrix_plan", "otmb_transportmatrix_fetch",
                 "otmb_lump_and_spray", "otmb_last_error", "otmb_ctx_set_reuse_grid"):
        assert must in seen, must


def test_status_codes_the_shim_maps_exist_in_the_header():
    codes = dict(re.findall(r"(OTMB_ERR_\w+)\s*=\s*(\d+)", HEADER))
    assert codes["OTMB_ERR_ALL_MISSING"] == "8" and "rc == 8 && throw(AssertionError" in SHIM
    assert codes["OTMB_ERR_INVALID_ARG"] == "11" and "rc == 11 && throw(ArgumentError" in SHIM
    assert codes["OTMB_ERR_ASYMMETRIC_PATTERN"] == "16" and "rc == 16 && throw(ArgumentError" in SHIM
