"""Build libotmb_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

-ffp-contract=off is REQUIRED for bit parity: hipcc's default (fast) would fuse ρ̄*v, κ*a and
d*V with neighbouring adds into FMAs, which round differently from the reference's separate
IEEE operations.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libotmb_hip.so")
# (LLVM's alternative machine-scheduler strategies -- max-memory-clause, max-ilp, iterative-* -- were compared with
# tools/ab_variants.py over several array placements each: none beats the default on the fill pass.)
# -amdgpu-atomic-optimizer-strategy=None: every atomic of this library with a wave-uniform address is issued by ONE elected lane already (the
# counts in facefluxes: two per wave and level); LLVM's optimizer wraps each in its own lane election + popcount multiply (~12 instructions).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-bitwise-instead-of-logical", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]

def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "otmb.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=(), name=None):
    """name: build a VARIANT library lib/libotmb_hip_<name>.so (perf A/B in one process, tools/ab_variants.py)."""
    lib = LIB if name is None else os.path.join(LIBDIR, f"libotmb_hip_{name}.so")
    if name is None and not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ("" if name is None else "_" + name) + ".o")
        objs.append(obj)
        drop = {e[len("-REMOVE:"):] for e in extra if e.startswith("-REMOVE:")}  # A/B variants: strip a default flag
        flags = [f for f in FLAGS if f not in drop] + [e for e in extra if not e.startswith("-REMOVE:")]
        cmd = [hipcc, *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
