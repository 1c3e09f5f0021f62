#!/bin/bash
# timing experiments: rebuild the library with debug macros on the GPU box and time the kernels
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  python - <<PY
import importlib.util, sys
spec = importlib.util.spec_from_file_location("b", "oceantransportmatrixbuilder.jl_amd/build.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
b.build(force=True, extra="$v".split())
PY
  echo "variant [$v]"; python tools/ab_protocols.py 2>&1 | grep twophase | tail -1
done
