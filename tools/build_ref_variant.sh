#!/bin/bash
# Build the library of an older commit as a variant for tools/ab_variants.py:
#   tools/build_ref_variant.sh <commit> <name>   ->  oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip_<name>.so
# then:  python tools/ab_variants.py old=@oceantransportmatrixbuilder.jl_amd/lib/libotmb_hip_<name>.so new=""
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
git -C "$REPO" archive "$1" oceantransportmatrixbuilder.jl_amd include | tar -x -C "$TMP"
python3 "$REPO/tools/_build_variant.py" "$TMP" "$REPO" "$2"
rm -rf "$TMP"
