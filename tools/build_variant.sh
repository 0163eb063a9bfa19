#!/bin/bash
# Dev tooling: build a variant of libflowmol_hip.so with extra compiler flags into build_ab/<name>/ (git-ignored; travels to the GPU box).
#   tools/build_variant.sh abl32 -DFM_ABLATE=32      then      python tools/ab_bench.py --lib build_ab/abl32/libflowmol_hip.so ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=$1; shift
mkdir -p $R/build_ab/$N
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -Wno-pass-failed "$@" -x hip $R/flowmol_amd/csrc/fm_all_units.cpp -o $R/build_ab/$N/libflowmol_hip.so
echo built build_ab/$N "$@"
