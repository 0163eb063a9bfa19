#!/bin/bash
# TEST TOOLING: build the host emulation of libflowmol_hip (same sources, host clang++, shim HIP header).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
[ -x "$CXX" ] || CXX=clang++
$CXX -O2 -g -std=c++17 -fPIC -shared -ffp-contract=off -mfma -Wno-unused-value \
    -I "$HERE" -x c++ "$ROOT/flowmol_amd/csrc/fm_engine.cpp" "$HERE/emu_rt.cpp" \
    -o "$HERE/libflowmol_emu.so" -lpthread
echo "built $HERE/libflowmol_emu.so"
