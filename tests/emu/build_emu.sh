#!/bin/bash
# TEST TOOLING: build the host emulation of libflowmol_hip (same sources, host clang++, shim HIP header); the translation units in parallel, like flowmol_amd/build.py.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
[ -x "$CXX" ] || CXX=clang++
TMP=$(mktemp -d /tmp/fm_emu_XXXXXX); trap 'rm -rf "$TMP"' EXIT
FLAGS="-O2 -std=c++17 -fPIC -ffp-contract=off -mfma -Wno-unused-value -I $HERE"
pids=()
for u in "$ROOT"/flowmol_amd/csrc/*.cpp; do
    [ "$(basename "$u")" == "fm_all_units.cpp" ] && continue
    $CXX $FLAGS -c -x c++ "$u" -o "$TMP/$(basename "$u" .cpp).o" & pids+=($!)
done
$CXX $FLAGS -c "$HERE/emu_rt.cpp" -o "$TMP/emu_rt.o" & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$CXX -shared -fPIC "$TMP"/*.o -o "$HERE/libflowmol_emu.so" -lpthread
echo "built $HERE/libflowmol_emu.so"
