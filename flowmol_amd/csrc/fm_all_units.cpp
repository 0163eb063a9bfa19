// Unity build of the library: every translation unit in ONE (dev tooling only -- tools/build_variant.sh, tools/asm_stats.sh, the -DFM_TRACE / -DFM_PHASE_TIMING builds,
// whose device-side counters must exist once).  flowmol_amd/build.py compiles the units separately and in parallel and does NOT compile this file.
#include "fm_engine.cpp"
#include "fm_tu_msg32.cpp"
#include "fm_tu_msg16.cpp"
#include "fm_tu_node.cpp"
