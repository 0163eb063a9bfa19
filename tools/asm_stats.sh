#!/bin/bash
# static instruction mix of the GVP kernels (device asm of the current sources)
D=${TMPDIR:-/tmp}/fm_asm; mkdir -p $D
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-pass-failed ${FM_ASM_FLAGS--ffp-contract=off} -x hip "$(dirname "$0")/../flowmol_amd/csrc/fm_all_units.cpp" -S --cuda-device-only -o $D/e.s || exit 1
python3 - "$D/e.s" <<'PY'
import re, sys
cur = None; stats = {}
for l in open(sys.argv[1]):
    m = re.match(r'^(_Z\d+fm_k_\S+):', l)
    if m: cur = m.group(1); stats[cur] = dict(mfma=0, fma=0, valu=0, salu=0, buf=0, glob=0, lds=0, saveexec=0); continue
    if l.startswith('.Lfunc_end'): cur = None; continue
    if cur is None: continue
    t = l.strip()
    if not t or t[0] in '.;': continue
    d = stats[cur]
    if t.startswith('v_mfma'): d['mfma'] += 1
    elif t.startswith('v_'):
        d['valu'] += 1
        if t.startswith(('v_fma_f32', 'v_fmac_f32', 'v_pk_fma_f32', 'v_fmaak_f32', 'v_fmamk_f32')): d['fma'] += 1
    elif t.startswith('s_'): d['salu'] += 1
    if t.startswith('buffer_'): d['buf'] += 1
    if t.startswith('global_'): d['glob'] += 1
    if t.startswith('ds_'): d['lds'] += 1
    if 's_and_saveexec' in t: d['saveexec'] += 1
txt = open(sys.argv[1]).read()
for k, d in stats.items():
    m = re.search(r'\.name:\s+' + re.escape(k) + r'\n(?:.*\n){0,14}?\s+\.vgpr_count:\s+(\d+)', txt)
    sp = re.search(re.escape(k) + r'\.private_seg_size, (\d+)', txt)
    if d['mfma']: print(k[3:60].ljust(58), d, 'vgpr', m.group(1) if m else '?')
PY
