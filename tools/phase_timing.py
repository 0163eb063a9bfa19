"""Dev aid: per-phase shader cycles inside fm_k_edge_message (thread 0 of every workgroup), from a
-DFM_PHASE_TIMING build of the library:

    hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -DFM_PHASE_TIMING -x hip flowmol_amd/csrc/fm_all_units.cpp -o /path/lib_timing.so
    python tools/phase_timing.py /path/lib_timing.so
"""
import ctypes
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch                                             # noqa: E402
from flowmol_amd import _lib, presets, weights           # noqa: E402
from flowmol_amd.engine import Engine                    # noqa: E402

libp = sys.argv[1]
lib = _lib.load(libp)
raw = ctypes.CDLL(libp)
cfg = presets.flowmol3()
sd = weights.synth_state_dict(cfg, 0)
eng = Engine(cfg, sd, device='cuda:0', lib=lib)
B, n = 1024, 47
eng.bind(torch.full((B,), n, dtype=torch.int64))
g = torch.Generator(device='cuda:0').manual_seed(0)
x0 = torch.randn(eng.N, 3, device='cuda:0', generator=g)
eng.remove_com(x0)
st = eng.prior_state(x0)
prev = eng.forward(st, 0.0, bootstrap=True)
out = eng.new_dst()
eng.forward(st, 0.3, prev=prev, out=out)
eng.synchronize()
buf = (ctypes.c_ulonglong * 64)()
raw.fm_tlog_read(buf, 1)
R = 3
for _ in range(R):
    eng.forward(st, 0.3, prev=prev, out=out)
eng.synchronize()
raw.fm_tlog_read(buf, 0)
tile = 32        # the automatic choice at this batch size
ntiles = (eng.E + tile - 1) // tile * cfg.n_convs * R
names = {0: 'meta+geom', 1: 'fill', 40: 'dbg', 41: 'aggregate'}
for base, g_ in ((10, 'g0'), (20, 'g1'), (30, 'g2')):
    for k, nm in enumerate(['gemm1', 'cross+sh', 'vuGEMM+bias', 'sGEMM', 'barrier', 'silu', 'gates', 'gating']):
        names[base + k] = f'{g_}.{nm}'
res = {names[i]: round(buf[i] / ntiles) for i in sorted(names) if buf[i]}
print(json.dumps({'cycles_per_tile_thread0': res, 'total': sum(res.values())}))
