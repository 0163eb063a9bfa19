"""Dev aid: per-phase shader cycles of the GVP kernels at a SMALL batch (thread 0 of every workgroup), from a -DFM_PHASE_TIMING build:

    tools/build_variant.sh timing -DFM_PHASE_TIMING
    python tools/phase_timing_small.py build_ab/timing/libflowmol_hip.so [mols=1] [atoms=47]

Prints, per kernel family, the cycles between consecutive marks of fm_gvp_core summed over one network evaluation and divided by the number of
tiles: edge message (marks 10.. / 20.. / 30.. = its three GVPs, 0 / 1 = prologue, 41 = aggregation) and the node kernels (marks 50..57: all
GVPs of fm_k_node_update and its fused tail share them; everything between two GVPs is charged to the next GVP's first mark)."""
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
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 47
lib = _lib.load(libp)
raw = ctypes.CDLL(libp)
cfg = presets.flowmol3()
eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cuda:0', lib=lib)
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
R = 5
for _ in range(R):
    eng.forward(st, 0.3, prev=prev, out=out)
eng.synchronize()
raw.fm_tlog_read(buf, 0)
te = 16 if (eng.E + 31) // 32 <= 256 else 32
tn = 16 if (eng.N + 31) // 32 <= 256 else 32
tiles_e, tiles_n = (eng.E + te - 1) // te, (eng.N + tn - 1) // tn
ph = ['gemm_Vh', 'cross+norms', 'Wu_gemm+acc_init', 'scalar_gemm', 'barrier', 'silu', 'gate_gemm', 'gating']
res = {'mols': B, 'atoms': n, 'edge_tile': te, 'node_tile': tn, 'edge_tiles': tiles_e, 'node_tiles': tiles_n}
edge = {'prologue.meta': buf[0], 'prologue.fill': buf[1], 'aggregate': buf[41]}
for base, gname in ((10, 'gvp0'), (20, 'gvp1'), (30, 'gvp2')):
    for k, nm in enumerate(ph):
        edge[f'{gname}.{nm}'] = buf[base + k]
res['edge_message_cycles_per_tile_and_evaluation'] = {k: round(v / R / tiles_e) for k, v in edge.items()}
res['edge_message_total'] = round(sum(edge.values()) / R / tiles_e)
node = {nm: buf[50 + k] for k, nm in enumerate(ph)}
res['node_kernels_cycles_per_tile_and_evaluation_all_33_gvps'] = {k: round(v / R / tiles_n) for k, v in node.items()}
res['node_kernels_total'] = round(sum(node.values()) / R / tiles_n)
print(json.dumps(res))
