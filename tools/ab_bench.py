"""A/B micro-benchmark of kernel variants on the GPU box (one process per variant).

    python tools/ab_bench.py <lib.so> <tile_edge> <tile_node> [mols] [n_atoms]
prints one JSON line: per-kernel avg us over 3 profiled network evaluations, eval wall ms, and the
max relative output error of a small parity batch against the CPU oracle."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
lib_path, te, tn = sys.argv[1], sys.argv[2], sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
n = int(sys.argv[5]) if len(sys.argv) > 5 else 47
os.environ['FM_TILE_EDGE'] = te
os.environ['FM_TILE_NODE'] = tn
if len(sys.argv) > 6:
    os.environ['FM_TILE_EUPD'] = sys.argv[6]
for kv in sys.argv[7:]:
    k_, v_ = kv.split('=')
    os.environ[k_] = v_

import torch                                            # noqa: E402
from flowmol_amd import _lib, presets, weights          # noqa: E402
from flowmol_amd.engine import Engine                   # noqa: E402
from oracle import cpu_ref                              # noqa: E402
from parity_util import forward_compare                 # noqa: E402

cfg = presets.flowmol3()
sd = weights.synth_state_dict(cfg, 0)
eng = Engine(cfg, sd, device='cuda:0', lib=_lib.load(lib_path))
orc = cpu_ref.OracleVF(cfg, sd)
errs, _, _ = forward_compare(eng, orc, cfg, torch.tensor([9, 70, 33, 2]), 0.4, True, taps=False)
n_atoms = torch.full((B,), n, dtype=torch.int64)
eng.bind(n_atoms)
g = torch.Generator(device='cuda:0').manual_seed(0)
x0 = torch.randn(eng.N, 3, device='cuda:0', generator=g)
eng.remove_com(x0)
st = eng.prior_state(x0)
st['a_t'] = torch.randint(0, cfg.n_atom_types + 1, (eng.N,), device='cuda:0', dtype=torch.int32, generator=g)
st['e_t'] = torch.randint(0, cfg.n_bond_types + 1, (eng.U,), device='cuda:0', dtype=torch.int32, generator=g)
prev = eng.forward(st, 0.0, bootstrap=True)
out = eng.new_dst()
for _ in range(2):
    eng.forward(st, 0.3, prev=prev, out=out)
eng.synchronize()
t0 = time.perf_counter()
R = 4
for _ in range(R):
    eng.forward(st, 0.3, prev=prev, out=out)
eng.synchronize()
wall = (time.perf_counter() - t0) / R * 1e3
eng.profile(True)
for _ in range(3):
    eng.forward(st, 0.3, prev=prev, out=out)
eng.synchronize()
kern = {}
for k in ('edge_message', 'edge_update', 'node_update', 'pos_update', 'node_proj', 'node_proj_asd', 'sc_edge', 'sc_node', 'edge_head', 'node_head'):
    ms, cnt = eng.profile_get(k)
    if cnt:
        kern[k] = round(ms * 1e3 / cnt, 1)
print(json.dumps({'lib': Path(lib_path).name, 'tile_edge': int(te), 'tile_node': int(tn), 'tile_eupd': os.environ.get('FM_TILE_EUPD', '32'), 'env': {k: v for k, v in os.environ.items() if k.startswith('FM_')}, 'mols': B, 'eval_ms': round(wall, 2),
                  'mol_per_s_at_250': round(B / (250 * wall / 1e3), 2), 'kernels_us': kern,
                  'parity_out_rel': {k: float(f'{v:.2e}') for k, v in errs.items()}}))
