"""A/B micro-benchmark of kernel variants on the GPU box (one process per variant).

    python tools/ab_bench.py --lib <lib.so> [--preset flowmol3] [--mols 1024] [--atoms 47] [--precision f32] [--tuning tile_edge=32,tile_node=32]
prints one JSON line: per-kernel avg us over 3 profiled network evaluations, eval wall ms, the max relative
output error of a small parity batch against the CPU oracle, and a sha256 of the full batch's outputs (equal for bit-identical variants)."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
ap = argparse.ArgumentParser()
ap.add_argument('--lib', default=str(ROOT / 'flowmol_amd' / 'libflowmol_hip.so'))
ap.add_argument('--preset', default='flowmol3')
ap.add_argument('--mols', type=int, default=1024)
ap.add_argument('--atoms', type=int, default=47)
ap.add_argument('--precision', default='f32')
ap.add_argument('--tuning', default='', help='fm_config launch-tuning overrides, k=v[,k=v...]')
args = ap.parse_args()
tuning = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.tuning.split(',') if kv}

import torch                                            # noqa: E402
from flowmol_amd import _lib, presets, weights          # noqa: E402
from flowmol_amd.engine import Engine                   # noqa: E402
from oracle import cpu_ref                              # noqa: E402
from parity_util import forward_compare                 # noqa: E402

cfg = presets.PRESETS[args.preset]()
sd = weights.synth_state_dict(cfg, 0)
eng = Engine(cfg, sd, device='cuda:0', lib=_lib.load(args.lib), precision=args.precision, tuning=tuning)
orc = cpu_ref.OracleVF(cfg, sd)
errs, _, _ = forward_compare(eng, orc, cfg, torch.tensor([9, 70, 33, 2]), 0.4, True, taps=False)
B, n = args.mols, args.atoms
n_atoms = torch.full((B,), n, dtype=torch.int64)
eng.bind(n_atoms)
g = torch.Generator(device='cuda:0').manual_seed(0)
x0 = torch.randn(eng.N, 3, device='cuda:0', generator=g)
eng.remove_com(x0)
st = eng.prior_state(x0)
st['a_t'] = torch.randint(0, cfg.n_atom_types + 1, (eng.N,), device='cuda:0', dtype=torch.int32, generator=g)
st['e_t'] = torch.randint(0, cfg.n_bond_types + 1, (eng.U,), device='cuda:0', dtype=torch.int32, generator=g)
sc = cfg.self_conditioning
prev = eng.forward(st, 0.0, bootstrap=True) if sc else None
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
for k in ('edge_message', 'edge_message_pq', 'edge_update', 'edge_update_head', 'node_update', 'pos_update', 'node_proj', 'node_proj_asd', 'sc', 'heads', 'sc_edge', 'sc_node', 'edge_head', 'node_head',
          'embed_table', 'gather_rows', 'event_overhead'):
    ms, cnt = eng.profile_get(k)
    if cnt:
        kern[k] = round(ms * 1e3 / cnt, 1)
import hashlib                                          # noqa: E402
# bit-level fingerprints of the full batch's outputs: equal between two libraries <=> same arithmetic
sha = {k: hashlib.sha256(out[k].detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12] for k in sorted(out) if torch.is_tensor(out[k])}
print(json.dumps({'lib': Path(args.lib).name, 'out_sha': sha, 'preset': args.preset, 'precision': args.precision, 'tuning': tuning, 'mols': B, 'atoms': n, 'eval_ms': round(wall, 2),
                  'mol_per_s_at_250': round(B / (250 * wall / 1e3), 2), 'kernels_us': kern,
                  'parity_out_rel': {k: float(f'{v:.2e}') for k, v in errs.items()}}))
