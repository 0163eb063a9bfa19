"""Output fingerprints of one library under every accepted launch tuning (dev tooling, GPU box).

    python tools/fingerprint_matrix.py --lib <lib.so> [--tag name]

One JSON line: for each of the tuning settings of tests/test_gpu_parity.py::test_forward_matches_oracle_under_every_accepted_tuning, the sha256 of
the outputs (x, a, c, e) of one self-conditioned network evaluation of a fixed 40-molecule batch (sizes 2 .. 130) with fixed state, plus the max
relative output error against the CPU oracle on a small batch.  Two libraries built from the same arithmetic -- e.g. two arrangements of a fill loop
under -ffp-contract=off -- must give identical lines; VERDICT r4 #2."""
import argparse
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
ap = argparse.ArgumentParser()
ap.add_argument('--lib', default=str(ROOT / 'flowmol_amd' / 'libflowmol_hip.so'))
ap.add_argument('--tag', default=None)
args = ap.parse_args()

import torch                                            # noqa: E402
from flowmol_amd import _lib, presets, weights          # noqa: E402
from flowmol_amd.engine import Engine                   # noqa: E402

TUNINGS = [{}, {'tile_edge': 64, 'tile_node': 64}, {'tile_edge_update': 64}, {'tile_edge': 64, 'tile_node': 64, 'tile_edge_update': 64, 'pair_slab': -1},
           {'pair_slab': -1}, {'pair_slab': 1}, {'pair_slab': 1, 'pair_mlps': -1, 'mlp_small_tiles': -1}, {'xcd_swizzle': -1, 'fuse_node': -1},
           {'tile_edge': 64, 'pair_slab': 1}, {'pair_slab': 1, 'pair_mlps': 1, 'mlp_small_tiles': -1},
           {'tile_node': 4}, {'tile_node': 4, 'tile_edge': 32, 'pair_slab': 1}, {'tile_node': 8}, {'tile_node': 12}, {'tile_node': 20}, {'tile_node': 16},
           {'mlp_small_tiles': 2}, {'mlp_small_tiles': 2, 'pair_mlps': -1}, {'mlp_small_tiles': 1}, {'fuse_node': 2}]
cfg = presets.flowmol3()
sd = weights.synth_state_dict(cfg, 0)
lib_path = Path(args.lib) if Path(args.lib).is_absolute() else ROOT / args.lib          # gpu_run.sh runs its steps from /tmp
lib = _lib.load(lib_path)
gsz = torch.Generator().manual_seed(3)
sizes = torch.cat([torch.tensor([70, 2, 47, 130]), torch.randint(5, 90, (36,), generator=gsz)])
out = {'lib': args.tag or lib_path.parent.name + '/' + lib_path.name, 'molecules': int(sizes.numel()), 'fingerprints': {}}
for tn in TUNINGS:
    eng = Engine(cfg, sd, device='cuda:0', lib=lib, precision='f32', tuning=tn)
    eng.bind(sizes)
    g = torch.Generator(device='cuda:0').manual_seed(0)
    x0 = torch.randn(eng.N, 3, device='cuda:0', generator=g)
    eng.remove_com(x0)
    st = eng.prior_state(x0)
    st['a_t'] = torch.randint(0, cfg.n_atom_types + 1, (eng.N,), device='cuda:0', dtype=torch.int32, generator=g)
    st['c_t'] = torch.randint(0, cfg.n_charges + 1, (eng.N,), device='cuda:0', dtype=torch.int32, generator=g)
    st['e_t'] = torch.randint(0, cfg.n_bond_types + 1, (eng.U,), device='cuda:0', dtype=torch.int32, generator=g)
    prev = eng.forward(st, 0.0, bootstrap=True)
    o = eng.forward(st, 0.3, prev=prev)
    eng.synchronize()
    h = hashlib.sha256()
    for k in 'xace':
        h.update(o[k].detach().cpu().contiguous().numpy().tobytes())
    out['fingerprints'][json.dumps(tn, sort_keys=True)] = h.hexdigest()[:16]
    eng.close()
print(json.dumps(out))
