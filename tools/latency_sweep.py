"""Per-step latency of the integration loop vs batch size (whole fm_integrate path, wall clock around a synchronised
window of steps), flowmol3 architecture, 47-atom molecules.  python tools/latency_sweep.py [sizes...]"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch                                             # noqa: E402
from flowmol_amd import presets, weights                 # noqa: E402
from flowmol_amd.engine import Engine, IntegrationRun, StepNoise, make_step_plan   # noqa: E402

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 8, 32, 128, 512]
tuning = {a.split('=')[0]: int(a.split('=')[1]) for a in sys.argv[1:] if '=' in a and not a.startswith('lib=')}      # fm_config launch-tuning overrides, e.g. fuse_node=-1
libarg = [a[4:] for a in sys.argv[1:] if a.startswith('lib=')]       # lib=<path relative to the repository root>: an A/B build (tools/build_variant.sh)
philox = 'philox' in sys.argv[1:]          # in-kernel noise: no torch RNG launches between the steps
cfg = presets.flowmol3()
from flowmol_amd import _lib                             # noqa: E402
eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cuda:0', tuning=tuning, lib=_lib.load(str(ROOT / libarg[0])) if libarg else None)
dev = eng.device
geom = 'geom' in sys.argv[1:]              # molecule sizes drawn from the shipped GEOM-drugs histogram (seed 1000 + B) instead of 47 atoms each
for B in sizes:
    if geom:
        from flowmol_amd.model import load_n_atoms_hist
        vals, counts = load_n_atoms_hist('geom_full_kekulized')
        n_atoms = vals[torch.multinomial(counts.double(), B, replacement=True, generator=torch.Generator().manual_seed(1000 + B))].to(torch.int64)
    else:
        n_atoms = torch.full((B,), 47, dtype=torch.int64)
    eng.bind(n_atoms)
    N, U = eng.N, eng.U
    plan = make_step_plan(250, cfg.stochasticity, cfg.high_confidence_threshold, cfg.cat_temperature, philox_seed=11 if philox else None)
    x0 = torch.randn(N, 3, device=dev)
    eng.remove_com(x0)
    state = eng.prior_state(x0)
    run = IntegrationRun(eng, state, plan, None if philox else (lambda i, last: StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, last, dev)))
    run.run(0, 8, chunk=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.run(8, 72, chunk=32)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 64
    eng.profile(True)
    run.run(72, 74, chunk=2)
    torch.cuda.synchronize()
    gpu_ms = 0.0
    nl = 0
    per_kernel = {}
    for k in ('edge_message', 'edge_message_pq', 'edge_update', 'edge_update_head', 'node_update', 'pos_update', 'node_proj', 'node_proj_asd', 'sc_edge', 'sc_node', 'edge_head',
              'node_head', 'sc', 'heads', 'ctmc', 'embed_table', 'remove_com', 'x_step'):
        ms, cnt = eng.profile_get(k)
        gpu_ms += ms
        nl += cnt
        if cnt:
            per_kernel[k] = round(ms / cnt * 1e3, 1)       # us per launch
    eng.profile(False)
    print(json.dumps({'lib': libarg[0] if libarg else 'flowmol_amd/libflowmol_hip.so', 'sizes': 'geom' if geom else 47, 'edges': int((n_atoms * (n_atoms - 1)).sum()),
                      'mols': B, 'noise': 'philox' if philox else 'torch', 'tuning': tuning, 'ms_per_step_wall': round(dt * 1e3, 3),
                      'sum_kernel_ms_per_step': round(gpu_ms / 2, 3),
                      'launches_per_step': nl / 2, 'mol_per_s_at_250': round(B / (250 * dt), 2), 'us_per_launch': per_kernel}))
