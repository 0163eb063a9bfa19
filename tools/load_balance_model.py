"""Load-balance model of the multi-GPU sampling path ("report measured 1-GPU throughput plus the load-balance
model" when no multi-GPU box is available): BASELINE configs[3] = 8192 molecules with sizes drawn (seeded) from the shipped GEOM-drugs
size histogram, dealt to 1/2/4/8 ranks by shard.partition_lpt (cost = directed edges n(n-1), which is what the step time follows:
profiles/r02q_bench_sizedist_geom.json).  Prints, per world size, the max/mean shard cost and the predicted whole-job throughput
N x R1 x mean/max, R1 = the measured 1-GPU rate on the same size distribution.

    python tools/load_balance_model.py [--r1 58.5] [--mols 8192] [--hist geom_full_kekulized]
"""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from flowmol_amd.model import load_n_atoms_hist        # noqa: E402
from flowmol_amd.shard import partition_lpt            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--r1', type=float, default=58.5, help='measured 1-GPU molecules/s on this size distribution (profiles/*_bench_sizedist_geom.json)')
ap.add_argument('--mols', type=int, default=8192)
ap.add_argument('--hist', default='geom_full_kekulized')
ap.add_argument('--seed', type=int, default=1000)
ap.add_argument('--gather-ms', type=float, default=1.0, help='the one all-gather of packed results (1.7 KB/molecule)')
args = ap.parse_args()
vals, counts = load_n_atoms_hist(args.hist)
sizes = vals[torch.multinomial(counts.double(), args.mols, replacement=True, generator=torch.Generator().manual_seed(args.seed))]
cost = (sizes * (sizes - 1)).double()
rows = []
t1 = args.mols / args.r1                      # seconds for the whole job on one GPU
for w in (1, 2, 4, 8):
    parts = partition_lpt(sizes, w)
    loads = torch.tensor([float(cost[p].sum()) for p in parts])
    naive = torch.tensor([float(cost[r::w].sum()) for r in range(w)])           # round-robin dealing, for contrast
    contiguous = torch.tensor([float(c.sum()) for c in torch.chunk(cost, w)])
    t = t1 * float(loads.max()) / float(cost.sum()) + (args.gather_ms / 1e3 if w > 1 else 0)
    rows.append({'ranks': w, 'molecules_per_rank': [int(len(p)) for p in parts], 'max_over_mean_cost_lpt': float(loads.max() / loads.mean()),
                 'max_over_mean_cost_round_robin': float(naive.max() / naive.mean()), 'max_over_mean_cost_contiguous': float(contiguous.max() / contiguous.mean()),
                 'predicted_molecules_per_s': args.mols / t, 'predicted_speedup': t1 / t, 'efficiency': t1 / t / w})
print(json.dumps({'molecules': args.mols, 'hist': args.hist, 'seed': args.seed, 'mean_atoms': float(sizes.double().mean()), 'max_atoms': int(sizes.max()),
                  'r1_molecules_per_s': args.r1, 'gather_ms': args.gather_ms, 'model': 'time(rank) = job_time_1gpu x cost(rank)/cost(all) [+ gather]; cost = sum n(n-1)',
                  'rows': rows}, indent=1))
