"""Build-container-only: convert the reference's shipped size histograms
(/root/reference/data/<set>/train_data_n_atoms_histogram.pt, the (n_atoms, counts) int64 pair that
reference FlowMol.build_n_atoms_dist loads, flowmol/models/flowmol.py:461-466) into the small JSON
data file flowmol_amd/data/n_atoms_hist.json used by sample_random_sizes."""
import json
from pathlib import Path

import torch

REF = Path('/root/reference/data')
OUT = Path(__file__).resolve().parent.parent / 'flowmol_amd' / 'data' / 'n_atoms_hist.json'
out = {}
for d in sorted(REF.iterdir()):
    f = d / 'train_data_n_atoms_histogram.pt'
    if f.exists():
        n, c = torch.load(f)
        out[d.name] = {'n_atoms': n.tolist(), 'counts': c.tolist()}
OUT.parent.mkdir(parents=True, exist_ok=True)
OUT.write_text(json.dumps(out, separators=(',', ':')))
print({k: (len(v['n_atoms']), sum(v['counts'])) for k, v in out.items()}, OUT.stat().st_size, 'bytes')
