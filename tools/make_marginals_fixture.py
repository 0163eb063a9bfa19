"""Build-container-only: convert the reference's shipped marginal distributions
(/root/reference/data/<set>/train_data_marginal_dists.pt = (p_a, p_c, p_e, p_c_given_a), the tensors the 'marginal' and
'c-given-a' priors of flowmol/data_processing/priors.py:67-98 sample from) into the small JSON data file
flowmol_amd/data/marginal_dists.json.  Values are float32; their Python-float repr round-trips exactly."""
import json
from pathlib import Path

import torch

REF = Path('/root/reference/data')
OUT = Path(__file__).resolve().parent.parent / 'flowmol_amd' / 'data' / 'marginal_dists.json'
out = {}
for d in sorted(REF.iterdir()):
    f = d / 'train_data_marginal_dists.pt'
    if f.exists():
        p_a, p_c, p_e, p_ca = torch.load(f)
        out[d.name] = {'p_a': p_a.tolist(), 'p_c': p_c.tolist(), 'p_e': p_e.tolist(), 'p_c_given_a': p_ca.tolist()}
OUT.write_text(json.dumps(out, separators=(',', ':')))
print({k: [len(v['p_a']), len(v['p_c']), len(v['p_e'])] for k, v in out.items()}, OUT.stat().st_size, 'bytes')
