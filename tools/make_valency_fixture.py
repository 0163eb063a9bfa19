"""Build-container-only: collect the reference's shipped valid-valency tables
(/root/reference/data/<set>/train_data_valencies_{kekulized,aromatic}.json, loaded by the reference's
SampleAnalyzer.__init__, flowmol/analysis/metrics.py:66-80) into flowmol_amd/data/valencies.json,
keyed by dataset name, plus the fixed MiDi table (metrics.py:27-41)."""
import json
from pathlib import Path

REF = Path('/root/reference/data')
OUT = Path(__file__).resolve().parent.parent / 'flowmol_amd' / 'data' / 'valencies.json'
out = {}
for d in sorted(REF.iterdir()):
    for f in sorted(d.glob('train_data_valencies_*.json')):
        out[d.name] = {'explicit_aromaticity': 'aromatic' in f.name, 'table': json.loads(f.read_text())}
OUT.write_text(json.dumps(out, separators=(',', ':')))
print({k: (v['explicit_aromaticity'], sorted(v['table'])) for k, v in out.items()}, OUT.stat().st_size, 'bytes')
