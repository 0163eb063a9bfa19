"""The CPU oracle on the long-horizon reference trajectories (tests/golden/long_*.npz) over their FULL length (the -m "not gpu"
suite runs only the first steps to stay within minutes):   python tools/oracle_long_parity.py > profiles/rNN_oracle_long_parity.jsonl"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
from flowmol_amd import presets, weights          # noqa: E402
from oracle import cpu_ref                         # noqa: E402
from oracle.make_golden import LONG_CASES          # noqa: E402
from parity_util import oracle_long_golden         # noqa: E402

torch.set_num_threads(8)
for tag in (sys.argv[1:] or LONG_CASES):
    name = LONG_CASES[tag][0]
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(ROOT / 'tests' / 'golden' / f'long_{tag}.npz').items()}
    cfg = presets.PRESETS[name]()
    sd = weights.long_fixture_weights(cfg, g)
    t0 = time.time()
    res = oracle_long_golden(cpu_ref.OracleVF(cfg, sd), cfg, g)
    print(json.dumps({'fixture': f'long_{tag}.npz', 'impl': 'oracle/cpu_ref.py (CPU, f32)', **res, 'seconds': round(time.time() - t0, 1)}), flush=True)
