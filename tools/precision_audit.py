"""Per-stage error of one library / precision against the CPU oracle evaluated in FLOAT64 (dev tooling, GPU box).

    python tools/precision_audit.py --lib <lib.so> --precision f32|bf16x3|bf16x6|f16x3 [--tag name]

One JSON line: {stage: max |kernel - float64| / max |float64|} for every tap and output of one self-conditioned evaluation of a
5 / 12 / 47 / 2 / 33-atom batch (the batch of tests/test_gpu_parity.py::test_three_term_split_is_f32_class_against_float64)."""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))
ap = argparse.ArgumentParser()
ap.add_argument('--lib', default='flowmol_amd/libflowmol_hip.so')
ap.add_argument('--precision', default='f32')
ap.add_argument('--tag', default=None)
args = ap.parse_args()

import torch                                            # noqa: E402
from flowmol_amd import _lib, presets, weights          # noqa: E402
from flowmol_amd.engine import Engine                   # noqa: E402
from parity_util import forward_compare, oracle_f64     # noqa: E402

lib_path = Path(args.lib) if Path(args.lib).is_absolute() else ROOT / args.lib
cfg = presets.flowmol3()
sd = weights.synth_state_dict(cfg, 0)
eng = Engine(cfg, sd, device='cuda:0', lib=_lib.load(lib_path), precision=args.precision)
errs, out, _ = forward_compare(eng, oracle_f64(cfg, sd), cfg, torch.tensor([5, 12, 47, 2, 33]), 0.5, True, dtype=torch.float64)
print(json.dumps({'lib': args.tag or lib_path.parent.name, 'precision': args.precision, 'stage_errors_vs_float64': errs}))
