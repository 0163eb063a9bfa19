"""Dev aid: how do the phases of workgroups that share a CU line up?  Needs a -DFM_TRACE build:
    hipcc ... -DFM_TRACE -x hip flowmol_amd/csrc/fm_all_units.cpp -o lib_trace.so ; python tools/trace_phases.py lib_trace.so out.npz"""
import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch                                             # noqa: E402
from flowmol_amd import _lib, presets, weights           # noqa: E402
from flowmol_amd.engine import Engine                    # noqa: E402

libp, outp = sys.argv[1], sys.argv[2]
lib = _lib.load(libp)
raw = ctypes.CDLL(libp)
cfg = presets.flowmol3()
sd = weights.synth_state_dict(cfg, 0)
eng = Engine(cfg, sd, device='cuda:0', lib=lib)
eng.bind(torch.full((1024,), 47, dtype=torch.int64))
g = torch.Generator(device='cuda:0').manual_seed(0)
x0 = torch.randn(eng.N, 3, device='cuda:0', generator=g)
eng.remove_com(x0)
st = eng.prior_state(x0)
prev = eng.forward(st, 0.0, bootstrap=True)
out = eng.new_dst()
eng.forward(st, 0.3, prev=prev, out=out)
eng.synchronize()
buf = (ctypes.c_ulonglong * (16384 * 16))()
raw.fm_trace_read(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(16384, 16).copy()
np.savez_compressed(outp, trace=a)
hw = a[:, 9]
cu = (hw >> np.uint64(8)) & np.uint64(0xF)
sh = (hw >> np.uint64(12)) & np.uint64(1)
se = (hw >> np.uint64(13)) & np.uint64(7)
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
key = (xcc * 8 + se) * 32 + sh * 16 + cu
print('distinct CU keys', len(np.unique(key)), 'of blocks', len(key))
# for each CU: sort its blocks by start; report overlap fraction of sGEMM intervals between co-resident blocks
t0 = a[:, 10].astype(np.int64)
tend = a[:, 8].astype(np.int64)
sg = [(a[:, 2].astype(np.int64), a[:, 3].astype(np.int64)), (a[:, 4].astype(np.int64), a[:, 5].astype(np.int64)), (a[:, 6].astype(np.int64), a[:, 7].astype(np.int64))]
tot_sg = 0
tot_ov = 0
durs = []
for k in np.unique(key):
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(t0[idx])]
    ivs = []
    for b in idx:
        for s_, e_ in sg:
            ivs.append((s_[b], e_[b], b))
        durs.append(tend[b] - t0[b])
    ivs.sort()
    for i, (s1, e1, b1) in enumerate(ivs):
        tot_sg += e1 - s1
        for s2, e2, b2 in ivs[i + 1:]:
            if s2 >= e1:
                break
            if b2 != b1:
                tot_ov += min(e1, e2) - s2
print('mean tile duration (cycles)', float(np.mean(durs)))
print('sum of sGEMM interval lengths', tot_sg, ' pairwise overlap between different workgroups on the same CU', tot_ov,
      ' overlap fraction', tot_ov / max(tot_sg, 1))
k0 = np.unique(key)[5]
idx = np.where(key == k0)[0]
idx = idx[np.argsort(t0[idx])][:8]
base = t0[idx].min()
for b in idx:
    print('block', b, 'start', t0[b] - base, 'sgemm', [(int(s_[b] - base), int(e_[b] - base)) for s_, e_ in sg], 'end', tend[b] - base)
