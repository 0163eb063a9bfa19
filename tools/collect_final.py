"""Turn the outputs of tools/gpu_final_round<N>.sh (merged into gpurun_out/) into the committed profiles/<tag>_* files: kernel stats, the
three PMC summaries, the bench lines, the HBM-traffic JSON, and profiles/current_pmc.json (the committed counters bench.py quotes,
labelled with the commit of the library they were measured on).

    python tools/collect_final.py r02f
"""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / 'gpurun_out', ROOT / 'profiles'
tag = sys.argv[1]
S = [sys.executable, str(ROOT / 'tools' / 'summarize_rocprof.py')]


def db(d):
    c = sorted((G / d).rglob('*_results.db'))
    assert c, f'no rocprofv3 db under {G / d}'
    return str(c[-1])


def run(args, out):
    txt = subprocess.run(S + args, check=True, capture_output=True, text=True).stdout.replace(str(ROOT) + '/', '')
    (P / out).write_text(txt)
    return txt


run(['stats', db('fprof')], f'{tag}_kernel_stats.txt')
run(['pmc', db('fpmc1'), 'fm_k_'], f'{tag}_pmc_sq.txt')
fetch = run(['pmc', db('fpmc2'), 'fm_k_'], f'{tag}_pmc_fetch.txt')
write = run(['pmc', db('fpmc3'), 'fm_k_'], f'{tag}_pmc_write_tcc.txt')
bench = json.loads((G / 'final_bench.json').read_text().strip().splitlines()[-1])
(P / f'{tag}_bench.json').write_text(json.dumps(bench) + '\n')


def counter(txt, kernel, name):
    blk = txt[txt.index(kernel):]
    return float(re.search(rf'{name}\s+([0-9.]+)', blk).group(1))


k = 'fm_k_edge_message<32, 32, 512, 0, 0>'
f, w = counter(fetch, k, 'FETCH_SIZE'), counter(write, k, 'WRITE_SIZE')
E = bench['config']['directed_edges_per_gpu']
N = bench['config']['nodes_per_gpu']
traffic = {
    'kernel': 'fm_k_edge_message<32,32,512,0,0>',
    'workload': bench['config']['workload'],
    'source': f'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`, '
              f'per-dispatch averages: profiles/{tag}_pmc_fetch.txt, profiles/{tag}_pmc_write_tcc.txt',
    'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w,
    'correction': 'MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE as reported',
    'hbm_bytes_per_launch': int((2 * f + w) * 1024),
    'algorithmic_bytes_per_launch': int(E * (512 + 8) + N * 2 * (1024 + 384)),   # ef row + src/dst ids per edge, ~2 partial-sum rows per node
    'mols_per_gpu': bench['config']['global_molecules'] // bench['n_gpus'], 'n_atoms': N // (bench['config']['global_molecules'] // bench['n_gpus']),
}
(P / f'{tag}_traffic.json').write_text(json.dumps(traffic, indent=1) + '\n')
sq = (P / f'{tag}_pmc_sq.txt').read_text()
busy, gui = counter(sq, k, 'SQ_VALU_MFMA_BUSY_CYCLES'), counter(sq, k, 'GRBM_GUI_ACTIVE')
commit = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
cur = {'preset': 'flowmol3', 'mols_per_gpu': traffic['mols_per_gpu'], 'n_atoms': traffic['n_atoms'], 'commit': commit,
       'kernel': traffic['kernel'], 'hbm_bytes_per_launch': traffic['hbm_bytes_per_launch'], 'source': f'profiles/{tag}_traffic.json',
       'mfma_busy_frac': busy / (gui / 8 * 1024), 'source_sq': f'profiles/{tag}_pmc_sq.txt',
       'note': 'GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs = 256 CUs x 4'}
(P / 'current_pmc.json').write_text(json.dumps(cur, indent=1) + '\n')
for extra in ('final_bench_sizedist.json',):
    if (G / extra).exists():
        (P / f'{tag}_bench_sizedist_geom.json').write_text((G / extra).read_text().strip().splitlines()[-1] + '\n')
print(json.dumps({'value': bench['value'], 'ms_per_step': bench['ms_per_step'], 'roofline': bench.get('roofline'), 'traffic': traffic['hbm_bytes_per_launch']}, indent=1)[:1500])
print((G / 'final_pytest.log').read_text())
