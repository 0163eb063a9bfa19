"""Turn the outputs of `tools/gpu_run.sh <gtag> ... bench <name> ... + profile <name> ...` (merged into gpurun_out/) into the committed
profiles/<ptag>_* files: kernel stats, the three PMC summaries, the bench line, the HBM-traffic JSON, and -- for the headline workload --
profiles/current_pmc.json (the committed counters bench.py quotes, labelled with the commit of the library they were measured on).

    python tools/collect_final.py <gtag> <name> <ptag> [--current]
e.g. python tools/collect_final.py r3f main r03f --current ;  python tools/collect_final.py r3b c5 r03b_c5
"""
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
G, P = ROOT / 'gpurun_out', ROOT / 'profiles'
gtag, name, ptag = sys.argv[1:4]
D = G / f'{gtag}_prof_{name}'


def take(src, out):
    txt = (D / src).read_text()
    txt = re.sub(r'/\S*?gpurun_out/', 'gpurun_out/', txt)
    (P / out).write_text(txt)
    return txt


take('stats.txt', f'{ptag}_kernel_stats.txt')
sq = take('pmc_sq.txt', f'{ptag}_pmc_sq.txt')
fetch = take('pmc_fetch.txt', f'{ptag}_pmc_fetch.txt')
write = take('pmc_write.txt', f'{ptag}_pmc_write_tcc.txt')
for extra in sorted(D.glob('pmc_extra_*.txt')):
    take(extra.name, f'{ptag}_{extra.stem[:48]}.txt')


def pass_line(log):
    """The bench line the profiled command itself printed (every pass logs it): workload sizes and the digest of the library the counters were measured on."""
    for line in reversed((D / log).read_text().splitlines()):
        if line.startswith('{"metric"'):
            return json.loads(line)
    raise SystemExit(f'{D / log}: no bench line')


passes = {log: pass_line(log + '.log') for log in ('stats', 'pmc_sq', 'pmc_fetch', 'pmc_write')}
digests = {b['config'].get('library_digest') for b in passes.values()}
if len(digests) != 1:
    raise SystemExit(f'the passes ran on different libraries: {digests}')
bench_file = G / f'{gtag}_bench_{name}.json'
bench = json.loads(bench_file.read_text().strip().splitlines()[-1]) if bench_file.exists() else None
if bench is not None and bench['config'].get('library_digest') in digests:
    (P / f'{ptag}_bench.json').write_text(json.dumps(bench) + '\n')          # the un-profiled bench line of the same call
else:
    bench = passes['stats']                                                     # sizes / algorithmic bytes: the profiled run's own line (its timings carry the profiler)


def instances(txt, prefix):
    """{kernel instance name: (dispatches, {counter: per-dispatch average})} of every summary block whose kernel name starts with `prefix`."""
    out = {}
    for blk in re.split(r'\n\s*\n', txt):
        m = re.match(r'\s*(' + re.escape(prefix) + r'<[^>]*>)\S*\s+dispatches=(\d+)\s+avg_duration_us=([0-9.]+)', blk)
        if m:
            ctr = {c: float(v) for c, v in re.findall(r'^\s+(\w+)\s+([0-9.]+)\s*$', blk, re.M)}
            ctr['_avg_duration_us'] = float(m.group(3))
            out[m.group(1)] = (int(m.group(2)), ctr)
    return out


k = 'fm_k_edge_message'
inst_sq = instances(sq, k)
k_names = sorted(inst_sq)
# the instance bench.py's roofline is quoted on: the one that takes most of the step (the FULL edge-message kernel; the pair-slab (PQ)
# instance of the two leading convolutions is listed beside it)
k_main = max(inst_sq, key=lambda name: inst_sq[name][0] * inst_sq[name][1]['_avg_duration_us'])


def counter(txt, name, cname):
    return instances(txt, k)[name][1][cname]


f, w = counter(fetch, k_main, 'FETCH_SIZE'), counter(write, k_main, 'WRITE_SIZE')
E = bench['config']['directed_edges_per_gpu']
N = bench['config']['nodes_per_gpu']
traffic = {
    'kernel': k_main, 'workload': bench['config']['workload'],
    'per_instance': {name: {'dispatches': d, 'FETCH_SIZE_KiB': instances(fetch, k).get(name, (0, {}))[1].get('FETCH_SIZE'), 'WRITE_SIZE_KiB': instances(write, k).get(name, (0, {}))[1].get('WRITE_SIZE'),
                            'mfma_busy_frac': c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024)} for name, (d, c) in instances(sq, k).items()},
    'source': f'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py ... --no-cpu-baseline --no-api-e2e`, '
              f'per-dispatch averages: profiles/{ptag}_pmc_fetch.txt, profiles/{ptag}_pmc_write_tcc.txt',
    'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w,
    'correction': 'MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE as reported',
    'hbm_bytes_per_launch': int((2 * f + w) * 1024),
    'algorithmic_bytes_per_launch': bench['roofline']['algorithmic_bytes_per_launch'],
}
(P / f'{ptag}_traffic.json').write_text(json.dumps(traffic, indent=1) + '\n')
busy, gui = counter(sq, k_main, 'SQ_VALU_MFMA_BUSY_CYCLES'), counter(sq, k_main, 'GRBM_GUI_ACTIVE')
print(json.dumps({'value': bench['value'], 'ms_per_step': bench['ms_per_step'], 'roofline_frac': bench['roofline']['frac'], 'executed_frac': bench['roofline']['executed_frac'],
                  'mfma_busy_frac': busy / (gui / 8 * 1024), 'hbm_bytes_per_launch': traffic['hbm_bytes_per_launch'],
                  'algorithmic_bytes_per_launch': traffic['algorithmic_bytes_per_launch']}, indent=1))
if '--current' in sys.argv:
    # profiles/current_pmc.json: {workload: counters of its dominant kernel}; bench.py quotes the entry of the workload it runs
    commit = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    wl = {'main': 'c3'}.get(name, name)
    path = P / 'current_pmc.json'
    try:
        allw = json.loads(path.read_text())
        if 'kernel' in allw:          # the single-workload format of rounds 1-2
            allw = {}
    except Exception:
        allw = {}
    allw[wl] = {'nodes_per_gpu': N, 'directed_edges_per_gpu': E, 'collected_at_commit': commit, 'library_digest': digests.copy().pop(),
                'kernel': k_main, 'other_instances': [n for n in k_names if n != k_main], 'hbm_bytes_per_launch': traffic['hbm_bytes_per_launch'], 'source': f'profiles/{ptag}_traffic.json',
                'mfma_busy_frac': busy / (gui / 8 * 1024), 'source_sq': f'profiles/{ptag}_pmc_sq.txt',
                'note': 'GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs = 256 CUs x 4'}
    path.write_text(json.dumps(allw, indent=1) + '\n')
