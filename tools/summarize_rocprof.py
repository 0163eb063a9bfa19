"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite output) into small text files for profiles/.

    python tools/summarize_rocprof.py stats <results.db> > profiles/rNN_kernel_stats.txt
    python tools/summarize_rocprof.py pmc <results.db> [kernel-substring ...] > profiles/rNN_pmc_*.txt
"""
import sqlite3
import sys
from collections import defaultdict


def short(name, n=70):
    name = name.replace('void ', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def stats(db):
    con = sqlite3.connect(db)
    rows = list(con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'))
    print(f'# rocprofv3 --kernel-trace --stats   ({db})')
    print(f'{"kernel":72s} {"calls":>6s} {"total_us":>12s} {"avg_us":>11s} {"pct":>7s}')
    for name, calls, tot, avg, pct in rows:
        print(f'{short(name):72s} {calls:6d} {tot:12.1f} {avg:11.2f} {pct:7.2f}')


def pmc(db, filters):
    con = sqlite3.connect(db)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    dur = defaultdict(float)
    q = 'select kernel_name,counter_name,value,dispatch_id,duration from counters_collection'
    for name, cname, val, did, d in con.execute(q):
        if filters and not any(f in name for f in filters):
            continue
        acc[name][cname] += val
        if did not in cnt[name]:
            cnt[name].add(did)
            dur[name] += d
    print(f'# rocprofv3 --pmc (per-dispatch AVERAGES; {db})')
    for name in sorted(acc, key=lambda k: -dur[k]):
        n = len(cnt[name])
        print(f'\n{short(name, 100)}   dispatches={n}  avg_duration_us={dur[name] / n / 1e3:.1f}')
        for cname in sorted(acc[name]):
            print(f'    {cname:32s} {acc[name][cname] / n:18.1f}')


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3:])
