# Round-2 call 8: split precision incl. EdgeUpdate: GPU tests, per-kernel timing, bench line of the opt-in mode
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl
timeout 1200 python -m pytest $R/tests -m gpu -q -x -k "split_precision" 2>&1 | tail -8 > $O/c8_pytest_sp.log
L=$R/flowmol_amd/libflowmol_hip.so
: > $O/c8_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' >> $O/c8_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 FM_PRECISION=bf16x3 2>&1 | grep '^{' >> $O/c8_ab.jsonl
timeout 600 python $R/bench.py --precision bf16x3 --no-cpu-baseline > $O/c8_bench_sp.json 2> $O/c8_bench_sp.err
cat $O/c8_pytest_sp.log; grep split $O/parity_report.jsonl | cut -c1-300
python - <<PY
import json
for l in open('$O/c8_ab.jsonl'):
    d = json.loads(l); print(d['env'].get('FM_PRECISION','f32'), d['eval_ms'], d['mol_per_s_at_250'], d['kernels_us'], d['parity_out_rel'])
d = json.loads(open('$O/c8_bench_sp.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['dtype'][:30], d['roofline'], {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}, d.get('api_end_to_end'))
PY
