# MLP pairing only for small batches: per-kernel times at 1024 molecules (paired vs separate), random-batch parity tests, bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/flowmol_amd/libflowmol_hip.so
rm -f $O/parity_report.jsonl
timeout 900 python -m pytest $R/tests -m gpu -q -k "random_batches or fixture_directly or emulated or sample_api or launch" 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/c16_pytest.log
: > $O/c16_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 FM_PAIR_MLPS=1 2>&1 | grep '^{' >> $O/c16_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' >> $O/c16_ab.jsonl
timeout 600 python $R/bench.py --no-cpu-baseline --no-api-e2e > $O/c16_bench.json 2>/dev/null
cat $O/c16_pytest.log; cut -c1-220 $O/c16_bench.json
python - <<PY
import json
for l in open('$O/c16_ab.jsonl'):
    d = json.loads(l); print(d['env'].get('FM_PAIR_MLPS','auto'), d['eval_ms'], d['mol_per_s_at_250'], d['kernels_us'])
PY
