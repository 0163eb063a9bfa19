# Split-precision vector-channel GEMMs (register split): SP parity tests, A/B per kernel, full GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/flowmol_amd/libflowmol_hip.so
rm -f $O/parity_report.jsonl
timeout 900 python -m pytest $R/tests -m gpu -q -k split_precision 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/c12_pytest_sp.log
cp $O/parity_report.jsonl $O/c12_sp_parity.jsonl 2>/dev/null
: > $O/c12_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 FM_PRECISION=bf16x3 2>&1 | grep '^{' >> $O/c12_ab.jsonl
timeout 300 python $R/bench.py --precision bf16x3 --no-cpu-baseline --no-api-e2e > $O/c12_bench_sp.json 2>/dev/null
timeout 1500 python -m pytest $R/tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 > $O/c12_pytest_all.log
cat $O/c12_pytest_sp.log $O/c12_pytest_all.log; cut -c1-300 $O/c12_bench_sp.json
python - <<PY
import json
for l in open('$O/c12_ab.jsonl'):
    d = json.loads(l); print(d['env'].get('FM_PRECISION','f32'), d['eval_ms'], d['mol_per_s_at_250'], d['kernels_us'], d['parity_out_rel'])
PY
