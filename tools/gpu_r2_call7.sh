# Round-2 call 7: opt-in split precision (bf16x3 edge messages): GPU tests, then f32 vs bf16x3 per-kernel timing at the bench size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl
timeout 1200 python -m pytest $R/tests -m gpu -q -x -k "split_precision" 2>&1 | tail -15 > $O/c7_pytest_sp.log
L=$R/flowmol_amd/libflowmol_hip.so
: > $O/c7_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' >> $O/c7_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 FM_PRECISION=bf16x3 2>&1 | grep '^{' >> $O/c7_ab.jsonl
cat $O/c7_pytest_sp.log; grep split $O/parity_report.jsonl | cut -c1-600
python - <<PY
import json
for l in open('$O/c7_ab.jsonl'):
    d = json.loads(l); print(d['env'], d['eval_ms'], d['mol_per_s_at_250'], d['kernels_us'], d['parity_out_rel'])
PY
