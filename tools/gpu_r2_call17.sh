# prologue order of fm_k_edge_message: hoisted-scalar gathers requested before (baseline) or after the hidden-vector gathers
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/c17_ab.jsonl
for rep in 1 2; do for L in $R/build_ab/pre_0.so $R/build_ab/pre_1.so; do
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' >> $O/c17_ab.jsonl
done; done
python - <<PY
import json
for l in open('$O/c17_ab.jsonl'):
    d = json.loads(l); print(d['lib'], d['eval_ms'], d['kernels_us']['edge_message'], d['parity_out_rel'])
PY
