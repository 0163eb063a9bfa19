# gate-GEMM weight fragments requested before the barrier that precedes the gate phase (build_ab/gp_1.so) vs baseline (gp_0.so), alternating
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/c24_ab.jsonl
for rep in 1 2 3; do for L in $R/build_ab/gp_0.so $R/build_ab/gp_1.so; do
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' >> $O/c24_ab.jsonl
done; done
python - <<PY
import json
for l in open('$O/c24_ab.jsonl'):
    d = json.loads(l); print(d['lib'], d['eval_ms'], d['kernels_us']['edge_message'], d['kernels_us']['node_update'], d['parity_out_rel'])
PY
