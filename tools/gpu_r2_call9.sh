# Round-2 call 9: anatomy of the split-precision kernels: ablation variants run with FM_PRECISION=bf16x3; EdgeUpdate with 4 workgroups per CU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/c9_ablate_sp.jsonl
for L in $R/build_ab/abl_*.so; do
  timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 FM_PRECISION=bf16x3 2>&1 | grep '^{' >> $O/c9_ablate_sp.jsonl
done
python - <<PY
import json
for l in open('$O/c9_ablate_sp.jsonl'):
    d = json.loads(l); print(d['lib'], d['eval_ms'], d['kernels_us'])
PY
