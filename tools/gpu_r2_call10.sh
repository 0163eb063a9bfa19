R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/flowmol_amd/libflowmol_hip.so
: > $O/c10_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 FM_PRECISION=bf16x3 2>&1 | grep '^{' >> $O/c10_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 64 32 1024 47 32 FM_PRECISION=bf16x3 2>&1 | grep '^{' >> $O/c10_ab.jsonl
python - <<PY
import json
for l in open('$O/c10_ab.jsonl'):
    d = json.loads(l); print(d['tile_edge'], d['eval_ms'], d['mol_per_s_at_250'], d['kernels_us'], d['parity_out_rel'])
PY
