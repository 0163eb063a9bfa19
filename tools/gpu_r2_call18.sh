# split precision: 32-row / 8-wave tiles (2 workgroups per CU) vs 64-row tiles on 8-wave and on 16-wave workgroups (1 per CU, half the weight stream)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/flowmol_amd/libflowmol_hip.so
: > $O/c18_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 FM_PRECISION=bf16x3 2>&1 | grep '^{' >> $O/c18_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 64 32 1024 47 32 FM_PRECISION=bf16x3 2>&1 | grep '^{' >> $O/c18_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 64 32 1024 47 32 FM_PRECISION=bf16x3 FM_EDGE_THREADS=1024 2>&1 | grep '^{' >> $O/c18_ab.jsonl
python - <<PY
import json
for l in open('$O/c18_ab.jsonl'):
    d = json.loads(l); print(d['tile_edge'], d['env'].get('FM_EDGE_THREADS', 512), d['eval_ms'], d['mol_per_s_at_250'], d['kernels_us'], d['parity_out_rel'])
PY
