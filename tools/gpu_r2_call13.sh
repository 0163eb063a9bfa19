# 16-row edge tiles on 4-wave workgroups (4 independent workgroups per CU) vs the 32-row / 8-wave kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/flowmol_amd/libflowmol_hip.so
: > $O/c13_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' >> $O/c13_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 16 32 1024 47 32 2>&1 | grep '^{' >> $O/c13_ab.jsonl
timeout 200 python $R/tools/ab_bench.py $L 16 32 1024 47 32 FM_EDGE_THREADS=256 2>&1 | grep '^{' >> $O/c13_ab.jsonl
python - <<PY
import json
for l in open('$O/c13_ab.jsonl'):
    d = json.loads(l); print(d['tile_edge'], d['env'], d['eval_ms'], d['mol_per_s_at_250'], d['kernels_us'], d['parity_out_rel'])
PY
