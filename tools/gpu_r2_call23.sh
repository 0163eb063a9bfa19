# edge-tile size vs batch size in the regime between "one workgroup per CU" and "chip full": auto (32 rows once 32-row tiles exceed the CU count) vs forced 16 / 32
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/c23_tiles.jsonl
for T in 16 32; do
  FM_TILE_EDGE=$T timeout 300 python $R/tools/latency_sweep.py 4 6 8 12 16 24 32 48 64 philox 2>&1 | grep "^{" | sed "s/^{/{\"tile_edge\": $T, /" >> $O/c23_tiles.jsonl
done
python - <<PY
import json
rows = [json.loads(l) for l in open('$O/c23_tiles.jsonl')]
for r in rows: print(r['tile_edge'], r['mols'], r['ms_per_step_wall'], r['us_per_launch'].get('edge_message'))
PY
