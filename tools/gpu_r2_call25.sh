# 16-row MLP / node-projection tiles at small batches: latency sweep (auto) vs forced 64-row tiles, then the full GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/c25_latency.jsonl
FM_MLP_SMALL_TILES=0 timeout 300 python $R/tools/latency_sweep.py 1 8 32 128 philox 2>&1 | grep "^{" | sed 's/^{/{"mlp_tiles": 64, /' >> $O/c25_latency.jsonl
timeout 300 python $R/tools/latency_sweep.py 1 4 8 16 32 64 128 1024 philox 2>&1 | grep "^{" | sed 's/^{/{"mlp_tiles": "auto", /' >> $O/c25_latency.jsonl
python - <<PY
import json
for l in open('$O/c25_latency.jsonl'):
    d = json.loads(l); u = d['us_per_launch']; print(d['mlp_tiles'], d['mols'], d['ms_per_step_wall'], {k: u[k] for k in u if k in ('sc', 'heads', 'sc_node', 'sc_edge', 'node_head', 'edge_head', 'node_proj')})
PY
timeout 1500 python -m pytest $R/tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -4
