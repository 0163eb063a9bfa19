# embedding tables of a whole chunk of steps in one launch: latency sweep + full GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/tools/latency_sweep.py 1 8 32 128 1024 philox 2>&1 | grep "^{" > $O/c21_latency.jsonl
cat $O/c21_latency.jsonl | cut -c1-420
timeout 1500 python -m pytest $R/tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -4
