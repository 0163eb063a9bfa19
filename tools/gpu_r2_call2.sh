# Round-2 call 2: GPU tests (incl. dev.yml models, Philox mode), bench, latency sweep (fused vs round-1 launch sequence, torch vs in-kernel noise),
# 2-rank harness run on the shared device
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl
timeout 1200 python -m pytest $R/tests -m gpu -q -x 2>&1 | tail -15 > $O/c2_pytest.log
timeout 300 python $R/__graft_entry__.py smoke > $O/c2_smoke.log 2>&1
timeout 600 python $R/bench.py > $O/c2_bench.json 2> $O/c2_bench.err
( FM_BENCH_BACKEND=gloo timeout 300 python $R/bench.py --gpus 2 --steps 4 --warmup 1 --mols-per-gpu 256 --no-cpu-baseline; echo "rc=$?" ) > $O/c2_gpus2_gloo.log 2>&1
: > $O/c2_latency.jsonl
FM_FUSE_NODE=0 timeout 300 python $R/tools/latency_sweep.py 1 8 32 128 >> $O/c2_latency.jsonl 2>/dev/null
timeout 300 python $R/tools/latency_sweep.py 1 8 32 128 1024 >> $O/c2_latency.jsonl 2>/dev/null
timeout 300 python $R/tools/latency_sweep.py 1 8 32 128 1024 philox >> $O/c2_latency.jsonl 2>/dev/null
FM_FUSE_NODE=0 timeout 200 python $R/bench.py --steps 20 --no-cpu-baseline --no-api-e2e > $O/c2_bench_unfused.json 2>/dev/null
cat $O/c2_pytest.log; tail -1 $O/c2_smoke.log; cut -c1-260 $O/c2_bench.json; echo; cut -c1-260 $O/c2_bench_unfused.json; echo; tail -2 $O/c2_gpus2_gloo.log | cut -c1-400; cat $O/c2_latency.jsonl
