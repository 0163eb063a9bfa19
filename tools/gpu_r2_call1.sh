# Round-2 call 1: GPU tests, smoke, bench (+ size distribution), multi-rank harness behaviour on a 1-GPU box, ablation A/B of
# the GVP kernels (build_ab/abl_*.so, -DFM_ABLATE=mask) and per-phase cycles of the edge-message kernel.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl
timeout 900 python -m pytest $R/tests -m gpu -q -x 2>&1 | tail -15 > $O/c1_pytest.log
timeout 300 python $R/__graft_entry__.py smoke > $O/c1_smoke.log 2>&1
timeout 600 python $R/bench.py > $O/c1_bench.json 2> $O/c1_bench.err
timeout 300 python $R/bench.py --size-dist geom_full_kekulized --no-cpu-baseline --no-api-e2e > $O/c1_bench_sizedist.json 2> $O/c1_bench_sizedist.err
( timeout 120 python $R/bench.py --gpus 2 --steps 2 --warmup 1; echo "rc=$?" ) > $O/c1_gpus2_on_1gpu.log 2>&1
( FM_BENCH_BACKEND=gloo timeout 300 python $R/bench.py --gpus 2 --steps 4 --warmup 1 --mols-per-gpu 256; echo "rc=$?" ) > $O/c1_gpus2_gloo.log 2>&1
: > $O/c1_ablate.jsonl
for L in $R/build_ab/abl_*.so; do
  timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' >> $O/c1_ablate.jsonl
done
timeout 200 python $R/tools/phase_timing.py $R/build_ab/timing/lib_timing.so 2>&1 | tail -1 > $O/c1_phase.json
cat $O/c1_pytest.log; tail -1 $O/c1_smoke.log; cut -c1-300 $O/c1_bench.json; tail -3 $O/c1_gpus2_on_1gpu.log; tail -2 $O/c1_gpus2_gloo.log | cut -c1-300
python - <<PY
import json
for l in open('$O/c1_ablate.jsonl'):
    d = json.loads(l); print(d['lib'], d['eval_ms'], d['kernels_us'])
PY
cat $O/c1_phase.json
