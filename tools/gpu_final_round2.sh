# Round-2 final measurement: GPU tests, smoke, bench (+ size distribution, + 2-rank harness run), rocprofv3 stats + PMC (SQ, FETCH, WRITE) of the same bench command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl; rm -rf $O/fprof $O/fpmc1 $O/fpmc2 $O/fpmc3
timeout 1500 python -m pytest $R/tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -4 > $O/final_pytest.log
timeout 300 python $R/__graft_entry__.py smoke > $O/final_smoke.log 2>&1
timeout 900 python $R/bench.py > $O/final_bench.json 2> $O/final_bench.err
timeout 300 python $R/bench.py --size-dist geom_full_kekulized --no-cpu-baseline --no-api-e2e > $O/final_bench_sizedist.json 2>/dev/null
timeout 300 python $R/bench.py --precision bf16x3 --no-cpu-baseline --no-api-e2e > $O/final_bench_sp.json 2>/dev/null
( FM_BENCH_BACKEND=gloo timeout 300 python $R/bench.py --gpus 2 --steps 8 --warmup 2 --mols-per-gpu 256 --no-cpu-baseline; echo "rc=$?" ) > $O/final_gpus2_gloo.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/fprof -o f -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-api-e2e > $O/fprof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/fpmc1 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api-e2e > $O/fpmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fpmc2 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api-e2e > $O/fpmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/fpmc3 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api-e2e > $O/fpmc3.log 2>&1
cat $O/final_pytest.log; tail -1 $O/final_smoke.log; cut -c1-400 $O/final_bench.json; echo; tail -2 $O/final_gpus2_gloo.log | cut -c1-300; ls $O/fprof $O/fpmc1 $O/fpmc2 $O/fpmc3 | head -20
