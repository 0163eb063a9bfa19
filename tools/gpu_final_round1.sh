# Round-1 final measurement: GPU tests, smoke, bench, rocprofv3 stats + PMC (SQ, FETCH, WRITE) of the same bench command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python -m pytest $R/tests -m gpu -q 2>&1 | tail -4 > $O/final_pytest.log
python $R/__graft_entry__.py smoke > $O/final_smoke.log 2>&1
python $R/bench.py > $O/final_bench.json 2> $O/final_bench.err
rocprofv3 --kernel-trace --stats -d $O/fprof -o f -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/fprof.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/fpmc1 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fpmc1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fpmc2 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fpmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/fpmc3 -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fpmc3.log 2>&1
cat $O/final_pytest.log; tail -1 $O/final_smoke.log; cut -c1-400 $O/final_bench.json
