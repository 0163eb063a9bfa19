R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python -m pytest $R/tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu2.log
rocprofv3 -L > $O/counters.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r1 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_prof.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc1 -o p1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc1b -o p1b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc1b.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc2 -o p2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc3 -o p3 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc3.log 2>&1
python $R/bench.py --steps 10 --warmup 3 > $O/bench2.log 2> $O/bench2.err
cat $O/pytest_gpu2.log; tail -2 $O/bench2.log | cut -c1-600; ls -la $O/prof_stats $O/pmc1 $O/pmc2 2>&1 | head -30; du -sh $O
