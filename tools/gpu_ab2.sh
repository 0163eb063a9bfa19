R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/flowmol_amd/libflowmol_hip.so
(python $R/tools/ab_bench.py $L 32 32 1024 47 32; python $R/tools/ab_bench.py $L 32 32 1024 47 64; python $R/tools/ab_bench.py $L 64 32 1024 47 32) 2>&1 | grep '^{' | tee $O/ab2.jsonl
python -m pytest $R/tests -m gpu -q 2>&1 | tail -3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc4 -o p4 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $O/pmc4b -o p4b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc4b.log 2>&1
