# Round-2 call 3: A/B of the VALU-trimmed edge-message kernel (build_ab/trim_a_base.so = HEAD~, trim_b_new.so = working tree): per-kernel
# times + SQ instruction counters of both; GPU tests on the new library.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl
timeout 1200 python -m pytest $R/tests -m gpu -q -x 2>&1 | tail -6 > $O/c3_pytest.log
: > $O/c3_ab.jsonl
for rep in 1 2; do for L in $R/build_ab/trim_*.so; do
  timeout 200 python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' >> $O/c3_ab.jsonl
done; done
for L in trim_a_base trim_b_new; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $O/c3_pmc_$L -o p -- python $R/tools/ab_bench.py $R/build_ab/$L.so 32 32 1024 47 32 > $O/c3_pmc_$L.log 2>&1
  python $R/tools/summarize_rocprof.py pmc $(ls $O/c3_pmc_$L/*/*_results.db | tail -1) fm_k_edge_message fm_k_edge_update > $O/c3_pmc_$L.txt 2>&1
done
cat $O/c3_pytest.log
python - <<PY
import json
for l in open('$O/c3_ab.jsonl'):
    d = json.loads(l); print(d['lib'], d['eval_ms'], d['kernels_us'], d['parity_out_rel'])
PY
head -30 $O/c3_pmc_trim_a_base.txt; head -30 $O/c3_pmc_trim_b_new.txt
