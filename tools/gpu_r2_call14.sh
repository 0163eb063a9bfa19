# full GPU suite on the ABI-4 tree (arch_variants fixtures), smoke, default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl
timeout 1500 python -m pytest $R/tests -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|FAIL" | tail -15 > $O/c14_pytest_all.log
timeout 300 python $R/__graft_entry__.py smoke > $O/c14_smoke.log 2>&1; echo "smoke rc=$?" >> $O/c14_smoke.log
timeout 900 python $R/bench.py > $O/c14_bench.json 2> $O/c14_bench.err
cat $O/c14_pytest_all.log; tail -2 $O/c14_smoke.log; cut -c1-600 $O/c14_bench.json
grep arch_variants $O/parity_report.jsonl | cut -c1-400
