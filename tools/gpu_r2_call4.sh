# Round-2 call 4: GPU tests after the endpoint-parameterization work; quick bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl
timeout 1200 python -m pytest $R/tests -m gpu -q -x 2>&1 | tail -6 > $O/c4_pytest.log
timeout 300 python $R/bench.py --steps 20 --no-cpu-baseline --no-api-e2e > $O/c4_bench.json 2> $O/c4_bench.err
cat $O/c4_pytest.log; cut -c1-240 $O/c4_bench.json; tail -3 $O/parity_report.jsonl | cut -c1-300
