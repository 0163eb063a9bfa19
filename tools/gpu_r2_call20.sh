R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -f $O/parity_report.jsonl
timeout 900 python -m pytest $R/tests -m gpu -q -k "rounding_sensitivity" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
grep rounding $O/parity_report.jsonl | cut -c1-1500
