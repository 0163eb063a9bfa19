#!/bin/bash
# ONE parameterised driver for everything that runs on the GPU box (replaces the per-call scripts of rounds 1-2).
#
#   gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <tag> <step> [+ <step> ...]'
#
# <tag> prefixes every file written under gpurun_out/.  Steps (each with its own timeout, outputs in gpurun_out/<tag>_*):
#   tests [pytest args]          python -m pytest tests -m gpu -q [args]
#   smoke                        __graft_entry__.smoke()
#   bench <name> [bench args]    python bench.py [args]        -> <tag>_bench_<name>.json
#   ab <reps> <lib>... [-- ab_bench args]    tools/ab_bench.py on every library, alternating, <reps> times -> <tag>_ab.jsonl
#   latency [latency_sweep args] tools/latency_sweep.py       -> <tag>_latency.jsonl
#   profile <name> [bench args]  rocprofv3 --kernel-trace --stats, then three separate PMC passes (SQ / FETCH / WRITE+TCC) of
#                                `python bench.py <args> --no-cpu-baseline --no-api-e2e --no-secondary`   -> <tag>_prof_<name>/{stats,pmc_sq,pmc_fetch,pmc_write}
#   pmc <name> "<counters>" [bench args]     one extra PMC pass with the given counters -> <tag>_prof_<name>/pmc_extra
#   py <script> [args]           python <script> [args]       -> <tag>_py.log (appended)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
TAG=$1; shift
run_step() {
  local s=$1; shift
  case $s in
    tests)   rm -f $O/parity_report.jsonl
             timeout 1700 python -m pytest $R/tests -m gpu -q "$@" 2>&1 | tail -25 > $O/${TAG}_pytest.log; cat $O/${TAG}_pytest.log | tail -8
             [ -f $O/parity_report.jsonl ] && cp $O/parity_report.jsonl $O/${TAG}_parity_report.jsonl ;;
    smoke)   timeout 300 python $R/__graft_entry__.py smoke > $O/${TAG}_smoke.log 2>&1; tail -1 $O/${TAG}_smoke.log ;;
    bench)   local n=$1; shift
             timeout 900 python $R/bench.py "$@" > $O/${TAG}_bench_$n.json 2> $O/${TAG}_bench_$n.err; cut -c1-600 $O/${TAG}_bench_$n.json; tail -2 $O/${TAG}_bench_$n.err ;;
    ab)      local reps=$1; shift; local libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
             for r in $(seq $reps); do for L in "${libs[@]}"; do
               timeout 300 python $R/tools/ab_bench.py --lib $R/$L "$@" 2>&1 | grep '^{' | tee -a $O/${TAG}_ab.jsonl | cut -c1-700
             done; done ;;
    latency) timeout 400 python $R/tools/latency_sweep.py "$@" 2>&1 | grep '^{' | tee -a $O/${TAG}_latency.jsonl | cut -c1-500 ;;
    profile) local n=$1; shift; local D=$O/${TAG}_prof_$n; rm -rf $D; mkdir -p $D
             local B="python $R/bench.py $* --no-cpu-baseline --no-api-e2e --no-secondary"
             timeout 700 rocprofv3 --kernel-trace --stats -d $D/stats -o s -- $B > $D/stats.log 2>&1
             timeout 700 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $D/pmc_sq -o p -- $B > $D/pmc_sq.log 2>&1
             timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $D/pmc_fetch -o p -- $B > $D/pmc_fetch.log 2>&1
             timeout 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $D/pmc_write -o p -- $B > $D/pmc_write.log 2>&1
             for k in stats pmc_sq pmc_fetch pmc_write; do
               db=$(find $D/$k -name '*_results.db' | tail -1)
               [ -n "$db" ] && python $R/tools/summarize_rocprof.py $([ $k == stats ] && echo stats || echo pmc) $db $([ $k == stats ] || echo fm_k_) > $D/$k.txt 2>&1
               find $D/$k -name '*_results.db' -delete          # summaries travel back, the databases (tens of MB) do not
             done; head -30 $D/stats.txt ;;
    pmc)     local n=$1; local ctr=$2; shift 2; local D=$O/${TAG}_prof_$n; mkdir -p $D; local x=pmc_extra_$(echo $ctr | tr ' ' '_' | cut -c1-40)
             timeout 700 rocprofv3 --kernel-trace --pmc $ctr -d $D/$x -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-api-e2e --no-secondary > $D/$x.log 2>&1
             db=$(find $D/$x -name '*_results.db' | tail -1); [ -n "$db" ] && python $R/tools/summarize_rocprof.py pmc $db fm_k_ > $D/$x.txt 2>&1
             find $D/$x -name '*_results.db' -delete; head -40 $D/$x.txt ;;
    py)      local sc=$1; shift; [ -f "$R/$sc" ] && sc="$R/$sc"          # the steps run from /tmp: a path relative to the repository root is resolved here
             timeout 900 python "$sc" "$@" 2>&1 | tee -a $O/${TAG}_py.log | tail -40 ;;
    *)       echo "unknown step $s"; return 2 ;;
  esac
}
args=(); for a in "$@" +; do
  if [ "$a" == "+" ]; then [ ${#args[@]} -gt 0 ] && { echo "=== ${args[*]}"; run_step "${args[@]}"; }; args=(); else args+=("$a"); fi
done
