# A/B of environment switches on the in-tree library: bash tools/gpu_ab_env.sh "K=V" "K=V K2=V2" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=$R/flowmol_amd/libflowmol_hip.so
: > $O/ab_env.jsonl
for E in "" "$@"; do
  python $R/tools/ab_bench.py $L 32 32 1024 47 32 $E 2>&1 | grep '^{' | tee -a $O/ab_env.jsonl | cut -c1-600
done
