# A/B of prebuilt library variants under build_ab/ (+ the full GPU test-suite on the in-tree library)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python -m pytest $R/tests -m gpu -q 2>&1 | tail -4 | tee $O/ab3_pytest.log
: > $O/ab3.jsonl
for L in $R/build_ab/*.so; do
  python $R/tools/ab_bench.py $L 32 32 1024 47 32 2>&1 | grep '^{' | sed "s|^{|{\"lib\": \"$(basename $L)\", |" | tee -a $O/ab3.jsonl
done
if [ -f $R/build_ab/timing/lib_timing.so ]; then python $R/tools/phase_timing.py $R/build_ab/timing/lib_timing.so 2>&1 | tail -1 | tee $O/ab3_phase.json; fi
