R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout 60 $R/build_ab/mfma_bf16_layout > $O/c6_layout.txt 2>&1; cat $O/c6_layout.txt
