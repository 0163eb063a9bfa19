R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
timeout 120 $R/build_ab/valu_cost > $O/c5_valu_cost.txt 2>&1
cat $O/c5_valu_cost.txt
