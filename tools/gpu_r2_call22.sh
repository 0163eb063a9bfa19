# the CLI end to end on the GPU: BASELINE config C5 (geom_full_kekulized model, 128 molecules, T=500, per-molecule trajectory files) and a
# plain 1024-molecule flowmol3 run with metrics; wall-clock as the reference's test.py reports it (sampling_time) + total
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/cli; cd $R; export TMPDIR=/tmp
( time python -m flowmol_amd.cli --preset geom_ctmc --n_mols 128 --n_timesteps 500 --xt_traj --seed 0 --max_batch_size 128 --output_file $O/cli/c5.sdf ) > $O/c22_cli_c5.log 2>&1
ls $O/cli | wc -l >> $O/c22_cli_c5.log; du -sh $O/cli >> $O/c22_cli_c5.log
( time python -m flowmol_amd.cli --preset flowmol3 --n_mols 1024 --n_timesteps 250 --seed 0 --max_batch_size 1024 --metrics --output_file $O/cli/fm3.sdf ) > $O/c22_cli_fm3.log 2>&1
grep -v "^$" $O/c22_cli_c5.log | tail -12; grep -v "^$" $O/c22_cli_fm3.log | tail -14
rm -rf $O/cli
