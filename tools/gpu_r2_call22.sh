# the CLI end to end on the GPU (outputs under /tmp, every command under its own timeout): a C5-like run (geom_full_kekulized model, T=500,
# per-molecule trajectory files; 16 molecules -- synthetic weights give dense random bonds, so the files are far larger than real ones) and a
# plain 1024-molecule flowmol3 run with metrics; wall-clock as the reference's test.py reports it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O /tmp/cli; cd $R; export TMPDIR=/tmp
( time timeout 240 python -m flowmol_amd.cli --preset geom_ctmc --n_mols 16 --n_timesteps 500 --xt_traj --seed 0 --max_batch_size 16 --output_file /tmp/cli/c5.sdf ) > $O/c22_cli_c5.log 2>&1
ls /tmp/cli | wc -l >> $O/c22_cli_c5.log; du -sh /tmp/cli >> $O/c22_cli_c5.log
( time timeout 240 python -m flowmol_amd.cli --preset flowmol3 --n_mols 1024 --n_timesteps 250 --seed 0 --max_batch_size 1024 --metrics --output_file /tmp/cli/fm3.sdf ) > $O/c22_cli_fm3.log 2>&1
grep -v "^$" $O/c22_cli_c5.log | tail -10; grep -v "^$" $O/c22_cli_fm3.log | tail -12
rm -rf /tmp/cli
