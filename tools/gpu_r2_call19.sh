# large-batch sanity: 8192 molecules x 47 atoms on ONE GPU (BASELINE configs[3]'s whole job on a single device): throughput holds, nothing overflows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 python $R/bench.py --mols-per-gpu 8192 --steps 6 --warmup 1 --no-cpu-baseline --no-api-e2e > $O/c19_bench_8192.json 2> $O/c19_bench_8192.err; echo "rc=$?"
cut -c1-260 $O/c19_bench_8192.json; tail -3 $O/c19_bench_8192.err
timeout 600 python - <<'PY' 2>&1 | tail -5
import os, sys; sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, flowmol_amd as flowmol
m = flowmol.FlowMol.from_preset('flowmol3').cuda().eval()
torch.manual_seed(0)
out, n = m.sample(torch.full((8192,), 47), n_timesteps=6, return_tensors='device')
torch.cuda.synchronize()
print('8192 x 47, 6 timesteps:', {k: tuple(v.shape) for k, v in out.items()}, 'finite', bool(torch.isfinite(out['x']).all()),
      'masks left', int((out['a'] == m.cfg.n_atom_types).sum()), 'workspace GB', m.engine.workspace_bytes / 1e9, 'mem GB', torch.cuda.max_memory_allocated() / 1e9)
# the same molecules in 8 chunks of 1024 with per-molecule Philox noise must give the same tokens (sharding-independent noise)
torch.manual_seed(1)
a = m.sample(torch.full((2048,), 47), n_timesteps=5, return_tensors='device', rng='philox')[0]
torch.manual_seed(1)
b1 = m.sample(torch.full((1024,), 47), n_timesteps=5, return_tensors='device', rng='philox')[0]
print('philox 2048 vs first 1024: tokens equal', bool((a['a'][:1024 * 47] == b1['a']).all()), 'max |dx|', float((a['x'][:1024 * 47] - b1['x']).abs().max()))
PY
