"""world_size-2 gloo test of the multi-GPU path's host logic: molecule sharding (LPT) and the single
all-gather of packed results restoring the original molecule order.  (The data path has no other collective.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flowmol_amd.shard import gather_results, partition_lpt


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_atoms, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    parts = partition_lpt(n_atoms, world)
    mine = parts[rank]
    # fabricate per-molecule results that encode the ORIGINAL molecule index
    xs, as_, cs, es = [], [], [], []
    for i in mine.tolist():
        n = int(n_atoms[i])
        u = n * (n - 1) // 2
        xs.append(torch.full((n, 3), float(i)) + torch.arange(n).float()[:, None] * 0.01)
        as_.append(torch.full((n,), i % 11, dtype=torch.int32))
        cs.append(torch.full((n,), i % 6, dtype=torch.int32))
        es.append(torch.full((u,), i % 5, dtype=torch.int32))
    i32 = dict(dtype=torch.int32)
    local = ({'x': torch.cat(xs), 'a': torch.cat(as_), 'c': torch.cat(cs), 'e': torch.cat(es)} if xs else
             {'x': torch.zeros(0, 3), 'a': torch.zeros(0, **i32), 'c': torch.zeros(0, **i32), 'e': torch.zeros(0, **i32)})      # a rank that owns nothing
    full = gather_results(local, n_atoms, parts)
    ok = True
    noff = poff = 0
    for i, n in enumerate(n_atoms.tolist()):
        u = n * (n - 1) // 2
        ok &= bool(torch.allclose(full['x'][noff:noff + n, 0], torch.full((n,), float(i)) + torch.arange(n).float() * 0.01))
        ok &= bool((full['a'][noff:noff + n] == i % 11).all()) and bool((full['e'][poff:poff + u] == i % 5).all())
        noff += n
        poff += u
    q.put((rank, ok, int(full['x'].shape[0])))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('sizes,world', [([5, 47, 12, 30, 8, 64, 3, 21, 47], 2),
                                         ([3, 2], 2),      # packed payloads of 45 and 29 bytes: slot size must be padded for the fp32 view
                                         ([5, 47, 12, 30, 8, 64, 3, 21, 47], 3),      # more than two ranks, parts of unequal length
                                         ([9, 4, 33, 17, 2, 61, 47, 5, 12, 3, 26], 8),      # the node's rank count
                                         ([7, 3], 4)])                                # two ranks own nothing: empty payloads
def test_shard_and_gather(sizes, world):
    n_atoms = torch.tensor(sizes)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_atoms, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res), res
    assert all(n == int(n_atoms.sum()) for _, _, n in res)


def _sample_worker(rank, world, port, sizes, q, noise='per_rank', preset='qm9'):
    """Each rank: emulated engine on the CPU, its own RNG stream (or the same one in replicated-noise mode),
    sample_distributed over gloo."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pathlib import Path
    import flowmol_amd as flowmol
    from flowmol_amd import _lib
    emu = _lib.load(Path(__file__).resolve().parent / 'emu' / 'libflowmol_emu.so')
    model = flowmol.FlowMol.from_preset(preset, _engine_lib=emu).to('cpu')
    torch.manual_seed(100 + (rank if noise == 'per_rank' else 0))
    full, n = model.sample_distributed(torch.tensor(sizes), n_timesteps=3, return_tensors=True, noise=noise)
    q.put((rank, {k: v.numpy().copy() for k, v in full.items()}))    # plain arrays: torch's fd-based tensor sharing needs the producer alive
    dist.destroy_process_group()


def test_sample_distributed_world2_matches_per_rank_runs(emu_lib_path):
    """FlowMol.sample_distributed on 2 gloo ranks (emulated kernels): every rank gets the whole batch in the caller's
    order, and each molecule equals what its owning rank computes alone with the same seed."""
    from pathlib import Path
    import flowmol_amd as flowmol
    from flowmol_amd import _lib
    from flowmol_amd.shard import partition_lpt
    emu_path = emu_lib_path
    sizes = [4, 6, 3, 5]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sample_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in (q.get(timeout=300) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
    for k in 'xace':
        assert torch.equal(res[0][k], res[1][k])            # both ranks hold the same gathered batch
    n_atoms = torch.tensor(sizes)
    parts = partition_lpt(n_atoms, 2)
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=_lib.load(emu_path)).to('cpu')
    node_off = torch.cumsum(n_atoms, 0) - n_atoms
    for r in range(2):
        torch.manual_seed(100 + r)
        alone, _ = model.sample(n_atoms[parts[r]], n_timesteps=3, return_tensors=True)
        o = 0
        for i in parts[r].tolist():
            n = sizes[i]
            assert torch.equal(res[0]['a'][node_off[i]:node_off[i] + n], alone['a'][o:o + n])
            assert torch.allclose(res[0]['x'][node_off[i]:node_off[i] + n], alone['x'][o:o + n])
            o += n


def _cli_worker(rank, world, port, out_path):
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'LOCAL_RANK': str(rank),
                       'WORLD_SIZE': str(world)})
    from pathlib import Path
    from flowmol_amd import _lib, cli
    emu = _lib.load(Path(__file__).resolve().parent / 'emu' / 'libflowmol_emu.so')
    args = cli.parse_args(['--preset', 'qm9', '--n_mols', '5', '--n_timesteps', '2', '--max_batch_size', '3', '--seed', '3',
                           '--device', 'cpu', '--output_file', out_path])
    cli.run(args, engine_lib=emu)
    dist.destroy_process_group()


def test_cli_under_two_ranks_writes_once(tmp_path, emu_lib_path):
    """The CLI launched as two ranks (torchrun-style environment, gloo on the CPU emulation): sizes drawn on rank 0 are
    broadcast, each batch is sharded, rank 0 writes all molecules exactly once."""
    out = tmp_path / 'dist.sdf'
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, str(out))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert out.read_text().count('$$$$') == 5


@pytest.mark.parametrize('preset,world', [('qm9', 2), ('endpoint_small', 2), ('qm9', 3), ('qm9', 5)])
def test_sample_distributed_replicated_noise_equals_single_process(emu_lib_path, preset, world):
    """Parity mode of the sharded path: with every rank drawing the full batch's noise from the same seed, the ranks
    reproduce the single-process sample(n_atoms): identical tokens, coordinates to summation order (molecules are
    independent: SURVEY.md §8e).  For an endpoint-parameterised model the only randomness is the priors, drawn for the full
    batch on every rank and sliced.  Three ranks: parts of unequal length; five ranks for four molecules: one rank owns nothing
    and still takes part in the gather."""
    import flowmol_amd as flowmol
    from flowmol_amd import _lib
    sizes = [4, 6, 3, 5]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sample_worker, args=(r, world, port, sizes, q, 'replicated', preset)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in (q.get(timeout=300) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
    model = flowmol.FlowMol.from_preset(preset, _engine_lib=_lib.load(emu_lib_path)).to('cpu')
    torch.manual_seed(100)
    single, _ = model.sample(torch.tensor(sizes), n_timesteps=3, return_tensors=True)
    for r in range(world):
        for k in 'ace':
            assert torch.equal(res[r][k], single[k].to(res[r][k].dtype))
        assert torch.equal(res[r]['x'], single['x'])      # canonical arithmetic: a molecule's bits do not depend on the shard it was computed in


def _philox_worker(rank, world, port, sizes, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pathlib import Path
    import flowmol_amd as flowmol
    from flowmol_amd import _lib
    emu = _lib.load(Path(__file__).resolve().parent / 'emu' / 'libflowmol_emu.so')
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=emu).to('cpu')
    torch.manual_seed(55 + rank)           # different per-rank RNG state on purpose: only rank 0's broadcast seed matters
    full, n = model.sample_distributed(torch.tensor(sizes), n_timesteps=3, return_tensors=True, noise='philox')
    q.put((rank, {k: v.numpy().copy() for k, v in full.items()}))
    dist.destroy_process_group()


def test_sample_distributed_philox_is_independent_of_the_world_size(emu_lib_path):
    """SURVEY.md §8e performance mode: with per-molecule Philox streams two ranks produce what one process produces with
    the same seed -- identical tokens, coordinates to summation order -- while each rank generates only its own shard's noise."""
    import flowmol_amd as flowmol
    from flowmol_amd import _lib
    sizes = [4, 6, 3, 5]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_philox_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: {k: torch.from_numpy(v) for k, v in d.items()} for r, d in (q.get(timeout=300) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
    for k in 'xace':
        assert torch.equal(res[0][k], res[1][k])
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=_lib.load(emu_lib_path)).to('cpu')
    torch.manual_seed(55)                  # rank 0's generator state -> the same broadcast seed
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    single, _ = model.sample(torch.tensor(sizes), n_timesteps=3, return_tensors=True, rng='philox', _philox=seed)
    for k in 'ace':
        assert torch.equal(res[0][k], single[k].to(res[0][k].dtype))
    assert torch.equal(res[0]['x'], single['x'])          # canonical arithmetic: bit for bit on 1 or 2 ranks


def _traj_worker(rank, world, port, sizes, q, preset, with_prior):
    """sample_distributed with everything one sample() call can do: trajectory frames (second gather) and a caller-supplied prior (sliced per rank)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from pathlib import Path
        import flowmol_amd as flowmol
        from flowmol_amd import _lib
        emu = _lib.load(Path(__file__).resolve().parent / 'emu' / 'libflowmol_emu.so')
        model = flowmol.FlowMol.from_preset(preset, _engine_lib=emu).to('cpu')
        n_atoms = torch.tensor(sizes)
        prior = _reference_prior(model.cfg, n_atoms) if with_prior else None
        torch.manual_seed(100)
        full, n, frames = model.sample_distributed(n_atoms, n_timesteps=4, return_tensors=True, noise='replicated', xt_traj=True, ep_traj=True, prior=prior)
        torch.manual_seed(100)
        mols = model.sample_distributed(n_atoms, n_timesteps=4, noise='replicated', xt_traj=True, ep_traj=True, prior=prior)
        blocks = [len(m.traj_mol_blocks(ep_traj=False)) for m in mols] + [len(m.traj_mol_blocks(ep_traj=True)) for m in mols]
        q.put((rank, {k: v.numpy().copy() for k, v in full.items()}, {k: v.numpy().copy() for k, v in frames.items()}, blocks))
    except Exception as e:
        q.put((rank, repr(e), None, None))
    finally:
        dist.destroy_process_group()


def _reference_prior(cfg, n_atoms):
    """A prior dict in the reference's format (flowmol.py:534-545) for a CTMC model: centred positions, mask one-hots, e_0 per directed edge."""
    from oracle import cpu_ref
    batch = cpu_ref.build_batch(n_atoms)
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn(batch.N, 3, generator=g)
    x0 = x0 - cpu_ref.segment_mean(x0, batch.node_batch_idx, batch.B)[batch.node_batch_idx]
    return {'x_0': x0, 'a_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_atom_types), 'c_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_charges),
            'e_0': cpu_ref.edge_prior(batch.upper_edge_mask, cfg.n_bond_types), 'fake_atoms': True}


@pytest.mark.parametrize('preset,world,with_prior', [('qm9', 2, False), ('qm9', 3, True), ('endpoint_small', 2, False)])
def test_sample_distributed_trajectories_and_priors_equal_single_process(emu_lib_path, preset, world, with_prior):
    """the sharded path does everything one sample() call of the reference does (flowmol.py:489-493,534-545,564-589): xt_traj / ep_traj
    (every frame of every molecule, gathered with a second all-gather) and a caller-supplied prior (sliced per rank) -- frames, final state and
    the packaged molecules' trajectory lengths equal the single-process sample() of the same seed BIT FOR BIT; three ranks for four molecules
    incl. unequal parts."""
    import flowmol_amd as flowmol
    from flowmol_amd import _lib
    sizes = [4, 6, 3, 5]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_traj_worker, args=(r, world, port, sizes, q, preset, with_prior)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert not [g_ for g_ in got if isinstance(g_[1], str)], got
    model = flowmol.FlowMol.from_preset(preset, _engine_lib=_lib.load(emu_lib_path)).to('cpu')
    n_atoms = torch.tensor(sizes)
    prior = _reference_prior(model.cfg, n_atoms) if with_prior else None
    torch.manual_seed(100)
    mols = model.sample(n_atoms, n_timesteps=4, xt_traj=True, ep_traj=True, prior=prior)
    pairs = [n * (n - 1) // 2 for n in sizes]
    for rank, full, frames, blocks in got:
        assert blocks == [4] * len(sizes) + [3] * len(sizes)                      # T frames / T - 1 endpoint frames per molecule
        no = po = 0
        for i, m in enumerate(mols):
            n, u = sizes[i], pairs[i]
            assert torch.equal(torch.from_numpy(full['x'][no:no + n]), m.x_1) and torch.equal(torch.from_numpy(full['a'][no:no + n]).long(), m.a_1.long())
            for k, v in m.traj_frames.items():
                lo, w = (po, u) if k.startswith('e') else (no, n)
                assert torch.equal(torch.from_numpy(frames[k][:, lo:lo + w]).to(v.dtype), v), (rank, i, k)
            no += n; po += u


def _bench_parity_worker(rank, world, port, sizes, q):
    """bench.py's multi-GPU self-check (multi_gpu_parity) on `world` gloo ranks over the emulated kernels."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import importlib.util
        from pathlib import Path
        from flowmol_amd import _lib, presets, weights
        from flowmol_amd.engine import Engine
        root = Path(__file__).resolve().parent.parent
        spec = importlib.util.spec_from_file_location('bench_mod_par', root / 'bench.py')
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        cfg = presets.qm9()
        sd = weights.synth_state_dict(cfg, 0)
        eng = Engine(cfg, sd, device='cpu', lib=_lib.load(root / 'tests' / 'emu' / 'libflowmol_emu.so'))
        q.put((rank, bench.multi_gpu_parity(cfg, sd, eng, world, rank, torch.device('cpu'), 'gloo', n_atoms=torch.tensor(sizes), T=3)))
    except Exception as e:
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_bench_multi_gpu_parity_block_on_two_gloo_ranks(emu_lib_path):
    """The block `bench.py --gpus N` (N > 1) runs before its timed region and emits as `multi_gpu_parity`, here on two gloo ranks
    over the emulated kernels: sample_distributed in the replicated and the Philox noise mode equals the single-process sample on rank 0 --
    0 differing tokens -- every rank holds the same gathered batch and receives the same verdict."""
    sizes = [4, 6, 3, 5, 2]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_parity_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert all(isinstance(v, dict) for v in res.values()), res
    d = res[0]
    assert res[1] == d                                                    # the verdict is broadcast: every rank decides the same
    for k in ('token_diffs', 'x_rel', 'philox_token_diffs', 'world_size', 'distinct_pci_devices', 'rccl_version', 'all_gather_bytes', 'ranks_hold_the_same_batch', 'ok'):
        assert k in d, k
    assert d['ok'] and d['token_diffs'] == 0 and d['philox_token_diffs'] == 0 and d['x_rel'] < 1e-4 and d['world_size'] == 2
    assert d['molecules'] == 5 and sum(d['molecules_per_rank']) == 5 and d['all_gather_bytes'] == 2 * d['all_gather_slot_bytes']
    assert d['tokens_compared'] == 2 * sum(sizes) + sum(n * (n - 1) // 2 for n in sizes)
