#!/usr/bin/env python
"""Benchmark of the FlowMol3 sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): the flowmol3
GEOM-drugs architecture, 1024 molecules per GPU of 47 atoms each (the GEOM-drugs mean size; 2,162
directed edges per molecule), n_timesteps = 250, synthetic weights-by-name (no checkpoint ships with the
reference, and there is no network) and synthetic noise.  With N GPUs every rank integrates its own 1024
molecules (weak scaling: configs[3] = 8192 molecules on 8 GPUs) and the packed results are exchanged with
ONE RCCL all-gather.

A "step" is one integration step of the batch: one evaluation of the vector-field network
(EndpointVectorField.forward, incl. self-conditioning) + the Euler/CTMC update.  The timed steps are
consecutive steps of a real trajectory that starts at the prior (the W warm-up steps come first and
contain the bootstrap evaluation).  A 250-timestep sample costs 250 network evaluations (249 steps + 1
bootstrap), hence   molecules/s @250 = global_molecules / (250 * seconds_per_step).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch                                     # noqa: E402
import torch.distributed as dist                 # noqa: E402

FP32_PEAK_TFLOPS = 157.3                         # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
HBM_PEAK_GBS = 8000.0


def conv_message_flops_per_edge(V=32, S=256, F=128, R=32, ncp=4):
    """Algorithmic FLOPs of one GVPConv edge message per directed edge, counting 2*MAC of every
    Linear/einsum exactly as the reference executes them (312,251 MAC for flowmol3)."""
    def gvp(vin, h, vout, sin):
        return vin * h * 3 + vin * 2 * ncp * 3 + (h + ncp) * vout * 3 + (sin + h + ncp) * S + S * vout
    mac = gvp(V + 1, V + 1, V, S + R + F) + 2 * gvp(V, V, V, S)
    return 2 * mac


def network_flops(n, V=32):
    """Reference FLOPs of one network evaluation of one n-atom molecule (flowmol3)."""
    return 4.8744e6 * n * (n - 1) + 6.50e6 * n


def _cpu_steps(cfg, sd, n_atoms_each, B, steps, T, threads):
    """Warm-up step (with the bootstrap evaluation) + `steps` timed integration steps of the CPU oracle."""
    from oracle import cpu_ref
    torch.set_num_threads(threads)
    n_atoms = torch.full((B,), n_atoms_each, dtype=torch.int64)
    batch = cpu_ref.build_batch(n_atoms)
    orc = cpu_ref.OracleVF(cfg, sd)
    torch.manual_seed(1)
    prior = orc.sample_prior(batch)
    t = torch.linspace(0, 1, T)
    alpha_t, alpha_tp = cpu_ref.alpha_tables(t)
    state = {'x_t': prior['x_0'], 'a_t': prior['a_0'], 'c_t': prior['c_0'], 'e_t': prior['e_0']}
    noise = cpu_ref.TorchNoise()
    dst = None
    times = []
    with torch.no_grad():
        for s_idx in range(1, steps + 2):
            t0 = time.perf_counter()
            new, dst = orc.step(batch, state, t[s_idx], t[s_idx - 1], alpha_t[s_idx - 1], alpha_tp[s_idx - 1], prev=dst,
                                eta=cfg.stochasticity, hc_thresh=cfg.high_confidence_threshold, last_step=False, noise=noise)
            state = {k: new[k] for k in ('x_t', 'a_t', 'c_t', 'e_t')}
            times.append(time.perf_counter() - t0)
    return sum(times[1:]) / len(times[1:])


def cpu_baseline(cfg, sd, n_atoms_each, B, steps, T):
    """Time the CPU oracle (the op-for-op restatement of the reference's PyTorch path, oracle/cpu_ref.py) on this
    box's host cores on a bounded sample of the same workload.  The intra-op thread count is chosen by a short
    probe (2 molecules, 1 step per candidate): torch with one thread per logical CPU of a many-core host
    oversubscribes these small operators badly, which would make the baseline look worse than it is."""
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64, 128) if c <= ncpu} | ({ncpu} if ncpu < 8 else set()))
    probe = {}
    for c in cands:
        probe[c] = _cpu_steps(cfg, sd, n_atoms_each, 2, 1, T, c)
    best = min(probe, key=probe.get)
    per_step = _cpu_steps(cfg, sd, n_atoms_each, B, steps, T, best)
    return {'value': B / (T * per_step), 'unit': 'molecules/s', 'cores': best, 'kind': 'port',
            'sample': f'{B} molecules x {n_atoms_each} atoms, {steps} timed integration steps after 1 warm-up step '
                      f'({per_step * 1e3:.0f} ms/step) with {best} torch threads (best of {cands} in a 2-molecule probe; host has '
                      f'{ncpu} logical CPUs), extrapolated linearly to {T} network evaluations per sample',
            'ms_per_step': per_step * 1e3, 'thread_probe_ms_per_step': {str(k): v * 1e3 for k, v in probe.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mols-per-gpu', type=int, default=1024)
    ap.add_argument('--n-atoms', type=int, default=47)
    ap.add_argument('--size-dist', default=None, help="draw the molecule sizes from a shipped training-set histogram (e.g. geom_full_kekulized) instead of --n-atoms; secondary measurement, the headline line uses fixed sizes")
    ap.add_argument('--timesteps', type=int, default=250)
    ap.add_argument('--preset', default='flowmol3')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-mols', type=int, default=8)
    ap.add_argument('--cpu-steps', type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    # one process per GPU; FM_BENCH_BACKEND=gloo lets several ranks share a device to exercise this path on a 1-GPU box
    backend = os.environ.get('FM_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from flowmol_amd import presets, weights, shard
    from flowmol_amd.engine import Engine, IntegrationRun, StepNoise, make_step_plan

    cfg = presets.PRESETS[args.preset]()
    sd = weights.synth_state_dict(cfg, 0)
    eng = Engine(cfg, sd, device=dev)
    B, n, T = args.mols_per_gpu, args.n_atoms, args.timesteps
    def sizes_of(r):
        """Molecule sizes of rank r's shard (fixed size, or a seeded draw from the shipped size histogram)."""
        if args.size_dist is None:
            return torch.full((B,), n, dtype=torch.int64)
        from flowmol_amd.model import load_n_atoms_hist
        vals, counts = load_n_atoms_hist(args.size_dist)
        g_ = torch.Generator().manual_seed(1000 + r)
        return vals[torch.multinomial(counts.double(), B, replacement=True, generator=g_)]

    n_atoms = sizes_of(rank)
    eng.bind(n_atoms)
    N, U, E = eng.N, eng.U, eng.E
    plan = make_step_plan(T, cfg.stochasticity, cfg.high_confidence_threshold, cfg.cat_temperature)
    n_plan = len(plan.scalars)
    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + rank)

    def fresh_state():
        g0 = torch.Generator(device=dev)
        g0.manual_seed(1 + rank)
        x0 = torch.randn(N, 3, device=dev, generator=g0)
        eng.remove_com(x0)
        return eng.prior_state(x0)

    def noise_for_step(i, last):
        return StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, last, dev, generator=gen)

    state = fresh_state()
    run = IntegrationRun(eng, state, plan, noise_for_step)
    pos = 0

    def advance(k):
        """k consecutive steps of the trajectory; a new trajectory starts from the prior when one ends."""
        nonlocal pos, state
        while k > 0:
            if pos >= n_plan:
                state = fresh_state()
                run.reset(state)
                pos = 0
            m = min(k, n_plan - pos)
            run.run(pos, pos + m, chunk=16)
            pos += m
            k -= m

    def gather_all():
        """The single collective of the sampling path: packed results over RCCL/xGMI, every rank gets the whole batch."""
        parts = [torch.arange(r * B, (r + 1) * B) for r in range(world)]
        return shard.gather_results({'x': state['x_t'], 'a': state['a_t'], 'c': state['c_t'], 'e': state['e_t']},
                                    torch.cat([sizes_of(r) for r in range(world)]), parts)

    advance(args.warmup)
    if world > 1:
        gather_all()             # untimed warm-up of the collective (communicator setup, first-use kernel loads)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    advance(args.steps)
    gather_ms = None
    if world > 1:   # inside the timed region: the job is not done until every rank holds the results
        torch.cuda.synchronize(dev)
        tg = time.perf_counter()
        gather_all()
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - tg) * 1e3
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / args.steps
    mols_per_s = B * world / (T * ms_per_step / 1e3)

    # ---- per-kernel timing (HIP events on the launch stream) for the roofline of the dominant kernel
    finite = bool(torch.isfinite(state['x_t']).all().item())
    eng.profile(True)
    advance(2)
    torch.cuda.synchronize(dev)
    kern = {}
    for k in ('edge_message', 'edge_update', 'node_update', 'pos_update', 'node_proj', 'node_proj_asd', 'sc_edge', 'sc_node',
              'edge_head', 'node_head', 'ctmc_pass1', 'ctmc_pass2', 'embed_table', 'gather_ef', 'gather_s', 'remove_com', 'x_step'):
        ms, cnt = eng.profile_get(k)
        if cnt:
            kern[k] = {'avg_us': ms * 1e3 / cnt, 'launches_per_step': cnt / 2}
    eng.profile(False)
    traffic = None
    try:     # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (same workload only)
        tj = json.loads((ROOT / 'profiles' / 'r01s_traffic.json').read_text())
        if tj['mols_per_gpu'] == B and tj['n_atoms'] == n and args.preset == 'flowmol3' and args.size_dist is None:
            traffic = tj['hbm_bytes_per_launch']
    except Exception:
        pass
    roofline = None
    if 'edge_message' in kern:
        flops = conv_message_flops_per_edge(cfg.n_vec_channels) * E
        ach = flops / (kern['edge_message']['avg_us'] * 1e-6) / 1e12
        roofline = {'bound': 'mfma', 'kernel': 'fm_k_edge_message', 'achieved': ach, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': ach / FP32_PEAK_TFLOPS, 'traffic': traffic,
                    'traffic_note': 'HBM bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/r01s_traffic.json (rocprofv3 PMC, gfx950 FETCH correction); '
                                    'algorithmic compulsory bytes/launch = E*(512+8) + partial sums = 1.29e9',
                    'avg_launch_us': kern['edge_message']['avg_us'],
                    'hbm_gb_per_s': (traffic / (kern['edge_message']['avg_us'] * 1e-6) / 1e9) if traffic else None,
                    'hbm_frac_of_8tb_per_s': (traffic / (kern['edge_message']['avg_us'] * 1e-6) / 8e12) if traffic else None,
                    'algorithmic_flop_per_launch': flops,
                    'note': 'algorithmic FLOPs = 2*312,251 MAC per directed edge (reference-executed count) x E edges per launch; '
                            'peak = f32-input MFMA (v_mfma_f32_16x16x4_f32)'}
    whole = 250 * sum(network_flops(int(k)) for k in n_atoms.tolist()) / B * mols_per_s / world / 1e12
    out = {
        'metric': 'molecules/sec at 250 timesteps (GEOM-drugs-sized graphs)', 'value': mols_per_s, 'unit': 'molecules/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{args.preset} GEOM-drugs model, {B} molecules/GPU x ' + (f'{n} atoms' if args.size_dist is None else f'sizes ~ {args.size_dist} histogram (mean {float(n_atoms.double().mean()):.1f}, max {int(n_atoms.max())})') + f', n_timesteps={T} '
                               f'(BASELINE.json configs[2]; configs[3] at 8 GPUs)',
                   'global_molecules': B * world, 'nodes_per_gpu': N, 'directed_edges_per_gpu': E, 'parallelism': f'molecule-shard x{world}',
                   'step': 'one integration step = 1 network evaluation + Euler/CTMC update of the whole batch',
                   'value_formula': 'global_molecules / (n_timesteps * ms_per_step/1000)', 'weights': 'synthetic by name (seed 0)',
                   'finite': finite},
        'network_eval_ms': ms_per_step, 'final_gather_ms': gather_ms,
        'whole_path_fp32_tflops_per_gpu': whole, 'whole_path_frac_of_fp32_peak': whole / FP32_PEAK_TFLOPS,
        'kernels': kern,
    }
    if roofline:
        out['roofline'] = roofline
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, sd, n, args.cpu_mols, args.cpu_steps, T)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
