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
BF16_PEAK_TFLOPS = 2500.0                        # dense bf16 MFMA (the split-precision mode's matrix instructions)
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


def executed_macs(cfg):
    """MACs the kernels actually issue on the matrix pipe per directed edge / per node and network evaluation (padded GEMM
    shapes of flowmol_amd/csrc, after the algebraic hoists of DESIGN.md §3), next to the reference-executed counts the
    algorithmic figures use.  Returned per kernel so every roofline fraction can be quoted both ways."""
    V, S, F, R = cfg.n_vec_channels, cfg.n_hidden_scalars, cfg.n_hidden_edge_feats, cfg.rbf_dim
    p8 = lambda k: (k + 7) // 8 * 8
    p16 = lambda k: (k + 15) // 16 * 16
    def gvp(first, vout):            # one GVP on one row (fm_gvp_core): [Wh|Wcp] (hoisted for the first edge GVP), Wu, Ws, gates
        vop = max(16, vout)
        return (0 if first else 3 * V * (V + 16)) + 3 * (V + 8) * vop + ((R + F if first else S) + V + 8) * S + S * vop
    msg = gvp(True, V) + 2 * gvp(False, V)
    n_upd = sum(1 for u in cfg.update_schedule() if u >= 0)
    eupd = (F + R) * F + F * F
    sc_e = (p8(cfg.n_bond_types + R) * F + F * F) / 2 if cfg.self_conditioning else 0          # per unordered pair
    head_e = (F * F + F * 16) / 2
    per_edge = cfg.n_convs * msg + n_upd * eupd + sc_e + head_e
    node_upd = 3 * gvp(False, V)
    pos = 2 * gvp(False, V) + gvp(False, 1)
    proj = S * S + 3 * V * (V + 16)
    sc_n = (p8(S + cfg.n_atom_types + cfg.n_charges + R) * S + S * S) if cfg.self_conditioning else 0
    head_n = S * S + S * p16(cfg.n_atom_types + cfg.n_charges)
    per_node = cfg.n_convs * (node_upd + proj) + n_upd * (pos + S * S) + sc_n + head_n
    return {'edge_message_per_edge': msg, 'edge_update_per_edge': eupd, 'per_edge': per_edge, 'per_node': per_node}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU, the
    driver's own launch line) and fail loudly when the node cannot host them."""
    import subprocess
    have = torch.cuda.device_count()
    backend = os.environ.get('FM_BENCH_BACKEND', 'nccl')
    if backend == 'nccl' and have < args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node; refusing to run fewer ranks than requested '
                         f'(FM_BENCH_BACKEND=gloo lets ranks share a device for harness tests only)')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def api_end_to_end(args, B, n, T, dev):
    """Secondary figure: ONE full `model.sample(B x n atoms, n_timesteps=T)` through the drop-in API -- bind, prior, all T-1 steps
    with torch-generated noise, device->host copy and the per-molecule SampledMolecule packaging -- as wall time."""
    import flowmol_amd as flowmol
    model = flowmol.FlowMol.from_preset(args.preset, precision=args.precision).to(dev).eval()
    sizes = torch.full((B,), n, dtype=torch.int64)
    model.sample(sizes[:8], n_timesteps=3)                   # engine creation + first-use costs are not part of the figure
    torch.manual_seed(7)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    mols = model.sample(sizes, n_timesteps=T)
    wall = time.perf_counter() - t0
    timing = dict(getattr(model, 'last_timing', {}))
    model.to('cpu')                                           # releases the second engine's workspace
    return {'wall_s': wall, 'molecules': len(mols), 'molecules_per_s': len(mols) / wall, 'n_timesteps': T,
            'breakdown_s': timing, 'note': 'FlowMol.sample() incl. packaging into SampledMolecule objects (no RDKit in this image)'}


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
    ap.add_argument('--precision', choices=('f32', 'bf16x3'), default='f32',
                    help="arithmetic of the edge-message GEMMs: 'f32' (default, the reference's arithmetic, the headline) or the OPT-IN split precision "
                         "'bf16x3' (f32 operands as hi+lo bf16, three products on the bf16 matrix cores) -- a separately reported mode")
    ap.add_argument('--no-api-e2e', action='store_true', help='skip the secondary end-to-end FlowMol.sample() timing')
    ap.add_argument('--cpu-mols', type=int, default=16)
    ap.add_argument('--cpu-steps', type=int, default=8)
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        _self_launch(args)                 # never returns
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    # one process per GPU; FM_BENCH_BACKEND=gloo lets several ranks share a device to exercise this path on a 1-GPU box
    backend = os.environ.get('FM_BENCH_BACKEND', 'nccl')
    if backend == 'nccl' and torch.cuda.device_count() < world:
        raise SystemExit(f'bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible')
    dev_index = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)       # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f'bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}')
        world = dist.get_world_size()

    from flowmol_amd import presets, weights, shard
    from flowmol_amd.engine import Engine, IntegrationRun, StepNoise, make_step_plan

    cfg = presets.PRESETS[args.preset]()
    sd = weights.synth_state_dict(cfg, 0)
    eng = Engine(cfg, sd, device=dev, precision=args.precision)
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
    own_elapsed = time.perf_counter() - t0          # this rank's steps + its part of the gather, before waiting for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    per_rank_ms = [own_elapsed * 1e3 / args.steps]
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        mine = torch.tensor([own_elapsed * 1e3 / args.steps, gather_ms], device=dev, dtype=torch.float64)
        flat = torch.empty(world * 2, device=dev, dtype=torch.float64)       # concatenated form: accepted by every backend
        dist.all_gather_into_tensor(flat, mine)
        every = flat.view(world, 2)
        per_rank_ms = every[:, 0].tolist()
        gather_ms = float(every[:, 1].max())
    ms_per_step = elapsed * 1e3 / args.steps
    mols_per_s = B * world / (T * ms_per_step / 1e3)

    # ---- per-kernel timing (HIP events on the launch stream) for the roofline of the dominant kernel: a separate
    #      event-instrumented pass of 2 more steps AFTER the timed region (event pairs around every launch add ~1 % to
    #      the step, so the sum of these averages slightly exceeds ms_per_step)
    finite = bool(torch.isfinite(state['x_t']).all().item())
    eng.profile(True)
    advance(2)
    torch.cuda.synchronize(dev)
    kern = {}
    for k in ('edge_message', 'edge_update', 'node_update', 'pos_update', 'node_proj', 'node_proj_asd', 'sc_edge', 'sc_node',
              'edge_head', 'node_head', 'sc', 'heads', 'ctmc', 'ctmc_gat', 'dst_proj', 'embed_table', 'gather_ef', 'gather_s', 'remove_com', 'x_step'):
        ms, cnt = eng.profile_get(k)
        if cnt:
            kern[k] = {'avg_us': ms * 1e3 / cnt, 'launches_per_step': cnt / 2}
    eng.profile(False)
    launches_per_step = sum(v['launches_per_step'] for v in kern.values())
    # counters of the dominant kernel from the committed rocprofv3 PMC passes (same workload only); never measured by this run
    pmc = None
    try:
        pj = json.loads((ROOT / 'profiles' / 'current_pmc.json').read_text())
        if pj['mols_per_gpu'] == B and pj['n_atoms'] == n and pj['preset'] == args.preset and args.size_dist is None:
            pmc = pj
    except Exception:
        pass
    ex = executed_macs(cfg)
    roofline = None
    if 'edge_message' in kern and args.precision == 'bf16x3':
        # opt-in mode: the scalar and gate GEMMs issue 3 bf16 products per term on padded K (7 / 10 / 10 k32 blocks); the vector path stays f32
        us = kern['edge_message']['avg_us']
        V = cfg.n_vec_channels
        ku0 = (V + 1 + 4 + 7) // 8 * 8
        kb = [(160 + ku0 + 31) // 32, (256 + V + 8 + 31) // 32, (256 + V + 8 + 31) // 32]
        bf16_mac = 3 * (sum(k * 32 * 256 for k in kb) + 3 * 256 * V)
        f32_mac = 3 * (V + 8) * V * 3 + 2 * 3 * V * (V + 16)
        roofline = {'bound': 'mfma', 'kernel': 'fm_k_edge_message (split precision)', 'unit': 'TFLOP/s', 'peak': BF16_PEAK_TFLOPS,
                    'achieved': 2 * bf16_mac * E / (us * 1e-6) / 1e12, 'frac': 2 * bf16_mac * E / (us * 1e-6) / 1e12 / BF16_PEAK_TFLOPS,
                    'traffic': None, 'avg_launch_us': us, 'executed_bf16_flop_per_launch': 2 * bf16_mac * E, 'executed_f32_flop_per_launch': 2 * f32_mac * E,
                    'f32_equivalent_tflops': conv_message_flops_per_edge(V) * E / (us * 1e-6) / 1e12,
                    'note': 'OPT-IN split precision, not the headline: achieved = bf16 MFMA FLOPs actually issued (3 products per term) against the dense bf16 peak; the '
                            'kernel is bound by the L1/L2 weight stream, the f32 vector-path GEMMs and VALU, not by the bf16 pipe. f32_equivalent_tflops = the reference '
                            'FLOP count of the op / launch time (exceeds the f32 peak because the work is not done in f32).'}
    elif 'edge_message' in kern:
        us = kern['edge_message']['avg_us']
        flops = conv_message_flops_per_edge(cfg.n_vec_channels) * E
        ex_flops = 2 * ex['edge_message_per_edge'] * E
        ach = flops / (us * 1e-6) / 1e12
        traffic = pmc['hbm_bytes_per_launch'] if pmc else None
        roofline = {'bound': 'mfma', 'kernel': 'fm_k_edge_message', 'achieved': ach, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': ach / FP32_PEAK_TFLOPS, 'traffic': traffic,
                    'traffic_source': (f"committed profile {pmc['source']} (library of commit {pmc['commit']}): (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, "
                                       f"rocprofv3 PMC with the gfx950 FETCH correction; not measured by this run") if pmc else None,
                    'algorithmic_bytes_per_launch': E * (512 + 8) + N * 4 * (256 + 3 * cfg.n_vec_channels) * 2,
                    'avg_launch_us': us,
                    'hbm_gb_per_s': (traffic / (us * 1e-6) / 1e9) if traffic else None,
                    'hbm_frac_of_8tb_per_s': (traffic / (us * 1e-6) / 8e12) if traffic else None,
                    'algorithmic_flop_per_launch': flops,
                    'executed_flop_per_launch': ex_flops,
                    'executed_tflops': ex_flops / (us * 1e-6) / 1e12,
                    'executed_frac': ex_flops / (us * 1e-6) / 1e12 / FP32_PEAK_TFLOPS,
                    'mfma_busy_frac': pmc.get('mfma_busy_frac') if pmc else None,
                    'mfma_busy_source': (f"SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs), {pmc['source_sq']} (library of commit {pmc['commit']})"
                                         if pmc and pmc.get('mfma_busy_frac') else None),
                    'note': 'frac = ALGORITHMIC FLOPs (2*312,251 MAC per directed edge, the reference-executed count, x E edges per launch) / launch time / peak; '
                            'executed_frac = the MFMA FLOPs the kernel really issues (padded GEMM shapes after hoisting the per-source terms, '
                            f"{ex['edge_message_per_edge']} MAC/edge) / launch time / peak -- the matrix-pipe occupancy by construction; "
                            'peak = f32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32) = f32 vector peak'}
    n_list = n_atoms.tolist()
    evals_per_s = mols_per_s / world * T / B                      # network evaluations of this rank's batch per second
    alg_tf = sum(network_flops(int(k)) for k in n_list) * evals_per_s / 1e12
    exe_tf = 2 * (ex['per_edge'] * E + ex['per_node'] * N) * evals_per_s / 1e12
    out = {
        'metric': 'molecules/sec at 250 timesteps (GEOM-drugs-sized graphs)' + ('' if args.precision == 'f32' else ' [opt-in split-precision mode]'), 'value': mols_per_s, 'unit': 'molecules/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'f32' else 'bf16x3 split precision (opt-in; f32 operands as hi+lo bf16, 3 products per term, f32 accumulate)', 'data': 'synthetic',
        'config': {'workload': f'{args.preset} GEOM-drugs model, {B} molecules/GPU x ' + (f'{n} atoms' if args.size_dist is None else f'sizes ~ {args.size_dist} histogram (mean {float(n_atoms.double().mean()):.1f}, max {int(n_atoms.max())})') + f', n_timesteps={T} '
                               f'(BASELINE.json configs[2]; configs[3] at 8 GPUs)',
                   'global_molecules': B * world, 'nodes_per_gpu': N, 'directed_edges_per_gpu': E, 'parallelism': f'molecule-shard x{world}',
                   'step': 'one integration step = 1 network evaluation + Euler/CTMC update of the whole batch',
                   'value_formula': 'global_molecules / (n_timesteps * ms_per_step/1000)', 'weights': 'synthetic by name (seed 0)',
                   'finite': finite},
        'network_eval_ms': ms_per_step, 'per_rank_ms_per_step': per_rank_ms, 'final_gather_ms': gather_ms,
        'launches_per_step': launches_per_step,
        'whole_path': None if args.precision != 'f32' else {'algorithmic_tflops_per_gpu': alg_tf, 'executed_tflops_per_gpu': exe_tf,
                       'executed_frac': exe_tf / FP32_PEAK_TFLOPS,
                       'algorithmic_over_executed': alg_tf / exe_tf,
                       'note': 'algorithmic = the reference-executed FLOP count of a network evaluation (BASELINE.md section 2) per second; executed = MFMA FLOPs '
                               'the kernels issue (per-source terms hoisted to per-node GEMMs, few-input embeddings tabulated); only executed_frac is a '
                               'fraction of the f32 peak -- the algorithmic rate may exceed the peak because fewer FLOPs are executed'},
        'kernels': kern,
        'kernels_note': 'per-kernel averages come from a separate HIP-event-instrumented pass of 2 steps after the timed region',
    }
    if roofline:
        out['roofline'] = roofline
    if rank == 0 and world == 1 and not args.no_api_e2e:
        out['api_end_to_end'] = api_end_to_end(args, B, n, T, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, sd, n, args.cpu_mols, args.cpu_steps, T)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
