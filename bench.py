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

Secondary workloads (kept under profiles/, never the headline): ``--workload c2`` = BASELINE configs[1] (QM9 model, 256 molecules of
18 atoms, n_timesteps = 100) and ``--workload c5`` = configs[4] (geom_full_kekulized model, 128 molecules with sizes
randint(5, 61, seed 0), n_timesteps = 500, trajectory sink on).

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
EMPTY_KERNEL_US = 3.5                            # duration of an empty kernel (rocprofv3 --kernel-trace): part of the calibration pair, not of the pair's overhead.  Only the
                                                 # informational `avg_us` of the per-kernel table uses it; every roofline fraction is quoted on RAW event-pair times


def conv_message_flops_per_edge(V=32, S=256, F=128, R=32, ncp=4):
    """Algorithmic FLOPs of one GVPConv edge message per directed edge, counting 2*MAC of every
    Linear/einsum exactly as the reference executes them (312,251 MAC for flowmol3)."""
    def gvp(vin, h, vout, sin):
        return vin * h * 3 + vin * 2 * ncp * 3 + (h + ncp) * vout * 3 + (sin + h + ncp) * S + S * vout
    mac = gvp(V + 1, V + 1, V, S + R + F) + 2 * gvp(V, V, V, S)
    return 2 * mac


def reference_macs(cfg):
    """(MAC per directed edge, MAC per node) of ONE network evaluation exactly as the reference executes its Linear / einsum ops, from the
    model dimensions (flowmol3 2.437 M / 3.25 M, geom_full_kekulized 1.795 M / 2.195 M -- asserted in
    tests/test_host_logic.py).  Elementwise work is not counted."""
    V, S, F, R, ncp = cfg.n_vec_channels, cfg.n_hidden_scalars, cfg.n_hidden_edge_feats, cfg.rbf_dim, cfg.n_cp_feats
    na, nc, ne = cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types
    m = 1 if cfg.has_mask else 0

    def gvp(vin, vout, sin):
        h = max(vin, vout)
        return vin * h * 3 + vin * 2 * ncp * 3 + (h + ncp) * vout * 3 + (sin + h + ncp) * S + S * vout
    n_upd = sum(1 for u in cfg.update_schedule() if u >= 0)
    msg = gvp(V + 1, V, S + R + F) + 2 * gvp(V, V, S)
    e_in = cfg.e_token_dim or (ne + m)
    per_edge = cfg.n_convs * msg + n_upd * ((2 * S + F + (R if cfg.update_edge_w_distance else 0)) * F + F * F) + e_in * F + F * F + (F * F + F * ne) / 2
    n_in = (cfg.a_token_dim or (na + m)) + (cfg.c_token_dim or (nc + m)) + cfg.time_embedding_dim
    per_node = cfg.n_convs * 3 * gvp(V, V, S) + n_upd * (2 * gvp(V, V, S) + gvp(V, 1, S)) + n_in * S + S * S + S * S + S * (na + nc)
    if cfg.self_conditioning:
        per_edge += ((F + ne + R) * F + F * F) / 2
        per_node += (S + na + nc + R) * S + S * S
    return per_edge, per_node          # n_recycles > 1 (no shipped YAML) would multiply the conv / update terms; not a bench workload


def network_flops(n, cfg):
    """Reference FLOPs of one network evaluation of one n-atom molecule."""
    pe, pn = reference_macs(cfg)
    return 2 * pe * n * (n - 1) + 2 * pn * n


def executed_macs(cfg, U=None, n_cus=256):
    """MACs the kernels actually issue on the matrix pipe per directed edge / per node and network evaluation (padded GEMM
    shapes of flowmol_amd/csrc, after the algebraic hoists of DESIGN.md §3), next to the reference-executed counts the
    algorithmic figures use.  Returned per kernel so every roofline fraction can be quoted both ways."""
    V, S, F, R = cfg.n_vec_channels, cfg.n_hidden_scalars, cfg.n_hidden_edge_feats, cfg.rbf_dim
    p8 = lambda k: (k + 7) // 8 * 8
    p16 = lambda k: (k + 15) // 16 * 16
    def gvp(first, vout):            # one GVP on one row (fm_gvp_core): [Wh|Wcp] (hoisted for the first edge GVP), Wu, Ws, gates
        vop = max(16, vout)
        # scalar GEMM: K = [rbf | ef | sh (padded to V + 8)] for the first GVP; [s | sh] = S + V + 4 for the others (the second MFMA pass of their last
        # k-superstep -- four zero k-slots -- is skipped)
        return (0 if first else 3 * V * (V + 16)) + 3 * (V + 8) * vop + ((R + F + V + 8) if first else (S + V + 4)) * S + S * vop
    msg = gvp(True, V) + 2 * gvp(False, V)
    sched = cfg.update_schedule()
    n_upd = sum(1 for u in sched if u >= 0)
    # pair-slab hoist (fm_config.pair_slab, ABI 6): in self-conditioned models, for the leading convolutions that run before any molecule update
    # (at most two), the [rbf | ef] slab of GVP0's scalar linear is computed once per unordered pair inside the self-conditioning edge kernel and
    # leaves the per-edge kernel -- for batches with at least 16 n_cus 32-row pair tiles, the engine's own gate (U = the workload's unordered pairs;
    # U = None: the model's eligibility only)
    n_pq = 0
    if cfg.self_conditioning and not getattr(cfg, 'use_dst_feats', False):
        for i in range(min(2, cfg.n_convs)):
            if any(u >= 0 for u in sched[:i]):
                break
            n_pq = i + 1
    if U is not None and (U + 31) // 32 < 16 * n_cus:
        n_pq = 0                   # the engine's own gate (fm_engine.cpp: pq_convs): the hoist is on for batches with >= 16 n_cus 32-row pair tiles
    slab = (R + F) * S
    eupd = (F + R) * F + F * F
    sc_e = (p8(cfg.n_bond_types + R) * F + F * F) / 2 if cfg.self_conditioning else 0          # per unordered pair
    head_e = (F * F + F * 16) / 2
    per_edge = cfg.n_convs * msg - n_pq * slab + n_pq * slab / 2 + n_upd * eupd + sc_e + head_e
    node_upd = 3 * gvp(False, V)
    pos = 2 * gvp(False, V) + gvp(False, 1)
    proj = S * S + 3 * V * (V + 16)
    sc_n = (p8(S + cfg.n_atom_types + cfg.n_charges + R) * S + S * S) if cfg.self_conditioning else 0
    head_n = S * S + S * p16(cfg.n_atom_types + cfg.n_charges)
    per_node = cfg.n_convs * (node_upd + proj) + n_upd * (pos + S * S) + sc_n + head_n
    return {'edge_message_per_edge': msg, 'edge_message_pq_per_edge': msg - slab, 'pair_slab_convs': n_pq,
            'pair_slab_per_pair': n_pq * slab, 'edge_update_per_edge': eupd, 'per_edge': per_edge, 'per_node': per_node}


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


def api_end_to_end(args, sizes, T, dev, traj=False):
    """Secondary figure: ONE full `model.sample(B x n atoms, n_timesteps=T)` through the drop-in API -- bind, prior, all T-1 steps
    with torch-generated noise, device->host copy and the per-molecule SampledMolecule packaging -- as wall time."""
    import flowmol_amd as flowmol
    model = flowmol.FlowMol.from_preset(args.preset, precision=args.precision).to(dev).eval()
    model.sample(sizes[:8], n_timesteps=3)                   # engine creation + first-use costs are not part of the figure
    torch.manual_seed(7)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    mols = model.sample(sizes, n_timesteps=T, xt_traj=traj, ep_traj=traj)
    wall = time.perf_counter() - t0
    timing = dict(getattr(model, 'last_timing', {}))
    model.to('cpu')                                           # releases the second engine's workspace
    return {'wall_s': wall, 'molecules': len(mols), 'molecules_per_s': len(mols) / wall, 'n_timesteps': T,
            'breakdown_s': timing, 'note': 'FlowMol.sample() incl. packaging into SampledMolecule objects (no RDKit in this image)'}


def _cpu_steps(cfg, sd, n_atoms, steps, T, threads, bootstrap=True):
    """Warm-up step + `steps` timed integration steps of the CPU oracle on molecules of the given sizes.  bootstrap=True: the trajectory starts at
    t = 0, so the warm-up step carries the bootstrap evaluation of self-conditioned models (two evaluations); False: it starts one step in with a
    synthetic previous endpoint (uniform probabilities, the prior's positions) -- the same per-step work, one evaluation per step throughout."""
    from oracle import cpu_ref
    torch.set_num_threads(threads)
    batch = cpu_ref.build_batch(n_atoms)
    orc = cpu_ref.OracleVF(cfg, sd)
    torch.manual_seed(1)
    prior = orc.sample_prior(batch)
    t = torch.linspace(0, 1, T)
    alpha_t, alpha_tp = cpu_ref.alpha_tables(t)
    state = {'x_t': prior['x_0'], 'a_t': prior['a_0'], 'c_t': prior['c_0'], 'e_t': prior['e_0']}
    noise = cpu_ref.TorchNoise()
    dst = None
    first = 1
    if not bootstrap and cfg.self_conditioning:
        first = 2
        dst = {'x': prior['x_0'].clone(), 'a': torch.full((batch.N, cfg.n_atom_types), 1.0 / cfg.n_atom_types), 'c': torch.full((batch.N, cfg.n_charges), 1.0 / cfg.n_charges),
               'e': torch.full((int(batch.upper_edge_mask.sum()), cfg.n_bond_types), 1.0 / cfg.n_bond_types)}
    times = []
    with torch.no_grad():
        for s_idx in range(first, first + steps + 1):
            t0 = time.perf_counter()
            new, dst = orc.step(batch, state, t[s_idx], t[s_idx - 1], alpha_t[s_idx - 1], alpha_tp[s_idx - 1], prev=dst,
                                eta=cfg.stochasticity, hc_thresh=cfg.high_confidence_threshold, last_step=False, noise=noise)
            state = {k: new[k] for k in ('x_t', 'a_t', 'c_t', 'e_t')}
            times.append(time.perf_counter() - t0)
    return sum(times[1:]) / len(times[1:])


def host_cpu_info():
    """CPU model, sockets, physical cores and logical CPUs of this box (from /proc/cpuinfo; no external tool)."""
    model, phys = None, set()
    pid = cid = None
    try:
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name' and model is None:
                model = v
            elif k == 'physical id':
                pid = v
            elif k == 'core id':
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid)); pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    return {'model': model, 'physical_cores': len(phys) or None, 'sockets': len({p for p, _ in phys}) or None, 'logical_cpus': os.cpu_count()}


def _cost_sample(all_sizes, k):
    """k molecules of the workload whose mean cost n(n-1) represents it (the first k molecules of a ragged workload can be
    20 % off): every (B/k)-th molecule of the list sorted by size, mid-quantiles.  Fixed-size workloads: the first k."""
    B = int(all_sizes.numel())
    k = min(k, B)
    if bool((all_sizes == all_sizes[0]).all()):
        return all_sizes[:k].clone()
    order = torch.argsort(all_sizes, stable=True)
    pick = ((torch.arange(k, dtype=torch.float64) + 0.5) * B / k).long().clamp_(max=B - 1)
    return all_sizes[order[pick]].clone()


def cpu_baseline(cfg, sd, all_sizes, cpu_mols, steps, T, evals, ref_batch=128, ref_steps=1):
    """Time the CPU oracle (the op-for-op restatement of the reference's PyTorch path, oracle/cpu_ref.py; bit-identical to the reference's
    own modules over whole trajectories, profiles/r03a_oracle_long_parity.jsonl) on this box's host cores on bounded samples of the same
    workload: (a) `cpu_mols` molecules (16: the sample of rounds 1-3) and (b) the reference's own batch size, test.py:30
    `--max_batch_size 128` -- the headline `value` is (b), the rate a user of the reference's CPU path would see.  The intra-op thread
    count is chosen by a probe AT THE BATCH SIZE THAT IS TIMED: torch with one thread per logical CPU of a many-core host oversubscribes
    these operators badly, and the best count depends on the operand sizes.  Ragged workloads: a cost-representative quantile sample,
    and the rate is rescaled by the sample's mean n(n-1) over the workload's."""
    ncpu = os.cpu_count() or 1
    host = host_cpu_info()
    phys = host.get('physical_cores') or ncpu
    cost_all = float((all_sizes * (all_sizes - 1)).double().mean())

    def describe(sizes):
        B = int(sizes.numel())
        if bool((sizes == sizes[0]).all()):
            return f'{B} molecules x {int(sizes[0])} atoms'
        return f'a size-quantile sample of {B} molecules of the workload ({int(sizes.min())}-{int(sizes.max())} atoms, mean {float(sizes.double().mean()):.1f})'

    def one(sizes, cands, probe_steps, timed_steps, bootstrap=True, repeat_best=False):
        probe = {}
        for c in cands:          # ascending; stop once more threads are clearly slower (256 threads on the 2 x 64-core host: 190 s per step, r03a)
            probe[c] = _cpu_steps(cfg, sd, sizes, probe_steps, T, c, bootstrap)
            if probe[c] > 1.5 * min(probe.values()):
                break
        best = min(probe, key=probe.get)
        runs = None
        if repeat_best:          # a second, independent timed run of the best thread count: the figure is the mean of the two, both are reported
            runs = [probe[best], _cpu_steps(cfg, sd, sizes, probe_steps, T, best, bootstrap)]
            per_step, n_timed = sum(runs) / 2, 2 * probe_steps
        else:
            per_step = _cpu_steps(cfg, sd, sizes, timed_steps, T, best) if timed_steps else probe[best]
            n_timed = timed_steps or probe_steps
        B = int(sizes.numel())
        ratio = float((sizes * (sizes - 1)).double().mean()) / cost_all            # 1 for fixed-size workloads
        return {'value': B / (evals * per_step) * ratio, 'molecules': B, 'cores': best, 'ms_per_step': per_step * 1e3,
                'timed_runs_ms_per_step': [r * 1e3 for r in runs] if runs else None,
                'sample_cost_over_workload_cost': ratio, 'thread_probe_ms_per_step': {str(k): v * 1e3 for k, v in probe.items()},
                'sample': f'{describe(sizes)}, {n_timed} timed integration steps' + (' (two runs of ' + str(probe_steps) + ', each' if runs else ' (')
                          + f' after 1 warm-up step; {per_step * 1e3:.0f} ms/step) with {best} torch threads '
                          f'(best of {sorted(probe)}, each probed with {probe_steps} step(s) of the same batch)'}
    small = one(_cost_sample(all_sizes, cpu_mols), sorted({c for c in (8, 16, 32, 64) if c <= ncpu} | ({ncpu} if ncpu < 8 else set())), 2, steps)
    out = dict(small)
    big = None
    if ref_batch and int(all_sizes.numel()) >= ref_batch and ref_batch > cpu_mols:
        # the reference's protocol batches 128 molecules (test.py:30).  A step of that batch is ~12 s of host work (r04a: 2 x 64 cores: 16 threads 12.3 s,
        # 32: 11.9 s, 64: 15.9 s, 128: 27.8 s per step), so the probe is three candidates -- 16, 32, 64 threads -- of ONE timed step each after a warm-up step
        # without the bootstrap evaluation (a candidate clearly slower than the best so far ends the probe), and the best one is timed a SECOND time:
        # the figure is the mean of its two timed steps (one step of two candidates was as much probe noise as measurement).
        cands = sorted({c for c in (16, 32, 64) if c <= min(phys, ncpu)} or {min(phys, ncpu)})
        big = one(_cost_sample(all_sizes, ref_batch), cands, ref_steps, 0, bootstrap=False, repeat_best=True)
        out = dict(big)
    out.update({'unit': 'molecules/s', 'kind': 'port', 'host': host,
                'sample': out['sample'] + f"; host: {host['model']}, {host['physical_cores']} physical cores / {ncpu} logical CPUs; extrapolated linearly to {evals} network evaluations per sample"
                          + ('' if out['sample_cost_over_workload_cost'] == 1 else '; rate rescaled by the sample\'s mean n(n-1) over the workload\'s'),
                'at_16_molecules': small if big is not None else None,
                'batch_note': ('value = the reference protocol\'s batch of 128 molecules (test.py:30 --max_batch_size); at_16_molecules = the sample of rounds 1-3'
                               if big is not None else f'{small["molecules"]}-molecule sample')})
    return out


WORKLOADS = {'c3': dict(preset='flowmol3', mols=1024, n=47, T=250, traj=False, label='BASELINE.json configs[2]; configs[3] at 8 GPUs'),
             'c2': dict(preset='qm9', mols=256, n=18, T=100, traj=False, label='BASELINE.json configs[1]'),
             'c5': dict(preset='geom_ctmc', mols=128, n=None, T=500, traj=True, label='BASELINE.json configs[4], trajectory sink on (--xt_traj / --ep_traj)')}
KERNEL_NAMES = ('edge_message', 'edge_message_pq', 'edge_update', 'edge_update_head', 'node_update', 'pos_update', 'node_proj', 'node_proj_asd', 'sc_edge', 'sc_node', 'edge_head', 'node_head', 'sc',
                'heads', 'ctmc', 'ctmc_gat', 'dst_proj', 'embed_table', 'gather_ef', 'gather_s', 'remove_com', 'x_step')


def job_sizes(world, B, n, size_dist):
    """The job's molecules: ONE global list (world * B molecules) dealt to the ranks.  Fixed-size workloads give every rank B molecules;
    ragged workloads (--size-dist, c5) are sharded by cost with shard.partition_lpt, exactly as FlowMol.sample_distributed does, so a
    multi-GPU run shows the real load imbalance of the size distribution (max over ranks is what is timed)."""
    from flowmol_amd import shard
    if size_dist is not None:
        from flowmol_amd.model import load_n_atoms_hist
        vals, counts = load_n_atoms_hist(size_dist)
        all_sizes = vals[torch.multinomial(counts.double(), B * world, replacement=True, generator=torch.Generator().manual_seed(1000))]
    elif n is None:          # c5: 128 molecules per GPU, randint(5, 61) with seed 0
        all_sizes = torch.randint(5, 61, (B * world,), generator=torch.Generator().manual_seed(0))
    else:
        all_sizes = torch.full((B * world,), n, dtype=torch.int64)
    ragged = bool((all_sizes != all_sizes[0]).any())
    parts = shard.partition_lpt(all_sizes, world) if (ragged and world > 1) else [torch.arange(r * B, (r + 1) * B) for r in range(world)]
    return all_sizes, parts, ragged


class Leg:
    """One workload bound on this rank's engine: a real trajectory advanced in windows of steps."""

    def __init__(self, eng, cfg, n_atoms, T, traj, rank, dev, philox=False):
        from flowmol_amd.engine import IntegrationRun, StepNoise, make_step_plan
        self.eng, self.cfg, self.dev, self.rank = eng, cfg, dev, rank
        eng.bind(n_atoms)
        self.N, self.U, self.E = eng.N, eng.U, eng.E
        N, U = self.N, self.U
        self.plan = make_step_plan(T, cfg.stochasticity, cfg.high_confidence_threshold, cfg.cat_temperature, philox_seed=11 if philox else None)
        self.n_plan = len(self.plan.scalars)
        gen = torch.Generator(device=dev)
        gen.manual_seed(2 + rank)
        self.traj = None
        if traj:          # c5: the per-step frames go to the trajectory sink during the timed steps (compact format: fp32 x + int32 tokens)
            i32 = dict(dtype=torch.int32, device=dev)
            n_plan = self.n_plan
            self.traj = {'x': torch.empty(n_plan, N, 3, device=dev), 'a': torch.empty(n_plan, N, **i32), 'c': torch.empty(n_plan, N, **i32), 'e': torch.empty(n_plan, U, **i32),
                         'x1': torch.empty(n_plan, N, 3, device=dev), 'a1': torch.empty(n_plan, N, **i32), 'c1': torch.empty(n_plan, N, **i32), 'e1': torch.empty(n_plan, U, **i32)}
        self.state = self.fresh_state()
        # philox: the CTMC noise is drawn inside the kernel (FlowMol.sample(rng='philox')): no torch RNG launches between the steps
        self.run = IntegrationRun(eng, self.state, self.plan, None if philox else
                                  (lambda i, last: StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, last, dev, generator=gen)), traj=self.traj)
        self.pos = 0

    def fresh_state(self):
        g0 = torch.Generator(device=self.dev)
        g0.manual_seed(1 + self.rank)
        x0 = torch.randn(self.N, 3, device=self.dev, generator=g0)
        self.eng.remove_com(x0)
        return self.eng.prior_state(x0)

    def advance(self, k):
        """k consecutive steps of the trajectory; a new trajectory starts from the prior when one ends."""
        while k > 0:
            if self.pos >= self.n_plan:
                self.state = self.fresh_state()
                self.run.reset(self.state)
                self.pos = 0
            m = min(k, self.n_plan - self.pos)
            self.run.run(self.pos, self.pos + m, chunk=16)
            self.pos += m
            k -= m

    def kernel_times(self, steps=2):
        """Per-kernel averages from a separate HIP-event-instrumented pass (events on the launch stream).  Every profiled step also times an
        EMPTY kernel ('event_overhead'): pair time of the empty kernel - its own 3.5 us = what an event pair adds to a launch.  avg_us = raw pair
        time - that overhead, which makes the figure comparable with rocprofv3's kernel durations also for the sub-millisecond kernels of small
        batches (397 vs 351 us on C2 in round 3; r04g: 332 vs 340)."""
        eng = self.eng
        # one untimed profiled step first: it fills the library's pool of timing events.  A pass that creates two events per launch makes the HOST
        # the bottleneck of a small workload; the GPU then idles between launches and its sub-millisecond kernels measure ~12 % long (r04k: 371 vs
        # 329 us for C2's edge kernel in a cold vs a warm pass; rocprofv3: 338)
        eng.profile(True)
        self.advance(1)
        torch.cuda.synchronize(self.dev)
        eng.profile(True)                # re-enabling returns the events to the pool and clears the accumulators
        self.advance(steps)
        torch.cuda.synchronize(self.dev)
        ms, cnt = eng.profile_get('event_overhead')
        # pair time of the empty kernel minus the empty kernel's own duration (3.5 us by rocprofv3: `fm_k_noop` in profiles/r04g_kernel_stats.txt)
        ovh = max(ms * 1e3 / cnt - EMPTY_KERNEL_US, 0.0) if cnt else 0.0
        kern = {}
        for k in KERNEL_NAMES:
            ms, cnt = eng.profile_get(k)
            if cnt:
                raw = ms * 1e3 / cnt
                kern[k] = {'avg_us': max(raw - ovh, 0.0), 'raw_event_pair_us': raw, 'launches_per_step': cnt / steps}
        eng.profile(False)
        return kern, ovh


def message_roofline(cfg, E, N, us, pmc=None, lib_digest=None, pq=None, us_corrected=None):
    """Roofline object of the dominant kernel -- the full instance of fm_k_edge_message -- for one launch of E edges taking `us` microseconds
    (the RAW HIP-event pair around the launch: every fraction is quoted on it; us_corrected = the same minus the event-pair overhead, beside it).
    pq = (launches per step, raw avg us) of its pair-slab (PQ) instance, reported beside it."""
    ex = executed_macs(cfg)
    flops = conv_message_flops_per_edge(cfg.n_vec_channels) * E
    ex_flops = 2 * ex['edge_message_per_edge'] * E
    ach = flops / (us * 1e-6) / 1e12
    stale = bool(pmc) and pmc.get('library_digest') != lib_digest
    traffic = pmc['hbm_bytes_per_launch'] if (pmc and not stale) else None
    busy = pmc.get('mfma_busy_frac') if (pmc and not stale) else None
    out = {'bound': 'mfma', 'kernel': 'fm_k_edge_message (full instance: the convolutions after the first molecule update)', 'achieved': ach, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
           'frac': ach / FP32_PEAK_TFLOPS, 'traffic': traffic,
           'traffic_source': (f"committed profile {pmc['source']} (library digest {pmc.get('library_digest')} = this run's): (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, "
                              f"rocprofv3 PMC with the gfx950 FETCH correction; not measured by this run") if traffic else
                             (f"committed counters were measured on library digest {pmc.get('library_digest')}, this run is {lib_digest}: not quoted" if stale else None),
           'algorithmic_bytes_per_launch': E * (512 + 8) + N * 4 * (256 + 3 * cfg.n_vec_channels) * 2,
           'avg_launch_us': us, 'avg_launch_us_minus_event_overhead': us_corrected,
           'hbm_gb_per_s': (traffic / (us * 1e-6) / 1e9) if traffic else None,
           'hbm_frac_of_8tb_per_s': (traffic / (us * 1e-6) / 8e12) if traffic else None,
           'algorithmic_flop_per_launch': flops,
           'executed_flop_per_launch': ex_flops,
           'executed_tflops': ex_flops / (us * 1e-6) / 1e12,
           'executed_frac': ex_flops / (us * 1e-6) / 1e12 / FP32_PEAK_TFLOPS,
           'mfma_busy_frac': busy,
           'mfma_busy_source': (f"SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs), {pmc['source_sq']} (library digest {pmc.get('library_digest')})" if busy else None),
           'note': f'frac = ALGORITHMIC FLOPs (2*{conv_message_flops_per_edge(cfg.n_vec_channels) // 2:,} MAC per directed edge, the reference-executed count, '
                   'x E edges per launch) / launch time / peak; '
                   'executed_frac = the MFMA FLOPs the kernel really issues (padded GEMM shapes after hoisting the per-source terms, '
                   f"{ex['edge_message_per_edge']} MAC/edge) / launch time / peak -- the matrix-pipe occupancy by construction; "
                   'avg_launch_us = the RAW HIP-event pair around the launch on the launch stream (every fraction here is quoted on it: conservative by the ~3 us a pair adds); '
                   'avg_launch_us_minus_event_overhead = the same minus the pair time of an empty kernel measured in the same pass (what rocprofv3 reports for the kernel); '
                   'peak = f32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2_f32) = f32 vector peak'}
    if pq:
        n_pq, us_pq = pq
        ex_pq = 2 * ex['edge_message_pq_per_edge'] * E
        out['pair_slab_instance'] = {
            'kernel': 'fm_k_edge_message<..., PQ = 1>', 'launches_per_step': n_pq, 'avg_launch_us': us_pq,
            'algorithmic_frac': flops / (us_pq * 1e-6) / 1e12 / FP32_PEAK_TFLOPS, 'executed_frac': ex_pq / (us_pq * 1e-6) / 1e12 / FP32_PEAK_TFLOPS,
            'executed_mac_per_edge': ex['edge_message_pq_per_edge'],
            'note': 'the convolutions before the first molecule update: the [rbf | ef] slab of their first scalar GEMM (40,960 MAC/edge of the reference count) is computed once per '
                    'unordered pair in the self-conditioning edge kernel and gathered here, so the ALGORITHMIC fraction (reference FLOPs / time / peak) exceeds 1 -- that work is not '
                    'executed in this kernel; executed_frac is the matrix-pipe share'}
    return out


def load_pmc(workload, N, E, ok):
    try:
        pj = json.loads((ROOT / 'profiles' / 'current_pmc.json').read_text()).get(workload)
        if pj and pj['nodes_per_gpu'] == N and pj['directed_edges_per_gpu'] == E and ok:
            return pj
    except Exception:
        pass
    return None


def secondary_legs(engines, dev, lib_digest, steps):
    """The rest of BASELINE.json's metric in the same driver-run line: GEOM-drugs size distribution, configs[1] (C2),
    configs[4] (C5, trajectory sink on) and the per-step latency at 1 / 8 / 32 / 128 molecules -- each a bounded window of a real trajectory
    on one GPU, with its own ms_per_step, molecules/s and the roofline of its dominant kernel."""
    from flowmol_amd import presets, weights
    from flowmol_amd.engine import Engine

    def engine(preset, tuning=None):
        key = preset if not tuning else (preset, tuple(sorted(tuning.items())))
        if key not in engines:
            cfg = presets.PRESETS[preset]()
            engines[key] = (cfg, Engine(cfg, weights.synth_state_dict(cfg, 0), device=dev, precision='f32', tuning=tuning))
        return engines[key]

    def leg(name, preset, sizes, T, traj, label, k_steps, warm=3, philox=False, tuning=None):
        cfg, eng = engine(preset, tuning)
        L = Leg(eng, cfg, sizes, T, traj, 0, dev, philox=philox)
        L.advance(warm)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        L.advance(k_steps)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3 / k_steps
        kern, ovh = L.kernel_times(2)
        evals = T if cfg.self_conditioning else T - 1
        B = int(sizes.numel())
        o = {'workload': label, 'molecules': B, 'nodes': L.N, 'directed_edges': L.E, 'n_timesteps': T, 'steps': k_steps, 'warmup': warm, 'ms_per_step': ms,
             'value': B / (evals * ms / 1e3), 'unit': f'molecules/s at {T} timesteps', 'network_evaluations_per_sample': evals, 'trajectory_sink': bool(traj),
             'finite': bool(torch.isfinite(L.state['x_t']).all().item()), 'event_pair_overhead_us': ovh,
             'kernels_us': {k: round(v['avg_us'], 1) for k, v in kern.items()}, 'launches_per_step': sum(v['launches_per_step'] for v in kern.values())}
        if 'edge_message' in kern:
            pq = (kern['edge_message_pq']['launches_per_step'], kern['edge_message_pq']['raw_event_pair_us']) if 'edge_message_pq' in kern else None
            r = message_roofline(cfg, L.E, L.N, kern['edge_message']['raw_event_pair_us'], load_pmc(name, L.N, L.E, True), lib_digest, pq, kern['edge_message']['avg_us'])
            o['roofline'] = {k: r[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_us', 'executed_frac', 'mfma_busy_frac')}
            o['dominant_kernel_share_of_step'] = sum(kern[k]['avg_us'] * kern[k]['launches_per_step'] for k in ('edge_message', 'edge_message_pq') if k in kern) / (ms * 1e3)
        del L
        return o
    out = {}
    sd_sizes, _, _ = job_sizes(1, 1024, None, 'geom_full_kekulized')
    out['geom_size_dist'] = leg('geom_size_dist', 'flowmol3', sd_sizes, 250, False,
                                'flowmol3 model, 1024 molecules with sizes ~ the shipped GEOM-drugs histogram '
                                f'(seed 1000: mean {float(sd_sizes.double().mean()):.1f}, max {int(sd_sizes.max())} atoms), n_timesteps=250 '
                                '(the metric\'s "GEOM-drugs size dist"; reference flowmol.py:461-471)', steps)
    out['c2'] = leg('c2', 'qm9', torch.full((256,), 18, dtype=torch.int64), 100, False, 'qm9 model, 256 molecules x 18 atoms, n_timesteps=100 (BASELINE.json configs[1])', 4 * steps)
    c5_sizes, _, _ = job_sizes(1, 128, None, None)
    out['c5'] = leg('c5', 'geom_ctmc', c5_sizes, 500, True,
                    'geom_ctmc model, 128 molecules with sizes randint(5, 61, seed 0), n_timesteps=500, trajectory sink on (BASELINE.json configs[4])', 4 * steps)
    # per-step latency in BOTH arithmetic modes: the default (canonical: a molecule's bits do not depend on its batch; small batches run 4-node tiles and
    # 4-row node MLPs whose GEMMs keep the regular tiles' summation order) and FlowMol(canonical=False) / fm_config.canonical = -1, where the pair slab
    # follows the batch size too (round 5's "latency mode": since the 4-row kernels are canonical the two differ by ~1 %)
    for key, tuning, what in (('latency_sweep', None, 'canonical arithmetic (default)'),
                              ('latency_sweep_latency_mode', {'canonical': -1}, 'canonical=False (the pair slab follows the batch size)')):
        sweep = []
        for B in (1, 8, 32, 128):
            o = leg(f'latency_{B}', 'flowmol3', torch.full((B,), 47, dtype=torch.int64), 250, False,
                    f"flowmol3 model, {B} molecule(s) x 47 atoms, {what}: per-step latency of network evaluation + CTMC update, "
                    "in-kernel Philox noise (sample(rng='philox'): no torch RNG launches between the steps)",
                    64, warm=8, philox=True, tuning=tuning)
            sweep.append({k: o[k] for k in ('molecules', 'ms_per_step', 'value', 'steps', 'launches_per_step', 'kernels_us', 'event_pair_overhead_us', 'workload')
                          } | {'roofline': o.get('roofline')})
        out[key] = sweep
    # the opt-in split-precision modes on the headline workload, each a short window (NOT the headline: `value` of the line is f32; these modes are
    # selected by an explicit argument only -- accuracy of each against float64 and on the reference trajectories: DESIGN.md section 3, profiles/r05c_*, r05f_*)
    opt = {}
    c3_sizes = torch.full((1024,), 47, dtype=torch.int64)
    for prec in ('f16x3', 'bf16x3', 'bf16x6'):
        cfg3 = presets.flowmol3()
        eng_p = Engine(cfg3, weights.synth_state_dict(cfg3, 0), device=dev, precision=prec)
        L = Leg(eng_p, cfg3, c3_sizes, 250, False, 0, dev)
        L.advance(3)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        L.advance(steps)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) * 1e3 / steps
        opt[prec] = {'ms_per_step': ms, 'value': 1024 / (250 * ms / 1e3), 'unit': 'molecules/s at 250 timesteps', 'steps': steps, 'warmup': 3,
                     'finite': bool(torch.isfinite(L.state['x_t']).all().item())}
        del L
        eng_p.close()
        del eng_p
        torch.cuda.empty_cache()
    opt['note'] = ('opt-in arithmetic, never the default: f16x3 = hi + lo IEEE-half operands (22 mantissa bits, <= 1.4x the f32 kernels\' error per stage, operands clamped to +-65504); '
                   'bf16x3 = hi + lo bf16 (16 bits, up to 21.6x); bf16x6 = hi + mid + lo bf16 for the edge messages only (24 bits, <= 1.32x)')
    out['opt_in_precisions'] = opt
    out['note'] = ('secondary legs of the same run (rank 0, one GPU): windows of real trajectories after the headline leg; ms_per_step = wall clock over `steps` consecutive '
                   'integration steps between device synchronisations; value = molecules / (network evaluations per sample x ms_per_step)')
    return out


def size_dist_leg(eng, cfg, world, rank, dev, B, T, steps, warm=3):
    """The metric's own wording -- "molecules/sec at 250 timesteps (GEOM-drugs size dist)" -- on N ranks: ONE global list of B x N sizes drawn from the shipped
    GEOM-drugs histogram (reference flowmol.py:461-471), dealt to the ranks by shard.partition_lpt (the only leg whose load balance is not trivially 1.0),
    `steps` integration steps timed between barriers, MAX over ranks.  Returns the fields rank 0 puts at the top level of the line."""
    sizes, parts, _ = job_sizes(world, B, None, 'geom_full_kekulized')
    mine = sizes[parts[rank]]
    L = Leg(eng, cfg, mine, T, False, rank, dev)
    L.advance(warm)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    L.advance(steps)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ms = float(el.item()) * 1e3 / steps
    evals = T if cfg.self_conditioning else T - 1
    cost = (sizes * (sizes - 1)).double()
    sc = torch.tensor([float(cost[p_].sum()) for p_ in parts])
    finite = bool(torch.isfinite(L.state['x_t']).all().item())
    del L
    return {'value_geom_size_dist': B * world / (evals * ms / 1e3), 'ms_per_step_geom_size_dist': ms,
            'workload_size_dist': f'flowmol3 model, {B * world} molecules with sizes ~ the shipped GEOM-drugs histogram '
                                  f'(seed 1000: mean {float(sizes.double().mean()):.1f}, max {int(sizes.max())} atoms), '
                                  f'n_timesteps={T}, dealt to {world} rank(s) by shard.partition_lpt; {steps} timed steps after {warm} warm-up steps, max over ranks',
            'size_dist_shard_cost_max_over_mean': float(sc.max() / sc.mean()), 'size_dist_finite': finite}


PARITY_MOLS_PER_RANK, PARITY_T, PARITY_SEED = 8, 12, 1234
DTYPE_LABEL = {       # `dtype` of the line: the arithmetic the path computes in
    'f32': 'f32',
    'bf16x3': 'bf16x3 split precision (opt-in; f32 operands as hi+lo bf16, 3 products per term, f32 accumulate)',
    'f16x3': 'f16x3 split precision (opt-in; f32 operands as hi+lo IEEE half = 22 mantissa bits, 3 products per term on v_mfma_f32_16x16x32_f16, f32 accumulate; '
             '|activations| clamped to 65504)',
    'bf16x6': 'bf16x6 three-term split precision of the edge-message GEMMs (opt-in; f32 operands as hi+mid+lo bf16, 6 products per term, f32 accumulate; '
              'node kernels and EdgeUpdate f32)',
}


def parity_job_sizes(world):
    """The small GEOM-distributed job of the multi-GPU self-check: 8 molecules per rank, sizes ~ the shipped GEOM-drugs histogram (fixed seed)."""
    from flowmol_amd.model import load_n_atoms_hist
    vals, counts = load_n_atoms_hist('geom_full_kekulized')
    return vals[torch.multinomial(counts.double(), PARITY_MOLS_PER_RANK * world, replacement=True, generator=torch.Generator().manual_seed(77))]


def gather_slot_bytes(n_atoms, parts):
    """Bytes one rank contributes to the one all-gather (shard.gather_results: the largest packed payload, 14 B/atom + 1 B/pair, padded to 16)."""
    pay = []
    for p_ in parts:
        nr = n_atoms[p_]
        pay.append(int(nr.sum()) * 14 + int((nr * (nr - 1) // 2).sum()))
    return (max(pay) + 15) // 16 * 16, pay


def multi_gpu_parity(cfg, sd, eng, world, rank, dev, backend, n_atoms=None, T=None):
    """The first thing a multi-GPU run does: prove the sharded path before timing it.  A small GEOM-distributed job
    (8 x world molecules, n_timesteps = 12) is sampled by the `world` ranks with FlowMol.sample_distributed in BOTH noise modes and by rank 0
    alone with the same seed:
      * noise='replicated' -- every rank draws the full batch's noise and keeps its rows -- must reproduce the single-process sample() token
        for token, coordinates within 1e-4 relative (the north-star tolerance; measured: f32 summation order);
      * noise='philox' -- the in-kernel per-molecule streams a throughput run uses -- must reproduce the single-process Philox sample.
    Every rank must hold the same gathered batch (digest compared across ranks).  The block also records what the run physically was: distinct
    PCI devices of the ranks, collective backend and RCCL version, bytes of the one all-gather.  Any token difference ends the run with a
    non-zero exit on EVERY rank before a single step is timed."""
    import hashlib
    import flowmol_amd as flowmol
    from flowmol_amd import shard
    model = flowmol.FlowMol(cfg, {'vector_field.' + k: v for k, v in sd.items()})
    model.device, model._engine = dev, eng                       # the rank's one engine (no second context / weight copy)
    n_atoms = parity_job_sizes(world) if n_atoms is None else n_atoms      # (the arguments exist for the CPU test of this block: tiny molecules on the emulated kernels)
    T = T or PARITY_T
    parts = shard.partition_lpt(n_atoms, world)
    slot, payloads = gather_slot_bytes(n_atoms, parts)
    res, digests = {}, {}
    for mode in ('replicated', 'philox'):
        torch.manual_seed(PARITY_SEED)
        full, _ = model.sample_distributed(n_atoms, n_timesteps=T, return_tensors=True, noise=mode)
        res[mode] = full
        digests[mode] = hashlib.sha256(b''.join(full[k].contiguous().numpy().tobytes() for k in 'xace')).hexdigest()[:16]
    try:
        p = torch.cuda.get_device_properties(dev)
        pci, uuid = f'{getattr(p, "pci_domain_id", 0):04x}:{getattr(p, "pci_bus_id", 0):02x}:{getattr(p, "pci_device_id", 0):02x}', str(getattr(p, 'uuid', ''))
    except Exception:          # not a GPU (the CPU test of this block)
        pci, uuid = str(dev), ''
    mine = {'rank': rank, 'pci': pci, 'uuid': uuid, 'digests': digests, 'molecules': int(len(parts[rank])), 'payload_bytes': payloads[rank]}
    infos = [None] * world
    dist.all_gather_object(infos, mine)
    verdict = [None]
    if rank == 0:
        single = {}
        torch.manual_seed(PARITY_SEED)
        single['replicated'], _ = model.sample(n_atoms, n_timesteps=T, return_tensors=True)
        torch.manual_seed(PARITY_SEED)
        single['philox'], _ = model.sample(n_atoms, n_timesteps=T, return_tensors=True, rng='philox')
        out = {'world_size': world, 'backend': backend, 'molecules': int(n_atoms.numel()), 'n_timesteps': T,
               'sizes_min_mean_max': [int(n_atoms.min()), float(n_atoms.double().mean()), int(n_atoms.max())],
               'molecules_per_rank': [i['molecules'] for i in infos]}
        tok = {}
        for mode in ('replicated', 'philox'):
            tok[mode] = int(sum((res[mode][k] != single[mode][k]).sum() for k in 'ace'))
            out[f'{mode}_x_rel'] = float((res[mode]['x'] - single[mode]['x']).abs().max() / single[mode]['x'].abs().max())
            out[f'{mode}_x_bit_identical'] = bool(torch.equal(res[mode]['x'], single[mode]['x']))
        out['token_diffs'] = tok['replicated']
        out['x_rel'] = out.pop('replicated_x_rel')
        out['x_bit_identical'] = out.pop('replicated_x_bit_identical')
        canonical = eng.tuning.get('canonical', 0) >= 0       # canonical arithmetic (the default): a molecule's bits do not depend on its batch or shard
        out['canonical'] = bool(canonical)
        out['philox_token_diffs'] = tok['philox']
        out['tokens_compared'] = int(sum(single['replicated'][k].numel() for k in 'ace'))
        out['ranks_hold_the_same_batch'] = all(i['digests'] == infos[0]['digests'] for i in infos)
        out['distinct_pci_devices'] = len({(i['pci'], i['uuid']) for i in infos})
        out['pci_devices'] = [i['pci'] for i in infos]
        try:
            out['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            out['rccl_version'] = None
        out['all_gather_bytes'] = slot * world
        out['all_gather_slot_bytes'] = slot
        out['payload_bytes_per_rank'] = [i['payload_bytes'] for i in infos]
        out['ok'] = bool(out['token_diffs'] == 0 and out['philox_token_diffs'] == 0 and out['x_rel'] < 1e-4 and out['philox_x_rel'] < 1e-4 and out['ranks_hold_the_same_batch']
                         and (not canonical or (out['x_bit_identical'] and out['philox_x_bit_identical'])))
        out['note'] = ("sample_distributed(noise='replicated' | 'philox') on all ranks vs the single-process sample() of the same seed on rank 0, before the timed region; "
                       'token_diffs / x_rel = replicated mode (north star: indices bit-exact, coordinates within 1e-4 relative); with canonical arithmetic (the default) the coordinates '
                       'must be BIT-IDENTICAL in both modes (x_bit_identical, philox_x_bit_identical); a run with ok = false exits non-zero without timing anything')
        verdict[0] = out
    dist.broadcast_object_list(verdict, src=0)
    return verdict[0]


def bind_rank_to_gpu_socket(dev_index):
    """Pin this rank's host threads to the CPUs of its GPU's NUMA node (8 ranks enqueue ~22 launches per 66 ms step each; a rank whose
    thread migrates to the other socket pays for every descriptor write).  Best effort: returns what was found / done."""
    info = {'device_index': dev_index, 'numa_node': None, 'cpus_bound': None}
    try:
        p = torch.cuda.get_device_properties(dev_index)
        bdf = f'{getattr(p, "pci_domain_id", 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
        info['pci'] = bdf
        node = int(Path(f'/sys/bus/pci/devices/{bdf}/numa_node').read_text())
        info['numa_node'] = node
        if node >= 0:
            cpus = set()
            for part in Path(f'/sys/devices/system/node/node{node}/cpulist').read_text().strip().split(','):
                a, _, b = part.partition('-')
                cpus.update(range(int(a), int(b or a) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
                info['cpus_bound'] = len(cpus)
    except Exception as e:          # containers without /sys, devices without a NUMA entry: run unpinned
        info['note'] = f'not pinned: {type(e).__name__}'
    return info


def dry_run(args):
    """`bench.py --gpus N --dry-run`: the N-rank plan without touching a GPU -- molecules and cost per rank, payload bytes of the one
    all-gather, the device each rank would take and the launch line."""
    world = args.gpus
    all_sizes, parts, ragged = job_sizes(world, args.mols_per_gpu, args.n_atoms, args.size_dist)
    cost = (all_sizes * (all_sizes - 1)).double()
    plan = []
    for r, p_ in enumerate(parts):
        nr = all_sizes[p_]
        N, U = int(nr.sum()), int((nr * (nr - 1) // 2).sum())
        plan.append({'rank': r, 'device': f'cuda:{r} (LOCAL_RANK)', 'molecules': int(len(p_)), 'nodes': N, 'directed_edges': 2 * U, 'cost': float(cost[p_].sum()),
                     'gather_payload_bytes': N * 14 + U, 'ef_bytes': 2 * U * 512})
    costs = torch.tensor([p_['cost'] for p_ in plan])
    slot = (max(p_['gather_payload_bytes'] for p_ in plan) + 15) // 16 * 16
    out = {'dry_run': True, 'n_gpus': world, 'workload': args.workload, 'global_molecules': int(all_sizes.numel()), 'ragged': ragged,
           'shard_cost_max_over_mean': float(costs.max() / costs.mean()), 'all_gather_slot_bytes': slot, 'all_gather_total_bytes': slot * world,
           'ranks': plan, 'gpus_visible_here': torch.cuda.device_count(),
           'launch': f'python -m torch.distributed.run --nnodes=1 --nproc-per-node {world} --master-addr 127.0.0.1 --master-port P bench.py --gpus {world} --steps {args.steps} --warmup {args.warmup}',
           'backend': 'nccl (RCCL over xGMI); one process per GPU; no collective during integration, ONE all_gather_into_tensor at the end',
           }
    if world > 1:          # what the run will check before it times anything (multi_gpu_parity)
        from flowmol_amd import shard
        pn = parity_job_sizes(world)
        pparts = shard.partition_lpt(pn, world)
        slot_p, pay = gather_slot_bytes(pn, pparts)
        out['multi_gpu_parity_plan'] = {
            'world_size': world, 'molecules': int(pn.numel()), 'n_timesteps': PARITY_T, 'seed': PARITY_SEED, 'sizes': pn.tolist(),
            'molecules_per_rank': [int(len(p_)) for p_ in pparts], 'payload_bytes_per_rank': pay, 'all_gather_slot_bytes': slot_p, 'all_gather_bytes': slot_p * world,
            'checks': "sample_distributed(noise='replicated' and 'philox') on the N ranks == sample() of the same seed on rank 0: token_diffs == 0, x_rel < 1e-4, every rank holds the same "
                      'gathered batch; emitted as `multi_gpu_parity` {token_diffs, x_rel, philox_token_diffs, world_size, distinct_pci_devices, rccl_version, all_gather_bytes, ok} in the JSON '
                      'line; a failing check exits non-zero before the timed region',
            'exercised_by': 'tests/test_gpu_parity.py::test_bench_multi_gpu_line_carries_parity_block (8 gloo ranks on one GPU; nccl with 2 and all devices when the box has them)'}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=('c3', 'c2', 'c5'), default='c3',
                    help="c3 (default, the headline): BASELINE configs[2] (flowmol3, 1024 x 47 atoms, T=250; configs[3] with --gpus 8).  Secondary: c2 = configs[1] "
                         "(QM9 model, 256 x 18 atoms, T=100); c5 = configs[4] (geom_full_kekulized model, 128 molecules of randint(5,61,seed 0) atoms, T=500, trajectory sink on)")
    ap.add_argument('--mols-per-gpu', type=int, default=None)
    ap.add_argument('--n-atoms', type=int, default=None)
    ap.add_argument('--size-dist', default=None, help="draw the molecule sizes from a shipped training-set histogram (e.g. geom_full_kekulized) instead of --n-atoms; "
                                                      "secondary measurement, the headline line uses fixed sizes")
    ap.add_argument('--timesteps', type=int, default=None)
    ap.add_argument('--preset', default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--precision', choices=('f32', 'bf16x3', 'bf16x6', 'f16x3'), default='f32',
                    help="arithmetic of the edge-message GEMMs: 'f32' (default, the reference's arithmetic, the headline) or the OPT-IN split precision "
                         "'bf16x3' (f32 operands as hi+lo bf16, three products on the bf16 matrix cores) / 'bf16x6' (hi+mid+lo, six products, edge messages only) / "
                         "'f16x3' (hi+lo IEEE half, three products) -- separately reported modes")
    ap.add_argument('--no-api-e2e', action='store_true', help='skip the secondary end-to-end FlowMol.sample() timing')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary legs (size distribution, C2, C5, latency sweep) of the default one-GPU run')
    ap.add_argument('--secondary-steps', type=int, default=10, help='timed steps of the size-distribution leg (C2 / C5: 4x, latency sweep: 64)')
    ap.add_argument('--dry-run', action='store_true', help='print the N-rank plan (molecules, cost, payload bytes, devices per rank) as one JSON line without touching a GPU')
    ap.add_argument('--cpu-mols', type=int, default=16)
    ap.add_argument('--cpu-steps', type=int, default=8)
    ap.add_argument('--cpu-ref-batch', type=int, default=128, help="CPU baseline at the reference protocol's batch size (test.py:30); 0 = only the --cpu-mols sample")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    args.preset = args.preset or wl['preset']
    args.mols_per_gpu = args.mols_per_gpu or wl['mols']
    args.n_atoms = args.n_atoms or wl['n']
    args.timesteps = args.timesteps or wl['T']
    if args.dry_run:
        return dry_run(args)

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
    affinity = bind_rank_to_gpu_socket(dev_index) if world > 1 else {'device_index': dev_index, 'note': 'single rank: not pinned'}
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
    from flowmol_amd.engine import Engine

    cfg = presets.PRESETS[args.preset]()
    sd = weights.synth_state_dict(cfg, 0)
    eng = Engine(cfg, sd, device=dev, precision=args.precision)
    B, n, T = args.mols_per_gpu, args.n_atoms, args.timesteps
    all_sizes, parts, ragged = job_sizes(world, B, n, args.size_dist)
    n_atoms = all_sizes[parts[rank]]
    cost = (all_sizes * (all_sizes - 1)).double()
    shard_cost = torch.tensor([float(cost[p_].sum()) for p_ in parts])
    parity = None
    if world > 1:
        parity = multi_gpu_parity(cfg, sd, eng, world, rank, dev, dist.get_backend())
        if not parity['ok']:
            if rank == 0:
                print('bench.py: MULTI-GPU PARITY FAILED -- nothing was timed: ' + json.dumps(parity), file=sys.stderr, flush=True)
            dist.destroy_process_group()
            raise SystemExit(3)
    leg = Leg(eng, cfg, n_atoms, T, wl['traj'], rank, dev)
    N, U, E = leg.N, leg.U, leg.E

    def gather_all():
        """The single collective of the sampling path: packed results over RCCL/xGMI, every rank gets the whole batch."""
        st = leg.state
        return shard.gather_results({'x': st['x_t'], 'a': st['a_t'], 'c': st['c_t'], 'e': st['e_t']}, all_sizes, parts)

    leg.advance(args.warmup)
    if world > 1:
        gather_all()             # untimed warm-up of the collective (communicator setup, first-use kernel loads)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    # the K timed steps as (up to) four consecutive sub-windows with an event between them on the launch stream: no synchronisation is added, the
    # windows' durations give min / median / max of ms_per_step inside the one timed region (">= 3 timed runs, median")
    n_win = 4 if args.steps >= 4 else 1
    win_steps = [args.steps // n_win + (1 if w < args.steps % n_win else 0) for w in range(n_win)]
    win_ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
    win_ev[0].record()
    for w in range(n_win):
        leg.advance(win_steps[w])
        win_ev[w + 1].record()
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
    rank_info = [{'rank': rank, 'device': torch.cuda.get_device_name(dev_index), **affinity}]
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
        infos = [None] * world
        dist.all_gather_object(infos, rank_info[0])
        rank_info = infos
    ms_per_step = elapsed * 1e3 / args.steps
    win_ms = sorted(win_ev[w].elapsed_time(win_ev[w + 1]) / win_steps[w] for w in range(n_win))
    windows = {'windows': n_win, 'steps_per_window': win_steps, 'min': win_ms[0], 'median': (win_ms[(n_win - 1) // 2] + win_ms[n_win // 2]) / 2, 'max': win_ms[-1],
               'spread_rel': (win_ms[-1] - win_ms[0]) / win_ms[0], 'rank': rank,
               'note': 'consecutive sub-windows of the timed steps on this rank, HIP events on the launch stream (no extra synchronisation); ms_per_step above is the wall clock over all of them'}
    evals = T if cfg.self_conditioning else T - 1      # network evaluations per sample: T-1 steps (+ the bootstrap evaluation of self-conditioned models)
    mols_per_s = B * world / (evals * ms_per_step / 1e3)

    # ---- per-kernel timing (HIP events on the launch stream) for the roofline of the dominant kernel: a separate
    #      event-instrumented pass of 2 more steps AFTER the timed region
    finite = bool(torch.isfinite(leg.state['x_t']).all().item())
    try:            # which build of the kernels produced this line: digest of csrc/ + header + flags (flowmol_amd/build.py), as in profiles/current_pmc.json
        from flowmol_amd import build as fm_build
        lib_digest = fm_build.STAMP.read_text().strip()[:16]
    except Exception:
        lib_digest = None
    kern, ev_overhead = leg.kernel_times(2)
    launches_per_step = sum(v['launches_per_step'] for v in kern.values())
    # counters of the dominant kernel from the committed rocprofv3 PMC passes (same workload AND same library digest only); never measured by this run
    pmc = load_pmc(args.workload, N, E, args.size_dist is None and args.precision == 'f32' and world == 1)
    ex = executed_macs(cfg, U, torch.cuda.get_device_properties(dev).multi_processor_count)
    roofline = None
    if 'edge_message' in kern and args.precision in ('bf16x3', 'bf16x6', 'f16x3'):
        # opt-in mode: the scalar and gate GEMMs issue 3 bf16 products per term on padded K (7 / 10 / 10 k32 blocks); the vector path stays f32
        us = kern['edge_message']['raw_event_pair_us']
        V = cfg.n_vec_channels
        ku0 = (V + 1 + 4 + 7) // 8 * 8
        kb = [(160 + ku0 + 31) // 32, (256 + V + 8 + 31) // 32, (256 + V + 8 + 31) // 32]
        n_prod = 6 if args.precision == 'bf16x6' else 3          # products per term: hi*hi + hi*lo + lo*hi | + hi*mid, mid*hi, mid*mid (three-term split)
        bf16_mac = n_prod * (sum(k * 32 * 256 for k in kb) + 3 * 256 * V)
        f32_mac = 3 * (V + 8) * V * 3 + 2 * 3 * V * (V + 16)
        roofline = {'bound': 'mfma', 'kernel': 'fm_k_edge_message (split precision)', 'unit': 'TFLOP/s', 'peak': BF16_PEAK_TFLOPS,
                    'achieved': 2 * bf16_mac * E / (us * 1e-6) / 1e12, 'frac': 2 * bf16_mac * E / (us * 1e-6) / 1e12 / BF16_PEAK_TFLOPS,
                    'traffic': None, 'avg_launch_us': us, 'executed_bf16_flop_per_launch': 2 * bf16_mac * E, 'executed_f32_flop_per_launch': 2 * f32_mac * E,
                    'f32_equivalent_tflops': conv_message_flops_per_edge(V) * E / (us * 1e-6) / 1e12,
                    'note': f'OPT-IN split precision ({args.precision}), not the headline: achieved = bf16 MFMA FLOPs actually issued ({n_prod} products per term) against the dense bf16 peak; the '
                            'kernel is bound by the L1/L2 weight stream, the f32 vector-path GEMMs and VALU, not by the bf16 pipe. f32_equivalent_tflops = the reference '
                            'FLOP count of the op / launch time (exceeds the f32 peak because the work is not done in f32).'}
    elif 'edge_message' in kern:
        pq = (kern['edge_message_pq']['launches_per_step'], kern['edge_message_pq']['raw_event_pair_us']) if 'edge_message_pq' in kern else None
        roofline = message_roofline(cfg, E, N, kern['edge_message']['raw_event_pair_us'], pmc, lib_digest, pq, kern['edge_message']['avg_us'])
    n_list = n_atoms.tolist()
    evals_per_s = mols_per_s / world * evals / B                      # network evaluations of this rank's batch per second
    alg_tf = sum(network_flops(int(k), cfg) for k in n_list) * evals_per_s / 1e12
    exe_tf = 2 * (ex['per_edge'] * E + ex['per_node'] * N) * evals_per_s / 1e12
    try:
        rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl = None
    out = {
        'metric': (f'molecules/sec at {T} timesteps ' + ('(GEOM-drugs-sized graphs)' if args.workload == 'c3' else f'[secondary workload {args.workload}]'))
                  + ('' if args.precision == 'f32' else ' [opt-in split-precision mode]'), 'value': mols_per_s, 'unit': 'molecules/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE_LABEL[args.precision], 'data': 'synthetic',
        'config': {'workload': f'{args.preset} model, {B} molecules/GPU x '
                               + (f'{n} atoms' if not ragged else (f'sizes ~ {args.size_dist} histogram' if args.size_dist else 'sizes randint(5, 61, seed 0)')
                                  + f' (mean {float(all_sizes.double().mean()):.1f}, max {int(all_sizes.max())}; ONE global list dealt to the ranks by shard.partition_lpt)') + f', n_timesteps={T} '
                               f"({wl['label']})",
                   'shard_cost_max_over_mean': float(shard_cost.max() / shard_cost.mean()), 'molecules_per_rank': [int(len(p_)) for p_ in parts],
                   'global_molecules': B * world, 'nodes_per_gpu': N, 'directed_edges_per_gpu': E, 'parallelism': f'molecule-shard x{world}',
                   'step': 'one integration step = 1 network evaluation + Euler/CTMC update of the whole batch',
                   'value_formula': 'global_molecules / (network_evaluations_per_sample * ms_per_step/1000)', 'network_evaluations_per_sample': evals,
                   'trajectory_sink': bool(wl['traj']), 'weights': 'synthetic by name (seed 0)',
                   'finite': finite, 'library_digest': lib_digest,
                   'process_group': {'size': world, 'backend': (dist.get_backend() if world > 1 else None), 'rccl_version': rccl, 'ranks': rank_info}},
        'network_eval_ms': ms_per_step, 'ms_per_step_windows': windows, 'per_rank_ms_per_step': per_rank_ms, 'final_gather_ms': gather_ms,
        'multi_gpu_parity': parity,
        'launches_per_step': launches_per_step,
        'whole_path': None if args.precision != 'f32' else {'algorithmic_tflops_per_gpu': alg_tf, 'executed_tflops_per_gpu': exe_tf,
                       'executed_frac': exe_tf / FP32_PEAK_TFLOPS,
                       'algorithmic_over_executed': alg_tf / exe_tf,
                       'note': 'algorithmic = the reference-executed FLOP count of a network evaluation (BASELINE.md section 2) per second; executed = MFMA FLOPs '
                               'the kernels issue (per-source terms hoisted to per-node GEMMs, few-input embeddings tabulated); only executed_frac is a '
                               'fraction of the f32 peak -- the algorithmic rate may exceed the peak because fewer FLOPs are executed'},
        'kernels': kern,
        'kernels_note': 'per-kernel averages come from a separate HIP-event-instrumented pass of 2 steps after the timed region; '
                        f'avg_us = event pair minus the pair overhead measured in the same pass on an empty kernel ({ev_overhead:.1f} us)',
    }
    if roofline:
        out['roofline'] = roofline
    out['config']['canonical_arithmetic'] = eng.tuning.get('canonical', 0) >= 0
    if rank == 0 and world == 1 and args.workload == 'c3' and args.precision == 'f32' and args.size_dist is None and not args.no_secondary:
        del leg
        out['secondary'] = secondary_legs({args.preset: (cfg, eng)}, dev, lib_digest, args.secondary_steps)
        # BASELINE.json's metric reads "(GEOM-drugs size dist)": that figure next to `value` (which north_star asks on fixed-size graphs), at the top level
        g = out['secondary']['geom_size_dist']
        out['value_geom_size_dist'], out['ms_per_step_geom_size_dist'] = g['value'], g['ms_per_step']
        out['config']['workload_size_dist'] = g['workload'] + f"; {g['steps']} timed steps after {g['warmup']} warm-up steps"
    elif world > 1 and args.workload == 'c3' and args.precision == 'f32' and args.size_dist is None and not args.no_secondary:
        del leg        # every rank takes part (barriers); LPT-balanced GEOM sizes, max over ranks
        sdl = size_dist_leg(eng, cfg, world, rank, dev, B, T, args.secondary_steps)
        out['value_geom_size_dist'], out['ms_per_step_geom_size_dist'] = sdl['value_geom_size_dist'], sdl['ms_per_step_geom_size_dist']
        out['config']['workload_size_dist'] = sdl['workload_size_dist']
        out['config']['size_dist_shard_cost_max_over_mean'] = sdl['size_dist_shard_cost_max_over_mean']
    if rank == 0 and world == 1 and not args.no_api_e2e:
        out['api_end_to_end'] = api_end_to_end(args, all_sizes, T, dev, wl['traj'])
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, sd, all_sizes, args.cpu_mols, args.cpu_steps, T, evals, ref_batch=args.cpu_ref_batch)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
