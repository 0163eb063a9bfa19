"""Sampling CLI: the MI355X counterpart of the reference's ``test.py`` (reference test.py:17-55,99-259).

    python -m flowmol_amd.cli --model_dir <dir> | --checkpoint <ckpt> | --preset flowmol3
        [--n_mols 100] [--n_atoms_per_mol N] [--n_timesteps 250] [--max_batch_size 128]
        [--xt_traj] [--ep_traj] [--stochasticity eta] [--hc_thresh p] [--seed s] [--output_file out.sdf]

Under ``torchrun --nproc-per-node N`` the molecules of every batch are sharded over the N GPUs of the node
(``FlowMol.sample_distributed``: one RCCL all-gather per batch) and rank 0 writes the files.

Writes an SDF of the sampled molecules (V2000 blocks written without RDKit; RDKit's writer is used when RDKit
is installed), or with --xt_traj / --ep_traj one ``<stem>_<i>_xt.sdf`` / ``<stem>_<i>_ep.sdf`` per molecule.
``--metrics`` writes ``<stem>_metrics.txt`` / ``.pkl`` like the reference (test.py:153-199) with the metrics that need
no RDKit: valence stability against the dataset's shipped valency table and bond-graph connectivity, counted on the
GPU (flowmol_amd/metrics.py); ``--n_subsets k`` reports mean and 95 % CI over k subsets.  RDKit validity, REOS/rings
and PoseBusters are outside the MI355X hot path.
"""
from __future__ import annotations

import argparse
import math
import pickle
import time
from pathlib import Path

import torch

from .model import FlowMol


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='FlowMol3 sampling on MI355X')
    p.add_argument('--model_dir', type=Path, default=None, help='model directory holding checkpoints/last.ckpt')
    p.add_argument('--checkpoint', type=Path, default=None, help='path to a Lightning checkpoint')
    p.add_argument('--preset', type=str, default=None, help='architecture preset with synthetic weights: ' + ', '.join(sorted(__import__('flowmol_amd.presets', fromlist=['PRESETS']).PRESETS)))
    p.add_argument('--precision', choices=('f32', 'f16x3', 'bf16x3', 'bf16x6'), default='f32',
                   help="(not a reference flag) arithmetic of the big GEMMs: f32 = the reference's (default); f16x3 / bf16x3 / bf16x6 = opt-in split precision on the 16-bit matrix cores (DESIGN.md section 3)")
    p.add_argument('--output_file', type=Path, default=None)
    p.add_argument('--n_mols', type=int, default=100)
    p.add_argument('--n_atoms_per_mol', type=int, default=None)
    p.add_argument('--n_timesteps', type=int, default=250)
    p.add_argument('--xt_traj', action='store_true')
    p.add_argument('--ep_traj', action='store_true')
    p.add_argument('--metrics', action='store_true', help='valence-stability / connectivity metrics of the samples')
    p.add_argument('--n_subsets', type=int, default=None, help='split the samples into subsets for mean / 95%% CI of the metrics')
    p.add_argument('--metrics_dataset', type=str, default=None, help='valency table to use (default: the model\'s dataset)')
    p.add_argument('--max_batch_size', type=int, default=128)
    p.add_argument('--baseline_comparison', action='store_true',
                   help='write (molecules, sampling_time) as a pickle like the reference (test.py:145-150)')
    p.add_argument('--stochasticity', type=float, default=None)
    p.add_argument('--hc_thresh', type=float, default=None)
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--device', type=str, default='cuda:0')
    args = p.parse_args(argv)
    if sum(x is not None for x in (args.model_dir, args.checkpoint, args.preset)) != 1:
        raise ValueError('specify exactly one of --model_dir, --checkpoint, --preset')
    if args.hc_thresh is not None and not (0 <= args.hc_thresh <= 1):
        raise ValueError('hc_thresh must be on the interval [0, 1]')
    return args


def load_model(args, engine_lib=None) -> FlowMol:
    kw = {'_engine_lib': engine_lib} if engine_lib is not None else {}
    if getattr(args, 'precision', 'f32') != 'f32':
        kw['precision'] = args.precision
    if args.preset is not None:
        return FlowMol.from_preset(args.preset, **kw)
    ckpt = args.checkpoint if args.checkpoint is not None else args.model_dir / 'checkpoints' / 'last.ckpt'
    return FlowMol.load_from_checkpoint(ckpt, **kw)


def write_sdf(path: Path, blocks):
    with open(path, 'w') as f:
        for b in blocks:
            f.write(b)
            f.write('$$$$\n')


def write_metrics(args, model, molecules, out: Path):
    """test.py:153-199: whole-set metrics, or mean and 95 % CI over --n_subsets subsets."""
    from .metrics import SampleAnalyzer
    analyzer = SampleAnalyzer(model, dataset=args.metrics_dataset)
    if args.n_subsets is not None and args.n_subsets > 1:
        per = len(molecules) / args.n_subsets
        subs = []
        for i in range(args.n_subsets):
            lo = int(i * per)
            subs.append(analyzer.analyze(molecules[lo:min(int(lo + per), len(molecules))]))
        metrics = {}
        for key in subs[0]:
            vals = torch.tensor([d[key] for d in subs], dtype=torch.float64)
            metrics[key] = float(vals.mean())
            metrics[f'{key}_ci95'] = float(1.96 * vals.std(unbiased=False) / math.sqrt(len(subs)))
    else:
        metrics = analyzer.analyze(molecules)
    txt, pkl = out.parent / f'{out.stem}_metrics.txt', out.parent / f'{out.stem}_metrics.pkl'
    print(f'Writing metrics to {txt} and {pkl}')
    with open(txt, 'w') as f:
        for k, v in metrics.items():
            f.write(f'{k}: {v}\n')
    with open(pkl, 'wb') as f:
        pickle.dump(metrics, f)
    return metrics


def _dist_setup(args):
    """torchrun launch (one process per GPU): RCCL process group, this rank's device.  Returns (world, rank); world is 0
    when the process was not started by a distributed launcher (plain single-GPU run).  A launcher with ONE rank still takes the
    sharded code path (one-rank process group), so `torchrun --nproc-per-node 1` exercises exactly what N ranks run."""
    import os
    if 'WORLD_SIZE' not in os.environ:
        return 0, 0
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.device.startswith('cuda'):
        args.device = f'cuda:{local}'
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group('nccl', device_id=torch.device(args.device))
    elif not dist.is_initialized():
        dist.init_process_group('gloo')
    return dist.get_world_size(), dist.get_rank()


def run(args, engine_lib=None):
    world, rank = _dist_setup(args)
    if args.seed is not None:
        torch.manual_seed(args.seed)        # the reference uses lightning's seed_everything (test.py:70-71)
    model = load_model(args, engine_lib).to(args.device).eval()
    molecules = []
    n_batches = math.ceil(args.n_mols / args.max_batch_size)
    start = time.time()
    for b in range(n_batches):
        bs = min(args.n_mols - len(molecules), args.max_batch_size)
        common = dict(n_timesteps=args.n_timesteps, stochasticity=args.stochasticity, high_confidence_threshold=args.hc_thresh)
        n_atoms = model.sample_n_atoms(bs) if args.n_atoms_per_mol is None else torch.full((bs,), args.n_atoms_per_mol, dtype=torch.long)
        if world == 0:
            molecules.extend(model.sample(n_atoms, device=args.device, xt_traj=args.xt_traj, ep_traj=args.ep_traj, **common))
        else:
            # every rank must shard the SAME size list: rank 0's draw is broadcast; noise streams differ per rank
            import torch.distributed as dist
            nb = n_atoms.to(args.device if args.device.startswith('cuda') else 'cpu')
            dist.broadcast(nb, src=0)
            if args.seed is not None:
                torch.manual_seed(args.seed + 7919 * (b + 1) + rank)
            molecules.extend(model.sample_distributed(nb.cpu(), xt_traj=args.xt_traj, ep_traj=args.ep_traj, **common))      # trajectories: a second gather of the frames
    sampling_time = time.time() - start
    if rank != 0:            # every rank holds the gathered batch; rank 0 writes
        return molecules, sampling_time
    base = args.model_dir if args.model_dir is not None else Path('.')
    if args.output_file is not None:
        out = args.output_file
    elif args.baseline_comparison:
        out = base / 'samples' / f'{base.resolve().name}_baseline_comparison.pkl'          # test.py:138-139
    else:
        out = base / 'samples' / 'sampled_mols.sdf'
    out.parent.mkdir(parents=True, exist_ok=True)
    if args.baseline_comparison:
        # the reference pickles (rdkit_mols, sampling_time) -- its only timing artefact (test.py:145-150, read by
        # fm3_evals/baselines/compute_baseline_comparison.py:38-40).  Same tuple here; without RDKit each molecule is the
        # tensor form the RDKit molecule is built from.
        print(f'Writing molecules to {out}')
        items = []
        for m in molecules:
            rd = m.rdkit_mol
            items.append(rd if rd is not None else m.to_record())
        with open(out, 'wb') as f:
            pickle.dump((items, sampling_time), f)
        return molecules, sampling_time
    if args.metrics:
        write_metrics(args, model, molecules, out)
    if out.suffix != '.sdf':
        raise ValueError('output file must be an sdf file')
    if not (args.xt_traj or args.ep_traj):
        print(f'Writing molecules to {out}')
        write_sdf(out, [m.to_sdf_block() for m in molecules])
    else:
        print('Trajectories requested, writing a separate output file for each molecule trajectory')
        for i, m in enumerate(molecules):
            if args.xt_traj:
                write_sdf(out.parent / f'{out.stem}_{i}_xt{out.suffix}', m.traj_mol_blocks(ep_traj=False))
            if args.ep_traj:
                write_sdf(out.parent / f'{out.stem}_{i}_ep{out.suffix}', m.traj_mol_blocks(ep_traj=True))
        print(f'All molecules written to {out.parent}')
    print(f'sampling_time: {sampling_time:.3f} s for {len(molecules)} molecules '
          f'({len(molecules) / sampling_time:.2f} molecules/s at {args.n_timesteps} timesteps)')
    return molecules, sampling_time


def main(argv=None):
    run(parse_args(argv))


if __name__ == '__main__':
    main()
