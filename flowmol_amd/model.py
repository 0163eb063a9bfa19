"""``FlowMol``: the sampling-side drop-in for the reference's LightningModule
(reference flowmol/models/flowmol.py: ``sample_random_sizes`` 473-486, ``sample`` 489-589,
``sample_n_atoms`` 468-471, ``build_n_atoms_dist`` 461-466) backed by the HIP engine.

    import flowmol_amd as flowmol
    model = flowmol.load_pretrained('flowmol3').cuda().eval()
    mols = model.sample_random_sizes(n_molecules=10, n_timesteps=250)

Training, losses, optimizers and the chemistry metrics are out of scope.
"""
from __future__ import annotations

import json
import os
import pickle
from pathlib import Path
from typing import Dict, List, Optional

import torch

from . import presets
from .config import VFConfig, from_reference_hparams
from .engine import Engine, StepNoise, cat_temp_schedule, forward_weight_schedule, make_step_plan
from .molecule import SampledMolecule
from .weights import synth_state_dict

PKG = Path(__file__).resolve().parent


def load_n_atoms_hist(name: str):
    data = json.loads((PKG / 'data' / 'n_atoms_hist.json').read_text())
    if name not in data:
        raise KeyError(f'no size histogram named {name!r}; have {sorted(data)}')
    return torch.tensor(data[name]['n_atoms'], dtype=torch.int64), torch.tensor(data[name]['counts'], dtype=torch.int64)


def load_marginal_dists(name: str):
    """(p_a, p_c, p_e, p_c_given_a) of a dataset: the reference's data/<set>/train_data_marginal_dists.pt as shipped JSON."""
    data = json.loads((PKG / 'data' / 'marginal_dists.json').read_text())
    if name not in data:
        raise KeyError(f'no marginal distributions named {name!r}; have {sorted(data)}')
    d = data[name]
    return tuple(torch.tensor(d[k], dtype=torch.float32) for k in ('p_a', 'p_c', 'p_e', 'p_c_given_a'))


def simplex_projection(x: torch.Tensor) -> torch.Tensor:
    """Euclidean projection of every row onto the probability simplex (Wang & Carreira-Perpinan, arXiv:1309.1541), the operation the
    reference's blurred barycenter prior applies (priors.py:36-44 -> flowmol/utils/dirflow.py:35-49): with the row sorted descending,
    u, and c_j = (sum_{i<=j} u_i - 1) / j, the threshold is c_rho for rho = #{j : u_j > c_j}; the result is max(x - c_rho, 0)."""
    y = x.reshape(-1, x.shape[-1])
    u, _ = torch.sort(y, dim=-1, descending=True)
    c = (torch.cumsum(u, dim=-1) - 1) / torch.arange(1, y.shape[1] + 1, dtype=y.dtype, device=y.device).unsqueeze(0)
    rho = (u > c).sum(dim=1, keepdim=True)
    tau = torch.gather(c, 1, rho - 1)
    return torch.max(y - tau, torch.zeros_like(y)).view(x.shape)


class _Lenient(pickle.Unpickler):
    """Unpickler for Lightning checkpoints without Lightning installed: unknown classes
    (pytorch_lightning AttributeDict & co.) become plain dict / object stand-ins."""
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except Exception:
            if 'AttributeDict' in name or name.endswith('Dict'):
                return dict
            return type(name, (), {'__setstate__': lambda self, st: self.__dict__.update(st if isinstance(st, dict) else {})})


class _LenientPickle:
    Unpickler = _Lenient
    __name__ = 'lenient_pickle'

    @staticmethod
    def load(f, **kw):
        return _Lenient(f, **kw).load()


def read_checkpoint(path):
    """Lightning-free reader of ``<model_dir>/checkpoints/last.ckpt``: returns (hyper_parameters, state_dict)
    (reference flowmol/__init__.py:54-55 -> FlowMol.load_from_checkpoint)."""
    ck = torch.load(str(path), map_location='cpu', weights_only=False, pickle_module=_LenientPickle)
    if 'state_dict' not in ck or 'hyper_parameters' not in ck:
        raise ValueError(f'{path}: not a Lightning checkpoint (need state_dict and hyper_parameters)')
    return dict(ck['hyper_parameters']), ck['state_dict']


SUPPORTED_PARAMETERIZATIONS = ('ctmc', 'endpoint')
SUPPORTED_SCHEDULES = ('linear', 'cosine')


ENDPOINT_PRIORS = ('gaussian', 'uniform-simplex', 'barycenter', 'biased-simplex', 'marginal', 'c-given-a')     # priors.py:253-262 minus x / ctmc


def check_reference_hparams(hp: dict) -> None:
    """Reject -- loudly, before any weight is touched -- every checkpoint configuration the HIP path does not reproduce, so that
    no model is ever integrated with the wrong schedule or prior.  Missing keys take the REFERENCE's defaults
    (FlowMol.__init__ flowmol.py:29-55: parameterization='endpoint'; InterpolantScheduler interpolant_scheduler.py:9:
    schedule_type='cosine'), so a checkpoint that relies on an unimplemented default is rejected too."""
    par = hp.get('parameterization', 'endpoint')
    if par not in SUPPORTED_PARAMETERIZATIONS:
        raise NotImplementedError(f"parameterization={par!r}: implemented: {SUPPORTED_PARAMETERIZATIONS}")
    isc = hp.get('interpolant_scheduler_config', {}) or {}
    st = isc.get('schedule_type', 'cosine')
    types = [st.get(f, 'cosine') for f in 'xace'] if isinstance(st, dict) else [st]
    bad = sorted({t for t in types if t not in SUPPORTED_SCHEDULES})
    if bad:
        raise NotImplementedError(f'interpolant schedule_type {bad}: implemented: {SUPPORTED_SCHEDULES} '
                                  '(any other schedule would be integrated with the wrong x and unmasking coefficients)')
    pc = hp.get('prior_config', {}) or {}
    xt = (pc.get('x', {}) or {}).get('type', 'centered-normal')
    if xt != 'centered-normal':
        raise NotImplementedError(f"position prior {xt!r}: only 'centered-normal' is implemented (a 'gaussian' prior would be centred silently)")
    for mod in ('a', 'c', 'e'):
        mt = (pc.get(mod, {}) or {}).get('type')
        if par == 'ctmc' and (mt or 'ctmc') != 'ctmc':
            raise NotImplementedError('only ctmc masked priors are supported for CTMC models (as in the reference, flowmol.py:189-193)')
        if par == 'endpoint' and mt not in ENDPOINT_PRIORS:
            raise NotImplementedError(f"categorical prior {mt!r} of an endpoint model: implemented: {ENDPOINT_PRIORS}")
        if par == 'endpoint' and mt == 'c-given-a' and mod != 'c':
            raise ValueError("the 'c-given-a' prior is the charge prior (flowmol.py:437-438)")
    if hp.get('exclude_charges', False):
        raise NotImplementedError('exclude_charges=True is not implemented (no shipped v3 model uses it)')


class FlowMol:
    canonical_feat_order = ['x', 'a', 'c', 'e']

    def __init__(self, cfg: VFConfig, state_dict: Dict[str, torch.Tensor], prefix: str = 'vector_field.',
                 n_atoms_hist: Optional[str] = None, _engine_lib=None, precision: Optional[str] = None, canonical: bool = True):
        self.cfg = cfg.validate()
        # canonical arithmetic (fm_config.canonical, default on): a molecule's coordinates and tokens are bit-for-bit independent of the batch it is sampled in
        # (size, position, sharding over GPUs) -- the reference's semantics, where every reduction is per molecule.  canonical=False: the one launch
        # choice that selects another summation order (the pair slab) follows the batch size too; differently composed batches then agree to f32 summation
        # order only.  (Up to round 5 this was a "latency mode"; since the small-batch kernels keep the canonical order it gains ~1 %: 0.55 vs 0.56 ms per step.)
        self.canonical = bool(canonical)
        self.precision = precision or 'f32'  # 'f16x3' / 'bf16x3' / 'bf16x6' = opt-in split precision (Engine); explicit argument only, recorded in last_timing
        self._sd = state_dict
        self._prefix = prefix
        self._lib = _engine_lib
        self.atom_type_map = list(cfg.atom_type_map)
        self.n_atom_types = cfg.n_atom_types
        self.n_atom_charges = cfg.n_charges
        self.n_bond_types = cfg.n_bond_types
        self.fake_atoms = cfg.fake_atoms
        self.explicit_aromaticity = cfg.explicit_aromaticity
        self.default_n_timesteps = cfg.default_n_timesteps
        self.parameterization = cfg.parameterization
        self.device = torch.device('cpu')
        self._engine: Optional[Engine] = None
        self.build_n_atoms_dist(n_atoms_hist or cfg.n_atoms_hist)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_preset(cls, name: str = 'flowmol3', seed: int = 0, **kw) -> "FlowMol":
        """Architecture ``name`` with deterministic weights-by-name (no checkpoint ships with the reference)."""
        cfg = presets.PRESETS[name]()
        sd = {'vector_field.' + k: v for k, v in synth_state_dict(cfg, seed).items()}
        return cls(cfg, sd, **kw)

    @classmethod
    def load_from_checkpoint(cls, ckpt_path, **kw) -> "FlowMol":
        hp, sd = read_checkpoint(ckpt_path)
        check_reference_hparams(hp)
        return cls(from_reference_hparams(hp), sd, **kw)

    # nn.Module-ish conveniences of the documented usage (readme.md:44-49)
    def to(self, device) -> "FlowMol":
        device = torch.device(device)
        if device != self.device or self._engine is None:
            if self._engine is not None:
                self._engine.close()
                self._engine = None
            self.device = device
        if device.type == 'cuda':
            self.engine          # like nn.Module.to(): the weights are repacked and uploaded NOW (fm_create), not inside the first sample() call
        return self

    def cuda(self, device=None) -> "FlowMol":
        return self.to('cuda:0' if device is None else (f'cuda:{device}' if isinstance(device, int) else device))

    def eval(self) -> "FlowMol":
        return self

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            if self.device.type != 'cuda' and self._lib is None:
                raise RuntimeError('flowmol_amd runs on MI355X only: move the model to a GPU with .cuda() '
                                   '(there is no CPU implementation of the sampling path)')
            self._engine = Engine(self.cfg, self._sd, device=self.device, prefix=self._prefix, lib=self._lib, precision=self.precision,
                                  tuning=None if self.canonical else {'canonical': -1})
        return self._engine

    # ------------------------------------------------------------------ sizes
    def build_n_atoms_dist(self, n_atoms_hist):
        """flowmol.py:461-466."""
        if isinstance(n_atoms_hist, (str, Path)) and str(n_atoms_hist).endswith('.pt'):
            n_atoms, counts = torch.load(str(n_atoms_hist))
        else:
            n_atoms, counts = load_n_atoms_hist(str(n_atoms_hist))
        self.n_atoms_dist = torch.distributions.Categorical(probs=counts / counts.sum())
        self.n_atoms_map = n_atoms

    def sample_n_atoms(self, n_molecules: int, **kwargs):
        """flowmol.py:468-471 (draws on the CPU generator like the reference)."""
        return self.n_atoms_map[self.n_atoms_dist.sample((n_molecules,), **kwargs)]

    # ------------------------------------------------------------------ sampling
    def sample_random_sizes(self, n_molecules: int, device=None, stochasticity=None, high_confidence_threshold=None,
                            xt_traj=False, ep_traj=False, **kwargs) -> List[SampledMolecule]:
        atoms_per_molecule = self.sample_n_atoms(n_molecules)
        return self.sample(atoms_per_molecule, device=device, stochasticity=stochasticity,
                           high_confidence_threshold=high_confidence_threshold, xt_traj=xt_traj, ep_traj=ep_traj, **kwargs)

    @torch.no_grad()
    def sample_distributed(self, n_atoms: torch.Tensor, n_timesteps: int = None, group=None, return_tensors=False,
                           noise: str = 'per_rank', **kwargs):
        """Multi-GPU sampling (no reference counterpart): every rank of ``group`` calls this with the
        SAME ``n_atoms``; the molecules are dealt to the ranks by cost (``shard.partition_lpt``), each rank integrates its
        shard on its own GPU with no communication, and ONE all-gather of the packed results (``shard.gather_results``,
        RCCL over xGMI with backend "nccl") gives every rank the full batch in the caller's order.

        ``noise='per_rank'``: each rank draws only its shard's noise from its own torch RNG stream (seed it per rank);
        results depend on the world size.  ``noise='replicated'`` (parity mode): every rank, seeded identically, draws the
        full batch's prior and per-step noise and keeps its molecules' rows, so the result reproduces the single-GPU
        ``sample(n_atoms)`` with that seed (to float summation order), at world_size times the (cheap) RNG work.
        ``noise='philox'`` (performance mode): prior and CTMC noise are drawn INSIDE the kernels from per-molecule Philox4x32-10
        streams keyed by (seed, original molecule index, step, modality): no noise tensors at all, and every molecule's
        trajectory is the same for any world size or batch composition -- bit for bit with canonical arithmetic (the default).

        Everything one ``sample()`` call of the reference takes (flowmol.py:489-493) shards too: ``prior`` (a reference-format prior dict of the WHOLE
        batch: every rank keeps its molecules' rows, flowmol.py:534-545) and ``xt_traj`` / ``ep_traj`` (flowmol.py:564-589, test.py:212-257): each rank
        records its shard's frames in HBM in the compact token format and a SECOND all-gather (``shard.gather_frames``; only when trajectories
        are asked for) gives every rank the full batch's frames in the caller's order."""
        import torch.distributed as dist
        from . import shard
        if noise not in ('per_rank', 'replicated', 'philox'):
            raise ValueError(f"noise must be 'per_rank', 'replicated' or 'philox', got {noise!r}")
        n_atoms = torch.as_tensor(n_atoms).detach().to('cpu', torch.int64)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        parts = shard.partition_lpt(n_atoms, world)
        dev = self.engine.device
        xt_traj, ep_traj = bool(kwargs.pop('xt_traj', False)), bool(kwargs.pop('ep_traj', False))
        visualize = xt_traj or ep_traj
        if kwargs.get('prior') is not None:
            kwargs['prior'] = shard.slice_prior(kwargs['prior'], n_atoms, parts[rank])
        if noise == 'replicated':
            pairs = n_atoms * (n_atoms - 1) // 2
            mine = parts[rank]
            node_rows = shard._ranges((torch.cumsum(n_atoms, 0) - n_atoms)[mine], n_atoms[mine]).to(dev)
            pair_rows = shard._ranges((torch.cumsum(pairs, 0) - pairs)[mine], pairs[mine]).to(dev)
            kwargs['_rows'] = (int(n_atoms.sum()), int(pairs.sum()), node_rows, pair_rows)
        if noise == 'philox':
            # performance mode of SURVEY.md section 8e: per-molecule counter-based streams keyed by (seed, ORIGINAL molecule index, step,
            # modality); rank 0's seed is broadcast; nothing but the shard's own noise is generated, and it is generated in-kernel
            seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
            sd = seed.to(dev if dist.get_backend(group) == 'nccl' else 'cpu')
            dist.broadcast(sd, src=0, group=group)
            kwargs.update(rng='philox', _philox=int(sd.item()), _mol_ids=parts[rank])
        local_frames = None
        if len(parts[rank]):
            got = self.sample(n_atoms[parts[rank]], n_timesteps=n_timesteps, return_tensors='device', xt_traj=visualize, _frames=visualize, **kwargs)   # stays in HBM
            local, local_frames = got[0], (got[2] if visualize else None)
        else:
            i32 = dict(dtype=torch.int32, device=dev)
            local = {'x': torch.zeros(0, 3, device=dev), 'a': torch.zeros(0, **i32), 'c': torch.zeros(0, **i32), 'e': torch.zeros(0, **i32)}
        full_dev = shard.gather_results(local, n_atoms, parts, group=group)
        frames_dev = None
        if visualize:
            ts = kwargs.get('tspan')
            n_frames = (self.default_n_timesteps if n_timesteps is None else n_timesteps) if ts is None else int(ts.shape[0])
            frames_dev = shard.gather_frames(local_frames, n_atoms, parts, n_frames, dev, group=group)
        if return_tensors == 'device':
            return (full_dev, n_atoms, frames_dev) if visualize else (full_dev, n_atoms)
        full = _to_host(full_dev)            # one packed device->host copy of the whole gathered batch (1.7 KB/molecule)
        frames = {k: v.cpu() for k, v in frames_dev.items()} if visualize else None
        if return_tensors:
            return (full, n_atoms, frames) if visualize else (full, n_atoms)
        return self._package(full, n_atoms, frames, xt_traj, ep_traj)

    @torch.no_grad()
    def sample(self, n_atoms: torch.Tensor, n_timesteps: int = None, device=None, stochasticity=None,
               high_confidence_threshold=None, xt_traj=False, ep_traj=False, prior=None, return_tensors=False,
               **kwargs):
        """Sample molecules with the given numbers of atoms (flowmol.py:489-589).

        RNG use mirrors the reference: ``randn(N,3)`` on the device for the position prior, then per step
        and per modality (a, c, e) ``Exp(1)`` of shape (rows, K), ``rand(rows)``, ``rand(rows)`` (the last
        one skipped on the final step)."""
        if device is not None and torch.device(device) != self.device:
            self.to(device)
        eng = self.engine
        dev = eng.device
        n_timesteps = self.default_n_timesteps if n_timesteps is None else n_timesteps
        if self.cfg.parameterization == 'endpoint':
            return self._sample_endpoint(n_atoms, n_timesteps, xt_traj, ep_traj, prior, return_tensors, kwargs)
        # integrator variants of CTMCVectorField.integrate/step (ctmc_vector_field.py:145-156,287-315): all optional
        dfm_type = kwargs.get('dfm_type') or self.cfg.dfm_type
        if dfm_type not in ('campbell', 'gat'):
            raise ValueError(f"Invalid dfm_type: {dfm_type}")
        unknown = set(kwargs) - {'dfm_type', 'tspan', 'cat_temp_func', 'forward_weight_func', 'inv_temp_func', '_rows', '_noise_for_step', 'rng', '_philox', '_mol_ids', '_frames'}
        if unknown:
            raise TypeError(f'sample() got unexpected keyword arguments {sorted(unknown)}')
        visualize = bool(xt_traj or ep_traj)
        n_atoms = torch.as_tensor(n_atoms).detach().to('cpu', torch.int64)
        eng.bind(n_atoms)
        N, U = eng.N, eng.U
        cfg = self.cfg
        # ---- prior (flowmol.py:417-448 / 534-545)
        rows = kwargs.get('_rows')     # sample_distributed(noise='replicated'): (N_full, U_full, node rows, pair rows) of this shard
        # rng='philox': position prior and CTMC noise come from per-molecule counter-based streams inside the kernels (no noise
        # tensors; results independent of batch composition / sharding).  rng='torch' (default): the reference's draws from torch's generator.
        rng = kwargs.get('rng', 'torch')
        if rng not in ('torch', 'philox'):
            raise ValueError(f"rng must be 'torch' or 'philox', got {rng!r}")
        philox_seed = None
        if rng == 'philox':
            if dfm_type != 'campbell' or rows is not None or kwargs.get('_noise_for_step') is not None:
                raise NotImplementedError("rng='philox' covers the default campbell integrator")
            philox_seed = kwargs.get('_philox')
            if philox_seed is None:                       # one 62-bit seed per call from torch's CPU generator: torch.manual_seed controls it
                philox_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            eng.set_molecule_ids(kwargs.get('_mol_ids'))
        if prior is None and philox_seed is not None:
            state = eng.prior_state(eng.prior_philox(philox_seed))
        elif prior is None:
            x0 = torch.randn(N, 3, device=dev) if rows is None else torch.randn(rows[0], 3, device=dev)[rows[2]].contiguous()
            eng.remove_com(x0)
            state = eng.prior_state(x0)
        else:
            state = self._state_from_prior(prior)
        eta = cfg.stochasticity if stochasticity is None else stochasticity
        hc = cfg.high_confidence_threshold if high_confidence_threshold is None else high_confidence_threshold
        ctf = kwargs.get('cat_temp_func') or cat_temp_schedule(cfg)
        fwf = kwargs.get('forward_weight_func') or forward_weight_schedule(cfg)
        plan = make_step_plan(n_timesteps, eta, hc, ctf, tspan=kwargs.get('tspan'), dfm_type=dfm_type,
                              forward_weight_func=fwf, inv_temp_func=kwargs.get('inv_temp_func'), philox_seed=philox_seed,
                              schedule_type=cfg.schedule_type, cosine_params=cfg.cosine_params)
        n_steps = len(plan.scalars)
        traj = None
        if visualize:
            i32 = dict(dtype=torch.int32, device=dev)
            traj = {'x': torch.empty(n_steps, N, 3, device=dev), 'a': torch.empty(n_steps, N, **i32),
                    'c': torch.empty(n_steps, N, **i32), 'e': torch.empty(n_steps, U, **i32),
                    'x1': torch.empty(n_steps, N, 3, device=dev), 'a1': torch.empty(n_steps, N, **i32),
                    'c1': torch.empty(n_steps, N, **i32), 'e1': torch.empty(n_steps, U, **i32)}
            init = {k: state[f'{k}_t'].clone() for k in 'xace'}

        def noise_for_step(i, last):
            if rows is None:
                return StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, last, dev, dfm_type=dfm_type)
            full = StepNoise.draw(rows[0], rows[1], cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, last, dev, dfm_type=dfm_type)
            return full.take_rows(rows[2], rows[3])

        import time
        t0 = time.perf_counter()
        # _noise_for_step(i, last) -> StepNoise: recorded draws instead of torch's generator (parity tests drive the public API with
        # the oracle's RNG tape); never set by the product itself
        eng.integrate(state, plan, None if philox_seed is not None else (kwargs.get('_noise_for_step') or noise_for_step), traj=traj)          # synchronises at the end
        t1 = time.perf_counter()
        out_dev = {k: state[f'{k}_t'] for k in 'xace'}
        if return_tensors == 'device':
            self.last_timing = {'integrate': t1 - t0, 'precision': eng.precision}
            if kwargs.get('_frames'):          # sample_distributed: this shard's frames stay in HBM for the second gather
                return out_dev, n_atoms, (_frames_of(init, traj) if visualize else None)
            return out_dev, n_atoms
        out = _to_host(out_dev)
        t2 = time.perf_counter()
        self.last_timing = {'integrate': t1 - t0, 'to_host': t2 - t1, 'precision': eng.precision}
        if return_tensors:
            return out, n_atoms
        frames = {k: v.cpu() for k, v in _frames_of(init, traj).items()} if visualize else None
        mols = self._package(out, n_atoms, frames, xt_traj, ep_traj)
        self.last_timing['package'] = time.perf_counter() - t2
        return mols

    # ------------------------------------------------------------------ endpoint-parameterised models
    @staticmethod
    def _categorical_prior(kind: str, n: int, d: int, kw: dict, a_0: Optional[torch.Tensor] = None, default_p=None) -> torch.Tensor:
        """The reference's categorical prior functions on the CPU generator, as FlowMol.sample_prior calls them (priors.py:8-107;
        flowmol.py:426-441) -- same draws in the same order.  'marginal' / 'c-given-a' take their distribution from the prior kwargs
        (`p` / `p_c_given_a`, as the reference would) or, when absent, from the shipped marginals of the model's dataset (`default_p`)."""
        F = torch.nn.functional
        if kind == 'gaussian':
            p = torch.randn(n, d) * kw.get('std', 1.0)
            return p + 1 / d if kw.get('simplex_center', False) else p
        if kind == 'uniform-simplex':
            sample = torch.distributions.Exponential(torch.tensor(1.0)).sample((n, d))
            return sample / sample.sum(dim=1, keepdim=True)
        if kind == 'barycenter':
            p = torch.ones(n, d) / d
            blur = kw.get('blur', 0.0)
            if blur != 0.0:
                p = simplex_projection(p + torch.randn_like(p) * blur)
            return p
        if kind == 'biased-simplex':          # priors.py:47-56
            vp, std, vi = kw.get('vertex_prob', 0.75), kw.get('std', 0.2), kw.get('vertex_idx', 0)
            mu = torch.ones(d) * ((1 - vp) / (d - 1))
            mu[vi] = vp
            return F.softmax((mu.unsqueeze(0) + torch.randn(n, d) * std) / (1 / d), dim=1)
        if kind in ('marginal', 'c-given-a'):
            key = 'p' if kind == 'marginal' else 'p_c_given_a'
            p = kw.get(key, default_p)
            if p is None:
                raise ValueError(f"prior {kind!r} needs kwargs[{key!r}] (or a dataset whose shipped marginals have {d} categories)")
            p = torch.as_tensor(p, dtype=torch.float32)
            if p.shape[-1] != d:
                raise ValueError(f"prior {kind!r}: distribution has {p.shape[-1]} categories, the model {d}")
            if kind == 'marginal':            # priors.py:67-79
                idx = torch.multinomial(p, n, replacement=True)
            else:                             # priors.py:81-98: charges conditioned on the sampled atom-type prior
                if a_0 is None:
                    raise ValueError("the 'c-given-a' prior needs the atom-type prior")
                idx = torch.multinomial(p[a_0.argmax(dim=1)], 1, replacement=True).squeeze(-1)
            one = F.one_hot(idx, num_classes=d).float()
            blur = kw.get('blur')
            if blur is not None:
                one = F.softmax((one + torch.randn_like(one) * blur) / (1 / d), dim=1)
            return one
        raise NotImplementedError(f'prior type {kind!r}')

    def _default_marginal(self, feat: str, d: int):
        """Shipped marginals of the model's dataset for the 'marginal' / 'c-given-a' priors, when their category count fits."""
        try:
            p_a, p_c, p_e, p_ca = load_marginal_dists(self.cfg.n_atoms_hist)
        except KeyError:
            return None
        p = {'a': p_a, 'c': p_c, 'e': p_e, 'c|a': p_ca}[feat]
        return p if p.shape[-1] == d else None

    def _sample_endpoint(self, n_atoms, n_timesteps, xt_traj, ep_traj, prior, return_tensors, kwargs):
        """FlowMol.sample for parameterization='endpoint' (flowmol.py:489-589 with EndpointVectorField.integrate, vector_field.py:388-499):
        the categorical modalities are continuous vectors integrated with the same Euler step as the positions; the sampled
        molecule takes their argmax.  RNG order = the reference's: randn(N,3) on the device, then a, c, e priors on the CPU generator."""
        if kwargs.get('rng', 'torch') != 'torch' or '_philox' in kwargs:
            raise NotImplementedError("per-molecule Philox noise covers CTMC models; endpoint models draw their priors with torch (noise='per_rank' / 'replicated')")
        unknown = set(kwargs) - {'inv_temp_func', 'tspan', '_rows', 'rng', '_frames'}
        if unknown:
            raise TypeError(f'sample() got unexpected keyword arguments {sorted(unknown)}')
        eng, cfg = self.engine, self.cfg
        dev = eng.device
        n_atoms = torch.as_tensor(n_atoms).detach().to('cpu', torch.int64)
        eng.bind(n_atoms)
        N, U = eng.N, eng.U
        rows = kwargs.get('_rows')      # sample_distributed(noise='replicated'): (N_full, U_full, node rows, pair rows) -- draw the full batch's priors, keep this shard's
        if prior is None:
            nN, nU = (rows[0], rows[1]) if rows is not None else (N, U)
            x0 = torch.randn(nN, 3, device=dev)
            pk = cfg.prior_kwargs
            a0 = self._categorical_prior(cfg.prior_types['a'], nN, cfg.n_atom_types, pk.get('a', {}), default_p=self._default_marginal('a', cfg.n_atom_types))
            c_kind = cfg.prior_types['c']
            c0 = self._categorical_prior(c_kind, nN, cfg.n_charges, pk.get('c', {}), a_0=a0,
                                         default_p=self._default_marginal('c|a' if c_kind == 'c-given-a' else 'c', cfg.n_charges))
            e0 = self._categorical_prior(cfg.prior_types['e'], nU, cfg.n_bond_types, pk.get('e', {}), default_p=self._default_marginal('e', cfg.n_bond_types))
            if rows is not None:
                x0 = x0[rows[2]].contiguous()
                a0, c0, e0 = a0[rows[2].cpu()], c0[rows[2].cpu()], e0[rows[3].cpu()]
            eng.remove_com(x0)          # per molecule, so centring after the row selection equals centring the full batch
        else:
            x0, a0, c0 = prior['x_0'], prior['a_0'], prior['c_0']
            # the reference adapts a prior drawn with / without the fake-atom column to the model (flowmol.py:540-545)
            if prior.get('fake_atoms', False) and not self.fake_atoms:
                a0 = a0[:, 1:]
            elif not prior.get('fake_atoms', False) and self.fake_atoms and a0.shape[-1] == cfg.n_atom_types - 1:
                a0 = torch.cat([torch.zeros(a0.shape[0], 1, device=a0.device, dtype=a0.dtype), a0], dim=-1)
            if a0.shape != (N, cfg.n_atom_types) or c0.shape != (N, cfg.n_charges) or tuple(x0.shape) != (N, 3):
                raise ValueError(f"prior shapes x_0 {tuple(x0.shape)}, a_0 {tuple(a0.shape)}, c_0 {tuple(c0.shape)} do not fit the batch "
                                 f"({N} atoms, {cfg.n_atom_types} atom types, {cfg.n_charges} charges)")
            e0 = prior['e_0']
            if e0.shape[0] == eng.E:          # directed edges, upper block first per molecule -> keep the upper halves
                idx, off = [], 0
                for k in n_atoms.tolist():
                    u = k * (k - 1) // 2
                    idx.append(torch.arange(off, off + u))
                    off += 2 * u
                e0 = e0[torch.cat(idx)]
        state = eng.make_dense_state(x0, a0, c0, e0)
        visualize = bool(xt_traj or ep_traj)
        traj = None
        if visualize:        # frames as category indices (the molecule of a frame is its argmax, molecule_builder.py:231-247) + fp32 coordinates
            ts = kwargs.get('tspan')
            n_steps = (n_timesteps if ts is None else int(ts.shape[0])) - 1
            i32 = dict(dtype=torch.int32, device=dev)
            traj = {'x': torch.empty(n_steps, N, 3, device=dev), 'a': torch.empty(n_steps, N, **i32),
                    'c': torch.empty(n_steps, N, **i32), 'e': torch.empty(n_steps, U, **i32),
                    'x1': torch.empty(n_steps, N, 3, device=dev), 'a1': torch.empty(n_steps, N, **i32),
                    'c1': torch.empty(n_steps, N, **i32), 'e1': torch.empty(n_steps, U, **i32)}
            init = {'x': state['x_t'].clone(), 'a': state['a_t'].argmax(-1).int(), 'c': state['c_t'].argmax(-1).int(), 'e': state['e_t'].argmax(-1).int()}
        import time
        t0 = time.perf_counter()
        eng.integrate_endpoint(state, n_timesteps, inv_temp_func=kwargs.get('inv_temp_func'), tspan=kwargs.get('tspan'), traj=traj)
        self.last_timing = {'integrate': time.perf_counter() - t0, 'precision': eng.precision}
        if return_tensors == 'dense':
            return {k: state[f'{k}_t'] for k in 'xace'}, n_atoms
        out_dev = {'x': state['x_t'], 'a': state['a_t'].argmax(-1).int(), 'c': state['c_t'].argmax(-1).int(), 'e': state['e_t'].argmax(-1).int()}
        if return_tensors == 'device':
            if kwargs.get('_frames'):
                return out_dev, n_atoms, (_frames_of(init, traj) if visualize else None)
            return out_dev, n_atoms
        out = _to_host(out_dev)
        if return_tensors:
            return out, n_atoms
        frames = {k: v.cpu() for k, v in _frames_of(init, traj).items()} if visualize else None
        return self._package(out, n_atoms, frames, xt_traj, ep_traj)

    # ------------------------------------------------------------------ helpers
    def _state_from_prior(self, prior):
        """Reference-format prior dict: x_0 (N,3), a_0/c_0 one-hot (N,*), e_0 one-hot (E,*) in reference edge order."""
        eng = self.engine
        a0 = prior['a_0']
        if prior.get('fake_atoms', False) and not self.fake_atoms:
            a0 = a0[:, 1:]
        elif not prior.get('fake_atoms', False) and self.fake_atoms:
            a0 = torch.cat([torch.zeros(a0.shape[0], 1, device=a0.device), a0], dim=-1)
        e0 = prior['e_0']
        n = eng.n_atoms
        if e0.shape[0] == eng.E:      # directed edges, upper block first per molecule -> keep the upper halves
            idx, off = [], 0
            for k in n.tolist():
                u = k * (k - 1) // 2
                idx.append(torch.arange(off, off + u))
                off += 2 * u
            e0 = e0[torch.cat(idx)]
        return eng.make_state(prior['x_0'], a0.argmax(-1), prior['c_0'].argmax(-1), e0.argmax(-1))

    def _package(self, out, n_atoms, frames, xt_traj, ep_traj) -> List[SampledMolecule]:
        """Host tensors of the whole batch -> one SampledMolecule per molecule, in batch order (flowmol.py:564-589)."""
        sizes = n_atoms.tolist()
        pairs = [n * (n - 1) // 2 for n in sizes]
        xs, as_, cs = torch.split(out['x'], sizes), torch.split(out['a'], sizes), torch.split(out['c'], sizes)
        es = torch.split(out['e'], pairs)
        fr = None
        if frames is not None:
            fr = {k: torch.split(v, pairs if k.startswith('e') else sizes, dim=1) for k, v in frames.items()}
        mols = []
        for i in range(len(sizes)):
            tf = {k: v[i] for k, v in fr.items()} if fr is not None else None
            mols.append(SampledMolecule(xs[i], as_[i], cs[i], es[i], self.atom_type_map, fake_atoms=self.fake_atoms,
                                        ctmc_mol=self.cfg.has_mask, explicit_aromaticity=self.explicit_aromaticity, traj_frames=tf,
                                        build_xt_traj=xt_traj, build_ep_traj=ep_traj, n_charges=self.n_atom_charges))
        return mols


def _frames_of(init: Dict[str, torch.Tensor], traj: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Compact frames of a visualised run on the device: 'x', 'a', 'c', 'e' = the prior + the state after every step (T frames), '*_1_pred' = every step's
    endpoint prediction (T - 1 frames) -- the frame set of the reference (ctmc_vector_field.py:188-202,235-283), tokens instead of one-hots."""
    frames = {k: torch.cat([init[k].unsqueeze(0), traj[k]]) for k in 'xace'}
    frames.update({f'{k}_1_pred': traj[f'{k}1'] for k in 'xace'})
    return frames


def _to_host(dev_out: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Final state of a batch, device -> host, as ONE packed copy (x fp32 | a, c, e as bytes: 14 B/atom + 1 B/pair)."""
    from .shard import pack_results, unpack_results
    if dev_out['x'].device.type == 'cpu':
        return {k: v for k, v in dev_out.items()}
    N, U = int(dev_out['x'].shape[0]), int(dev_out['e'].shape[0])
    return unpack_results(pack_results(dev_out['x'], dev_out['a'], dev_out['c'], dev_out['e']).cpu(), N, U)


# names accepted by the reference's load_pretrained (flowmol/__init__.py:5-28)
pretrained_model_names = [
    'flowmol3', 'fm3_nodistort', 'fm3_none', 'fm3_ahigh', 'fm3_alow', 'fm3_chigh', 'fm3_clow', 'fm3_distort_extreme',
    'fm3_distort_highp', 'fm3_distort_hight', 'fm3_distort_lowp', 'fm3_distort_lowt', 'fm3_ehigh', 'fm3_elow',
    'fm3_fa_highp', 'fm3_fa_highstd', 'fm3_fa_lowp', 'fm3_fa_lowstd', 'fm3_scprop_high', 'fm3_scprop_low',
    'fm3_xhigh', 'fm3_xlow',
]


def load_pretrained(model_name: str = 'flowmol3', precision: Optional[str] = None) -> FlowMol:
    """Load ``<models_dir>/<model_name>/checkpoints/last.ckpt`` (reference flowmol/__init__.py:30-56).  ``precision`` (not a reference argument;
    None = 'f32', the reference's arithmetic) selects an opt-in split-precision mode of the engine ('f16x3', 'bf16x3', 'bf16x6').

    ``models_dir`` is ``$FLOWMOL_MODELS_DIR`` or ``flowmol_amd/trained_models``.  The reference downloads
    missing models with wget; there is no network in this environment, so a missing directory is an error
    that says where to put the files (``FlowMol.from_preset`` gives the same architecture with synthetic weights)."""
    if model_name not in pretrained_model_names:
        raise ValueError(f'Model {model_name} not found. Supported models: {pretrained_model_names}')
    root = Path(os.environ.get('FLOWMOL_MODELS_DIR', PKG / 'trained_models'))
    ckpt = root / model_name / 'checkpoints' / 'last.ckpt'
    if not ckpt.exists():
        raise FileNotFoundError(
            f'{ckpt} not found. Download the model directory from https://bits.csb.pitt.edu/files/FlowMol/trained_models_v3.1/'
            f'{model_name}/ into {root} (or set FLOWMOL_MODELS_DIR); this build has no network access and does not wget.')
    kw = {}
    if 'qm9' in model_name:
        kw['n_atoms_hist'] = 'qm9'
    if precision is not None:
        kw['precision'] = precision
    return FlowMol.load_from_checkpoint(ckpt, **kw)
