"""Python host side of the HIP engine: owns the ``fm_ctx``, the per-batch workspace and the state
tensors, and mirrors the call structure of the reference's ``CTMCVectorField`` for the sampling path
(reference flowmol/models/ctmc_vector_field.py:145-411, flowmol/models/vector_field.py:212-293).

PyTorch is used for device memory, the current stream and (optionally) noise generation only; all
arithmetic of the hot path happens in libflowmol_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import _lib
from ._lib import (FM_DFM_CAMPBELL, FM_DFM_GAT, FM_NOISE_PHILOX, FlowMolHipError, fm_dense_state, fm_endpoint_scalars, fm_config, fm_dst, fm_sampled, fm_state, fm_step_noise, fm_step_scalars,
                   fm_tensor_desc, fm_traj_sink)
from .config import VFConfig
from .weights import check_state_dict, state_dict_shapes


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def time_embedding_host(t: float, dim: int) -> torch.Tensor:
    """Sinusoidal time embedding of one scalar t, float32 on the host
    (reference flowmol/utils/embedding.py:5-17; max_positions=1000).  dim == 1 -> raw t."""
    tt = torch.tensor([t], dtype=torch.float32)
    if dim == 1:
        return tt.clone()
    ts = tt * 1000
    half = dim // 2
    emb = math.log(1000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = ts.float()[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb[0].contiguous()


@dataclass
class StepPlan:
    """Per-step scalars computed with the reference's float32 tensor arithmetic
    (ctmc_vector_field.py:170-176,205-233,331-334,430-434)."""
    t: torch.Tensor                # (T,) float32 time points
    scalars: List[fm_step_scalars]


def cat_temp_schedule(cfg: VFConfig) -> Callable:
    """CTMCVectorField.build_cat_temp_schedule (ctmc_vector_field.py:71-82) for the configured schedule."""
    sched = cfg.cat_temperature_schedule
    if sched == 'decay':
        mx, a = cfg.cat_temp_decay_max, cfg.cat_temp_decay_a
        return lambda t: mx * torch.pow(1 - t, a)
    val = cfg.cat_temperature if isinstance(sched, str) else sched
    return lambda t: val


def forward_weight_schedule(cfg: VFConfig) -> Callable:
    """CTMCVectorField.build_fw_schedule (ctmc_vector_field.py:84-95)."""
    sched = cfg.forward_weight_schedule
    if sched == 'beta':
        a, b, mx = cfg.fw_beta_a, cfg.fw_beta_b, cfg.fw_beta_max
        return lambda t: 1 + mx * torch.pow(t, a) * torch.pow(1 - t, b)
    return lambda t: sched


def _f32(v) -> float:
    """A Python number or 0-dim tensor as the float32 value a torch op on float32 tensors would use."""
    return float(v.to(torch.float32)) if torch.is_tensor(v) else float(torch.tensor(v, dtype=torch.float32))


def alpha_tables(t: torch.Tensor, schedule_type=None, cosine_params=None):
    """InterpolantScheduler.alpha_t / alpha_t_prime (interpolant_scheduler.py:97-153) for the time points ``t`` (T,), columns in the
    canonical order x, a, c, e, with the reference's float32 tensor arithmetic -- INCLUDING its side effect: the cosine
    derivative clamps ``t`` in place to >= 1e-9 (:140-141), after alpha_t was taken from the unclamped tensor
    (ctmc_vector_field.py:175-176), so a cosine-scheduled trajectory starts at t = 1e-9 and has no bootstrap evaluation."""
    schedule_type = schedule_type or {}
    cosine_params = cosine_params or {}
    nus = {k: torch.tensor(cosine_params[k]).unsqueeze(0) for k in cosine_params}        # interpolant_scheduler.py:53-55
    cols = []
    for k in 'xace':
        if schedule_type.get(k, 'linear') == 'cosine':
            cols.append(1 - torch.cos(torch.pi * 0.5 * torch.pow(t.unsqueeze(-1), nus[k])).square())
        else:
            cols.append(t.unsqueeze(-1))
    alpha = torch.cat(cols, dim=1)
    cols = []
    for k in 'xace':
        if schedule_type.get(k, 'linear') == 'cosine':
            t = torch.clamp_(t, min=1e-9)
            tt = t.unsqueeze(-1)
            cols.append(torch.pi * 0.5 * torch.sin(torch.pi * torch.pow(tt, nus[k])) * nus[k] * torch.pow(tt, nus[k] - 1))
        else:
            cols.append(torch.ones_like(t).unsqueeze(-1))
    return alpha, torch.cat(cols, dim=1)


def make_step_plan(n_timesteps: int, eta: float, hc_thresh: float, cat_temperature,
                   tspan: Optional[torch.Tensor] = None, dfm_type: str = 'campbell', forward_weight_func: Optional[Callable] = None,
                   inv_temp_func: Optional[Callable] = None, philox_seed: Optional[int] = None,
                   schedule_type=None, cosine_params=None) -> StepPlan:
    """Per-step scalars of CTMCVectorField.integrate/step (ctmc_vector_field.py:169-178, 287-340), computed with the
    reference's own float32 tensor arithmetic.  ``cat_temperature`` is a number or a callable of the 0-dim tensor t_i."""
    t = torch.linspace(0, 1, n_timesteps) if tspan is None else tspan.detach().to('cpu', torch.float32).clone()
    alpha, alpha_p = alpha_tables(t, schedule_type, cosine_params)      # (T,4) each; may clamp t[0] in place (cosine)
    if dfm_type not in ('campbell', 'gat'):
        raise ValueError(f"Invalid dfm_type: {dfm_type}")
    out = []
    for s_idx in range(1, t.shape[0]):
        s_i, t_i = t[s_idx], t[s_idx - 1]
        a_i, ap_i = alpha[s_idx - 1], alpha_p[s_idx - 1]      # columns x, a, c, e
        dt = s_i - t_i
        sc = fm_step_scalars()
        sc.t = float(t_i)
        sc.dt = float(dt)
        sc.x_coef = float(ap_i[0] / (1 - a_i[0]))
        mask = torch.clamp(dt * eta, min=0, max=1)
        for k in range(3):
            sc.unmask_prob[k] = float(torch.clamp(dt * (ap_i[k + 1] + eta * a_i[k + 1]) / (1 - a_i[k + 1]), min=0, max=1))
            sc.mask_prob[k] = float(mask)
        sc.hc_thresh = float(torch.tensor(hc_thresh, dtype=torch.float32))
        sc.cat_temperature = _f32(cat_temperature(t_i) if callable(cat_temperature) else cat_temperature)
        sc.last_step = 1 if s_idx == t.shape[0] - 1 else 0
        sc.x_scale = 1.0 if inv_temp_func is None else _f32(inv_temp_func(t_i))
        sc.dfm_type = FM_DFM_GAT if dfm_type == 'gat' else FM_DFM_CAMPBELL
        if dfm_type == 'gat':
            fw = 1.0 if forward_weight_func is None else forward_weight_func(t_i)
            bw = fw - 1                                     # python or tensor arithmetic, as in gat_step
            for k in range(3):
                sc.gat_cf[k] = float(ap_i[k + 1] / (1 - a_i[k + 1]))
                sc.gat_cb[k] = float(ap_i[k + 1] / (a_i[k + 1] + 1e-8))
            sc.gat_fw, sc.gat_bw = _f32(fw), _f32(bw)
        if philox_seed is not None:        # noise drawn inside the CTMC kernel from per-molecule counter-based streams (fm_noise_mode)
            if dfm_type != 'campbell':
                raise NotImplementedError("the in-kernel Philox noise covers dfm_type='campbell'")
            sc.noise_mode = FM_NOISE_PHILOX
            sc.step_index = s_idx - 1
            sc.philox_seed_lo, sc.philox_seed_hi = philox_seed & 0xffffffff, (philox_seed >> 32) & 0xffffffff
        out.append(sc)
    return StepPlan(t, out)


class StepNoise:
    """The nine noise tensors of one CTMC step, in the reference's draw order
    (per modality a, c, e: Exp(1) (rows,K) -> rand(rows) -> rand(rows) [not drawn on the last step])."""
    __slots__ = ('q_a', 'u1_a', 'u2_a', 'q_c', 'u1_c', 'u2_c', 'q_e', 'u1_e', 'u2_e')

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    @staticmethod
    def draw(N: int, U: int, na: int, nc: int, ne: int, last_step: bool, device, generator=None, dfm_type: str = 'campbell') -> "StepNoise":
        """Draw with torch's generator on ``device`` exactly as the reference's ops would
        (torch.multinomial's Exp(1) draw, then the two torch.rand calls; 'gat': one Exp(1) draw over K+1 classes)."""
        out = {}
        if dfm_type == 'gat':
            for tag, rows, k in (('a', N, na), ('c', N, nc), ('e', U, ne)):
                out[f'q_{tag}'] = torch.empty(rows, k + 1, device=device, dtype=torch.float32).exponential_(1, generator=generator)
            return StepNoise(**out)
        for tag, rows, k in (('a', N, na), ('c', N, nc), ('e', U, ne)):
            out[f'q_{tag}'] = torch.empty(rows, k, device=device, dtype=torch.float32).exponential_(1, generator=generator)
            out[f'u1_{tag}'] = torch.rand(rows, device=device, generator=generator)
            out[f'u2_{tag}'] = None if last_step else torch.rand(rows, device=device, generator=generator)
        return StepNoise(**out)

    @staticmethod
    def from_tape(tape: Sequence[torch.Tensor], pos: int, last_step: bool, device, dfm_type: str = 'campbell') -> "tuple[StepNoise, int]":
        out = {}
        if dfm_type == 'gat':
            for tag in 'ace':
                out[f'q_{tag}'] = tape[pos].to(device, torch.float32).contiguous(); pos += 1
            return StepNoise(**out), pos
        for tag in 'ace':
            out[f'q_{tag}'] = tape[pos].to(device, torch.float32).contiguous(); pos += 1
            out[f'u1_{tag}'] = tape[pos].to(device, torch.float32).contiguous(); pos += 1
            if last_step:
                out[f'u2_{tag}'] = None
            else:
                out[f'u2_{tag}'] = tape[pos].to(device, torch.float32).contiguous(); pos += 1
        return StepNoise(**out), pos

    def take_rows(self, node_rows: torch.Tensor, pair_rows: torch.Tensor) -> "StepNoise":
        """The draws of a subset of the batch's nodes / unordered pairs (sharded runs that replicate the full-batch noise)."""
        out = {}
        for k in self.__slots__:
            t = getattr(self, k)
            out[k] = None if t is None else t[pair_rows if k.endswith('_e') else node_rows].contiguous()
        return StepNoise(**out)

    def c_struct(self) -> fm_step_noise:
        s = fm_step_noise()
        for k in self.__slots__:
            setattr(s, k, _ptr(getattr(self, k)))
        return s


class Engine:
    """One ``fm_ctx`` on one device."""

    def __init__(self, cfg: VFConfig, state_dict: Dict[str, torch.Tensor], device='cuda:0', prefix: str = '', lib=None, precision: Optional[str] = None,
                 tuning: Optional[Dict[str, int]] = None):
        """``precision``: 'f32' (default, also for None; the reference's arithmetic), 'bf16x3' (OPT-IN split precision of the edge-message
        GEMMs on the bf16 matrix cores: faster, ~10x larger per-stage error, never used for parity claims) or 'bf16x6' (OPT-IN three-term split of the
        edge-message GEMMs only: f32-class accuracy on the bf16 matrix cores; round 5's measurement of whether that beats the f32 roof).  It is an explicit argument
        only -- no environment variable changes what an Engine computes -- and is recorded in ``self.precision``.
        ``tuning``: launch-tuning overrides of fm_config (ABI 5 / 6: tile_edge, tile_node, tile_edge_update, xcd_swizzle, fuse_node, pair_mlps, pair_slab,
        mlp_small_tiles; 0 / absent = automatic) for A/B measurements and the parity tests that run every tile size."""
        precision = precision or 'f32'
        if precision not in ('f32', 'bf16x3', 'bf16x6', 'f16x3'):
            raise ValueError(f"precision must be 'f32', 'bf16x3', 'bf16x6' or 'f16x3', got {precision!r}")
        self.precision = precision
        self.tuning = {k: int(v) for k, v in (tuning or {}).items() if int(v) != 0}
        unknown = set(self.tuning) - set(_lib.TUNING_FIELDS)
        if unknown:
            raise ValueError(f'unknown tuning fields {sorted(unknown)}; have {_lib.TUNING_FIELDS}')
        cfg.validate()
        self.cfg = cfg
        self.device = torch.device(device)
        self.lib = lib if lib is not None else _lib.get_lib()
        check_state_dict(cfg, state_dict, prefix)
        shapes = state_dict_shapes(cfg)
        # ---- flat host blob + descriptors, keys named as in the reference state dict
        descs = (fm_tensor_desc * len(shapes))()
        chunks, off = [], 0
        self._names = []
        for i, (key, shape) in enumerate(shapes.items()):
            t = state_dict[prefix + key].detach().to('cpu', torch.float32).contiguous().reshape(-1)
            chunks.append(t)
            nm = key.encode()
            self._names.append(nm)
            descs[i].name = nm
            descs[i].offset = off
            descs[i].ndim = len(shape)
            descs[i].shape[0] = shape[0]
            descs[i].shape[1] = shape[1] if len(shape) > 1 else 0
            off += t.numel()
        blob = torch.cat(chunks) if chunks else torch.zeros(0)
        c = fm_config()
        c.abi_version = _lib.FM_ABI_VERSION
        c.n_atom_types, c.n_charges, c.n_bond_types = cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types
        c.n_vec_channels, c.n_hidden_scalars, c.n_hidden_edge_feats = cfg.n_vec_channels, cfg.n_hidden_scalars, cfg.n_hidden_edge_feats
        c.rbf_dim, c.n_convs, c.n_updaters = cfg.rbf_dim, cfg.n_convs, cfg.n_updaters
        sched = cfg.update_schedule()
        if len(sched) > _lib.FM_MAX_CONVS:
            raise NotImplementedError(f'more than {_lib.FM_MAX_CONVS} convolutions')
        for i in range(_lib.FM_MAX_CONVS):
            c.update_after[i] = sched[i] if i < len(sched) else -1
        c.self_conditioning = int(cfg.self_conditioning)
        c.time_embedding_dim = cfg.time_embedding_dim
        c.a_token_dim, c.c_token_dim, c.e_token_dim = cfg.a_token_dim, cfg.c_token_dim, cfg.e_token_dim
        c.rbf_dmax = float(cfg.rbf_dmax)
        c.msg_z = float(cfg.msg_z)
        c.s_dst_feats, c.v_dst_feats = cfg.s_dst_feats, cfg.v_dst_feats
        c.has_mask = int(cfg.has_mask)
        c.precision = {'f32': _lib.FM_PREC_F32, 'bf16x3': _lib.FM_PREC_BF16X3, 'bf16x6': _lib.FM_PREC_BF16X6, 'f16x3': _lib.FM_PREC_F16X3}[precision]
        c.n_recycles = int(cfg.n_recycles)
        c.edge_update_no_distance = int(not cfg.update_edge_w_distance)
        for k, v in self.tuning.items():
            setattr(c, k, v)
        self._ctx = C.c_void_p()
        with self._dev():
            rc = self.lib.fm_create(C.byref(c), descs, len(shapes), C.c_void_p(blob.data_ptr()), C.byref(self._ctx))
        if rc != 0:
            raise FlowMolHipError(f'fm_create failed ({rc}): {self.lib.fm_last_error(None).decode()}')
        self._keep = []          # tensors referenced by raw pointer inside the library
        self.N = self.E = self.U = self.B = 0
        self._ws = None
        self.n_atoms = None

    # ------------------------------------------------------------------ plumbing
    def _dev(self):
        if self.device.type == 'cuda':
            return torch.cuda.device(self.device)
        import contextlib
        return contextlib.nullcontext()

    def _stream(self):
        if self.device.type == 'cuda':
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def _check(self, rc, what):
        if rc != 0:
            raise FlowMolHipError(f'{what} failed ({rc}): {self.lib.fm_last_error(self._ctx).decode()}')

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            if self.device.type == 'cuda':
                torch.cuda.synchronize(self.device)
            self.lib.fm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ batch
    def bind(self, n_atoms: torch.Tensor):
        """Bind a batch of molecules given their sizes (order preserved).  Replaces the reference's
        dgl.graph / dgl.batch / upper-edge-mask / batch-index construction (flowmol.py:509-529)."""
        n = n_atoms.detach().to('cpu', torch.int32).contiguous()
        if n.dim() != 1 or n.numel() == 0:
            raise ValueError('n_atoms must be a non-empty 1-D tensor')
        need = C.c_size_t()
        self._check(self.lib.fm_workspace_bytes(self._ctx, C.c_void_p(n.data_ptr()), n.numel(), C.byref(need)), 'fm_workspace_bytes')
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            self._ws = torch.empty(need.value + 256, dtype=torch.uint8, device=self.device)
        base = self._ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        with self._dev():
            self._check(self.lib.fm_batch_bind(self._ctx, self._stream(), C.c_void_p(n.data_ptr()), n.numel(),
                                               C.c_void_p(aligned), need.value), 'fm_batch_bind')
        n64 = n.to(torch.int64)
        self.n_atoms = n64
        self.B = int(n.numel())
        self.N = int(n64.sum())
        self.E = int((n64 * (n64 - 1)).sum())
        self.U = self.E // 2
        self.workspace_bytes = int(need.value)
        return self

    def query(self, name: str) -> torch.Tensor:
        cnt = {'e_src': self.E, 'e_dst': self.E, 'e_pair': self.E, 'p_e0': self.U, 'p_e1': self.U,
               'node_mol': self.N, 'pair_mol': self.U}[name]
        out = torch.empty(cnt, dtype=torch.int32, device=self.device)
        with self._dev():
            self._check(self.lib.fm_batch_query(self._ctx, self._stream(), name.encode(), _ptr(out)), 'fm_batch_query')
        self.synchronize()
        return out

    # ------------------------------------------------------------------ tensors
    def new_dst(self) -> Dict[str, torch.Tensor]:
        d = self.device
        return {'x': torch.empty(self.N, 3, device=d), 'a': torch.empty(self.N, self.cfg.n_atom_types, device=d),
                'c': torch.empty(self.N, self.cfg.n_charges, device=d), 'e': torch.empty(self.U, self.cfg.n_bond_types, device=d)}

    @staticmethod
    def _dst_struct(d) -> fm_dst:
        s = fm_dst()
        s.x, s.a, s.c, s.e = _ptr(d['x']), _ptr(d['a']), _ptr(d['c']), _ptr(d['e'])
        return s

    @staticmethod
    def _state_struct(st) -> fm_state:
        s = fm_state()
        s.x_t, s.a_t, s.c_t, s.e_t = _ptr(st['x_t']), _ptr(st['a_t']), _ptr(st['c_t']), _ptr(st['e_t'])
        return s

    def make_state(self, x_t, a_t, c_t, e_t) -> Dict[str, torch.Tensor]:
        """x_t (N,3) float; a_t, c_t (N) and e_t (U, per unordered pair in reference upper-edge order) int tokens."""
        d = self.device
        st = {'x_t': x_t.detach().to(d, torch.float32).contiguous().clone(),
              'a_t': a_t.detach().to(d, torch.int32).contiguous().clone(),
              'c_t': c_t.detach().to(d, torch.int32).contiguous().clone(),
              'e_t': e_t.detach().to(d, torch.int32).contiguous().clone()}
        assert st['x_t'].shape == (self.N, 3) and st['a_t'].shape == (self.N,) and st['e_t'].shape == (self.U,)
        return st

    def prior_state(self, x_0: torch.Tensor) -> Dict[str, torch.Tensor]:
        """CTMC masked prior for a, c, e (priors.py:101-107,305-316) around given positions."""
        d = self.device
        return self.make_state(x_0, torch.full((self.N,), self.cfg.n_atom_types, dtype=torch.int32),
                               torch.full((self.N,), self.cfg.n_charges, dtype=torch.int32),
                               torch.full((self.U,), self.cfg.n_bond_types, dtype=torch.int32))

    def remove_com(self, x: torch.Tensor) -> torch.Tensor:
        """In-place per-molecule centring of an (N,3) tensor on the engine's device."""
        assert x.is_contiguous() and x.dtype == torch.float32 and x.shape == (self.N, 3)
        with self._dev():
            self._check(self.lib.fm_remove_com(self._ctx, self._stream(), _ptr(x)), 'fm_remove_com')
        return x

    def set_molecule_ids(self, ids: Optional[torch.Tensor]):
        """Global ids of the bound batch's molecules for the Philox noise streams (None = 0..B-1): a molecule keeps its id, and
        therefore its draws, however the batch is sharded."""
        if ids is None:
            ptr = None
        else:
            ids = ids.detach().to('cpu', torch.int32).contiguous()
            assert ids.shape == (self.B,)
            ptr = C.c_void_p(ids.data_ptr())
        with self._dev():
            self._check(self.lib.fm_set_molecule_ids(self._ctx, self._stream(), ptr), 'fm_set_molecule_ids')

    def prior_philox(self, seed: int) -> torch.Tensor:
        """Centred Gaussian position prior (priors.py:27-35) from the per-molecule Philox streams."""
        x0 = torch.empty(self.N, 3, device=self.device)
        with self._dev():
            self._check(self.lib.fm_prior_philox(self._ctx, self._stream(), C.c_uint64(seed), _ptr(x0)), 'fm_prior_philox')
        return x0

    def stability(self, state, table: torch.Tensor, fake_atom_token: int = -1, explicit_aromaticity: bool = False) -> torch.Tensor:
        """Valence stability + connectivity of the bound batch's molecules from their tokens, on the device
        (see fm_stability).  table: (n_types, n_charges) int32 bit masks.  Returns (B,4) int32:
        stable atoms, real atoms, connected components, largest component."""
        tab = table.detach().to(self.device, torch.int32).contiguous()
        assert tab.dim() == 2 and tab.shape[1] == self.cfg.n_charges
        out = torch.empty(self.B, 4, dtype=torch.int32, device=self.device)
        st = self._state_struct(state)
        with self._dev():
            self._check(self.lib.fm_stability(self._ctx, self._stream(), C.byref(st), _ptr(tab), int(tab.shape[0]),
                                              int(fake_atom_token), int(bool(explicit_aromaticity)), _ptr(out)), 'fm_stability')
        return out

    # ------------------------------------------------------------------ hot path
    def forward(self, state, t: float, prev=None, bootstrap=False, remove_com=True, out=None, taps: Optional[Dict[str, torch.Tensor]] = None):
        """One network evaluation -> dst dict of probabilities (EndpointVectorField.forward with
        apply_softmax=True)."""
        out = out if out is not None else self.new_dst()
        temb = time_embedding_host(float(t), self.cfg.time_embedding_dim).to(self.device)
        st = self._state_struct(state)
        o = self._dst_struct(out)
        p = self._dst_struct(prev) if prev is not None else None
        self.lib.fm_clear_taps(self._ctx)
        if taps:
            for k, v in taps.items():
                self._check(self.lib.fm_set_tap(self._ctx, k.encode(), _ptr(v)), 'fm_set_tap')
        with self._dev():
            rc = self.lib.fm_forward(self._ctx, self._stream(), C.byref(st), _ptr(temb), C.byref(p) if p is not None else None,
                                     int(bool(bootstrap)), int(bool(remove_com)), C.byref(o))
        self._check(rc, 'fm_forward')
        if taps:
            self.lib.fm_clear_taps(self._ctx)
        self._keep = [temb, state, out, prev]
        return out

    # ------------------------------------------------------------------ endpoint parameterization (EndpointVectorField)
    @staticmethod
    def _dense_struct(st) -> fm_dense_state:
        s = fm_dense_state()
        s.x_t, s.a_t, s.c_t, s.e_t = _ptr(st['x_t']), _ptr(st['a_t']), _ptr(st['c_t']), _ptr(st['e_t'])
        return s

    def make_dense_state(self, x_t, a_t, c_t, e_t) -> Dict[str, torch.Tensor]:
        """x_t (N,3), a_t (N,na), c_t (N,nc), e_t (U,ne): continuous features of an endpoint-parameterised model (copies)."""
        d = self.device
        st = {k: v.detach().to(d, torch.float32).contiguous().clone() for k, v in (('x_t', x_t), ('a_t', a_t), ('c_t', c_t), ('e_t', e_t))}
        assert st['x_t'].shape == (self.N, 3) and st['a_t'].shape == (self.N, self.cfg.n_atom_types)
        assert st['c_t'].shape == (self.N, self.cfg.n_charges) and st['e_t'].shape == (self.U, self.cfg.n_bond_types)
        return st

    def forward_dense(self, state, t: float, remove_com=True, out=None, taps: Optional[Dict[str, torch.Tensor]] = None):
        """EndpointVectorField.forward on continuous categorical features (vector_field.py:212-293, no self-conditioning)."""
        out = out if out is not None else self.new_dst()
        temb = time_embedding_host(float(t), self.cfg.time_embedding_dim).to(self.device)
        st, o = self._dense_struct(state), self._dst_struct(out)
        self.lib.fm_clear_taps(self._ctx)
        if taps:
            for k, v in taps.items():
                self._check(self.lib.fm_set_tap(self._ctx, k.encode(), _ptr(v)), 'fm_set_tap')
        with self._dev():
            rc = self.lib.fm_forward_dense(self._ctx, self._stream(), C.byref(st), _ptr(temb), int(bool(remove_com)), C.byref(o))
        self._check(rc, 'fm_forward_dense')
        if taps:
            self.lib.fm_clear_taps(self._ctx)
        self._keep = [temb, state, out]
        return out

    def endpoint_step(self, state, dst, dt: float, coef, scale: float = 1.0):
        """x_s = x_t + ((coef (x_1 - x_t)) scale) dt for x, a, c, e in place (EndpointVectorField.step, vector_field.py:528-543)."""
        sc = fm_endpoint_scalars()
        sc.dt, sc.scale = float(dt), float(scale)
        for k in range(4):
            sc.coef[k] = float(coef[k])
        st, d = self._dense_struct(state), self._dst_struct(dst)
        with self._dev():
            rc = self.lib.fm_endpoint_step(self._ctx, self._stream(), C.byref(st), C.byref(d), C.byref(sc))
        self._check(rc, 'fm_endpoint_step')
        return state

    def integrate_endpoint(self, state, n_timesteps: int, inv_temp_func=None, tspan=None, traj: Optional[Dict[str, torch.Tensor]] = None):
        """EndpointVectorField.integrate (vector_field.py:388-499): Euler steps of all four modalities; the per-step scalars are computed
        with the reference's float32 tensor arithmetic on the host, the steps are enqueued back to back (no host synchronisation).
        ``traj`` (optional, vector_field.py:412-466 `visualize`): preallocated per-step frames -- 'x' (steps,N,3) and the argmax categories
        'a','c' (steps,N), 'e' (steps,U) of the state after each step, 'x1','a1','c1','e1' of the step's endpoint prediction."""
        cfg = self.cfg
        t = torch.linspace(0, 1, n_timesteps) if tspan is None else tspan.detach().to('cpu', torch.float32).clone()
        alpha, alpha_p = alpha_tables(t, cfg.schedule_type, cfg.cosine_params)
        if inv_temp_func is None:
            if cfg.continuous_inv_temp_schedule == 'linear':
                mx = cfg.continuous_inv_temp_max
                inv_temp_func = lambda tt: mx * (1 - tt)                     # vector_field.py:203-204
            else:
                inv_temp_func = lambda tt: 1.0
        dst = [self.new_dst(), self.new_dst()]
        for s_idx in range(1, t.shape[0]):
            s_i, t_i = t[s_idx], t[s_idx - 1]
            a_i, ap_i = alpha[s_idx - 1], alpha_p[s_idx - 1]
            out = dst[s_idx & 1]
            self.forward_dense(state, float(t_i), remove_com=True, out=out)
            self.endpoint_step(state, out, float(s_i - t_i), [float(ap_i[k] / (1 - a_i[k])) for k in range(4)], _f32(inv_temp_func(t_i)))
            if traj is not None:
                f = s_idx - 1
                traj['x'][f].copy_(state['x_t']); traj['x1'][f].copy_(out['x'])
                for k in 'ace':
                    traj[k][f].copy_(state[f'{k}_t'].argmax(-1)); traj[f'{k}1'][f].copy_(out[k].argmax(-1))
        self.synchronize()
        return dst[(t.shape[0] - 1) & 1] if t.shape[0] > 1 else None

    def ctmc_step(self, state, dst, noise: StepNoise, sc: fm_step_scalars, sampled: Optional[Dict[str, torch.Tensor]] = None):
        st = self._state_struct(state)
        d = self._dst_struct(dst)
        nz = noise.c_struct()
        smp = fm_sampled()
        if sampled is not None:
            smp.a1, smp.c1, smp.e1 = _ptr(sampled['a1']), _ptr(sampled['c1']), _ptr(sampled['e1'])
        with self._dev():
            rc = self.lib.fm_ctmc_step(self._ctx, self._stream(), C.byref(st), C.byref(d), C.byref(nz), C.byref(sc), C.byref(smp))
        self._check(rc, 'fm_ctmc_step')
        self._keep = [state, dst, noise, sampled]
        return state

    def integrate(self, state, plan: StepPlan, noise_for_step, chunk: int = 32, traj: Optional[Dict[str, torch.Tensor]] = None):
        """Run all steps of ``plan`` (CTMCVectorField.integrate).  ``noise_for_step(i, last)`` returns the
        StepNoise of step i; noise is produced ``chunk`` steps ahead and each chunk is one fm_integrate
        call (no host synchronisation in between).  Returns the final endpoint prediction dict."""
        run = IntegrationRun(self, state, plan, noise_for_step, traj=traj)
        run.run(0, len(plan.scalars), chunk=chunk)
        self.synchronize()
        return run.last_dst()

    # ------------------------------------------------------------------ profiling
    def profile(self, on: bool):
        self._check(self.lib.fm_profile_enable(self._ctx, int(on)), 'fm_profile_enable')

    def profile_get(self, kernel: str):
        ms, n = C.c_double(), C.c_int64()
        self._check(self.lib.fm_profile_get(self._ctx, kernel.encode(), C.byref(ms), C.byref(n)), 'fm_profile_get')
        return ms.value, n.value


class IntegrationRun:
    """A trajectory in progress: owns the two endpoint buffers and remembers which one holds the previous
    step's prediction, so a trajectory can be advanced in several fm_integrate calls (bench.py times a
    window of steps; Engine.integrate runs them all)."""

    def __init__(self, eng: Engine, state, plan: StepPlan, noise_for_step, traj=None):
        self.eng, self.state, self.plan, self.noise_for_step, self.traj = eng, state, plan, noise_for_step, traj
        tt = eng.cfg.time_embedding_dim
        self.temb_all = (torch.stack([time_embedding_host(sc.t, tt) for sc in plan.scalars]) if plan.scalars
                         else torch.zeros(0, tt)).to(eng.device).contiguous()      # n_timesteps = 1: no step, the prior is the result
        self.dst = [eng.new_dst(), eng.new_dst()]
        self._dsts = [eng._dst_struct(self.dst[0]), eng._dst_struct(self.dst[1])]
        self._st = eng._state_struct(state)
        self.prev_idx = None
        self._keep = []

    def reset(self, state):
        self.state = state
        self._st = self.eng._state_struct(state)
        self.prev_idx = None

    def last_dst(self):
        return None if self.prev_idx is None else self.dst[self.prev_idx]

    NOISE_BYTES_PER_CHUNK = 256 << 20     # torch-generated noise kept alive per fm_integrate call (up to 3 chunks are in flight)

    def run(self, lo: int, hi: int, chunk: int = 32):
        eng = self.eng
        final = C.c_int(0)
        if self.noise_for_step is not None:
            # the reference's draws are tensors: (N, na) + (N, nc) + (U, ne) Exp(1) and two uniforms per row and step.  Bound the
            # transient footprint (30 MB per step at 1024 x 47 atoms) instead of letting it grow with the batch; the in-kernel
            # Philox mode has no noise tensors at all.
            cfg = eng.cfg
            per_step = 4 * (eng.N * (cfg.n_atom_types + cfg.n_charges + 4) + eng.U * (cfg.n_bond_types + 2))
            chunk = max(1, min(chunk, self.NOISE_BYTES_PER_CHUNK // max(per_step, 1)))
        for a in range(lo, hi, chunk):
            b = min(hi, a + chunk)
            k = b - a
            scal = (fm_step_scalars * k)(*self.plan.scalars[a:b])
            if self.noise_for_step is None:        # Philox plan: the kernel draws its own noise
                noises, nzs = None, None
            else:
                noises = [self.noise_for_step(i, bool(self.plan.scalars[i].last_step)) for i in range(a, b)]
                nzs = (fm_step_noise * k)(*[nz.c_struct() for nz in noises])
            sink = None
            if self.traj is not None:
                sink = fm_traj_sink()
                for name in ('x', 'a', 'c', 'e', 'x1', 'a1', 'c1', 'e1'):
                    tsr = self.traj.get(name)
                    setattr(sink, name, _ptr(tsr[a:]) if tsr is not None else None)
            with eng._dev():
                rc = eng.lib.fm_integrate(eng._ctx, eng._stream(), C.byref(self._st), k, scal, _ptr(self.temb_all[a:]), nzs,
                                          C.byref(self._dsts[self.prev_idx]) if self.prev_idx is not None else None,
                                          C.byref(self._dsts[0]), C.byref(self._dsts[1]),
                                          C.byref(sink) if sink is not None else None, C.byref(final))
            eng._check(rc, 'fm_integrate')
            self.prev_idx = final.value
            # bound the noise kept alive: a chunk's tensors are released once the GPU has passed the event recorded behind it.  The host waits for
            # the chunk BEFORE the previous one only, so two chunks stay enqueued and the device never idles (a device-wide synchronise here
            # drained the queue every third chunk)
            ev = None
            if eng.device.type == 'cuda':
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(eng.device))
            self._keep.append((noises, scal, nzs, ev))
            while len(self._keep) > 2:
                old = self._keep.pop(0)
                if old[3] is not None:
                    old[3].synchronize()
