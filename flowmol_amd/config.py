"""Model configuration for the FlowMol3 sampling hot path.

The reference splats the YAML ``vector_field:`` block into
``CTMCVectorField.__init__`` (reference flowmol/models/flowmol.py:146-153) and
the ``mol_fm:`` block into ``FlowMol.__init__`` (flowmol/model_utils/load.py:43-47).
``VFConfig`` is the flattened, validated form of the knobs that shape the
sampling path; it is what gets serialised into the C-ABI ``fm_config`` struct
(include/flowmol_hip.h).

Only the configurations the HIP kernels implement are accepted; anything else
raises ``NotImplementedError`` loudly (no silent fallback).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List, Optional, Union


@dataclass
class VFConfig:
    # ---- categorical sizes (reference flowmol.py:57-80, vector_field.py:93-97)
    atom_type_map: List[str] = field(default_factory=lambda: ['C', 'H', 'N', 'O', 'F', 'P', 'S', 'Cl', 'Br', 'I'])
    fake_atoms: bool = True            # fake_atom_p > 0  -> one extra atom type (flowmol.py:76-80)
    n_charges: int = 6
    n_bond_types: int = 4              # 5 with explicit aromaticity (flowmol.py:59)
    # ---- network dims (configs/flowmol3.yml:80-106)
    n_vec_channels: int = 32
    n_cp_feats: int = 4
    n_hidden_scalars: int = 256
    n_hidden_edge_feats: int = 128
    n_molecule_updates: int = 6
    convs_per_update: int = 1
    n_message_gvps: int = 3
    n_update_gvps: int = 3
    separate_mol_updaters: bool = True
    message_norm: Union[float, str] = 'sum'
    update_edge_w_distance: bool = True
    rbf_dmax: float = 10.0
    rbf_dim: int = 32
    time_embedding_dim: int = 64
    a_token_dim: int = 64
    c_token_dim: int = 64
    e_token_dim: int = 64
    self_conditioning: bool = True
    use_dst_feats: bool = False                         # GVPConv: reduced destination-node features join the message inputs (gvp.py:300-316,527-537)
    dst_feat_msg_reduction_factor: float = 4
    n_recycles: int = 1
    # ---- flow-matching parameterization (flowmol.py:40,137-153): 'ctmc' = CTMCVectorField (categorical tokens with a mask state),
    # 'endpoint' = EndpointVectorField (categoricals as continuous simplex-like vectors, Euler step for all four modalities)
    parameterization: str = 'ctmc'
    prior_types: dict = field(default_factory=lambda: {'a': 'ctmc', 'c': 'ctmc', 'e': 'ctmc'})     # prior_config[feat]['type'] (flowmol.py:189-193,417-448)
    prior_kwargs: dict = field(default_factory=lambda: {'a': {}, 'c': {}, 'e': {}})
    continuous_inv_temp_schedule: Optional[str] = None     # EndpointVectorField.step scaling of the vector field: None | 'linear' (vector_field.py:199-209)
    continuous_inv_temp_max: float = 10.0
    # ---- CTMC integrator defaults (ctmc_vector_field.py:23-34)
    stochasticity: float = 30.0
    high_confidence_threshold: float = 0.9
    cat_temperature: float = 0.05                       # constant schedule value (cat_temperature_schedule given as a number)
    cat_temperature_schedule: Union[float, str] = 0.05  # number, or 'decay': max*(1-t)^a (ctmc_vector_field.py:71-82)
    cat_temp_decay_max: float = 0.8
    cat_temp_decay_a: float = 2
    dfm_type: str = 'campbell'                          # 'campbell' | 'gat' (ctmc_vector_field.py:357,373)
    forward_weight_schedule: Union[float, str] = 'beta' # 'gat' only: number, or 'beta': 1 + max*t^a*(1-t)^b (:84-95)
    fw_beta_a: float = 0.25
    fw_beta_b: float = 0.25
    fw_beta_max: float = 10.0
    # ---- interpolant schedule per modality (interpolant_scheduler.py:9-58): 'linear' | 'cosine' (+ exponent nu per cosine modality)
    schedule_type: dict = field(default_factory=lambda: {k: 'linear' for k in 'xace'})
    cosine_params: dict = field(default_factory=dict)
    # ---- sampling defaults (flowmol.py:46)
    default_n_timesteps: int = 250
    # name of the size histogram shipped in flowmol_amd/data/n_atoms_hist.json
    n_atoms_hist: str = 'geom_full_kekulized'
    explicit_aromaticity: bool = False

    # ------------------------------------------------------------------ derived
    @property
    def n_atom_types(self) -> int:
        """Number of real categories for 'a' (incl. the fake-atom type, excl. mask)."""
        return len(self.atom_type_map) + (1 if self.fake_atoms else 0)

    @property
    def n_convs(self) -> int:
        return self.convs_per_update * self.n_molecule_updates

    @property
    def n_updaters(self) -> int:
        return self.n_molecule_updates if self.separate_mol_updaters else 1

    @property
    def token_dims(self):
        """Width of the token features fed to the embedding MLPs.

        token_dim == 0 -> raw one-hot incl. mask column (vector_field.py:102-119)."""
        m = 1 if self.has_mask else 0
        a = self.a_token_dim if self.a_token_dim else self.n_atom_types + m
        c = self.c_token_dim if self.c_token_dim else self.n_charges + m
        e = self.e_token_dim if self.e_token_dim else self.n_bond_types + m
        return a, c, e

    @property
    def has_mask(self) -> bool:
        """CTMC models carry a mask category in their categorical inputs (vector_field.py:100, ctmc_vector_field.py:36)."""
        return self.parameterization == 'ctmc'

    @property
    def s_dst_feats(self) -> int:
        """Scalars of the reduced destination-node message features (gvp.py:302)."""
        return int(self.n_hidden_scalars / self.dst_feat_msg_reduction_factor) if self.use_dst_feats else 0

    @property
    def v_dst_feats(self) -> int:
        return int(self.n_vec_channels / self.dst_feat_msg_reduction_factor) if self.use_dst_feats else 0

    @property
    def msg_z(self) -> float:
        """Divisor applied to the aggregated messages (gvp.py:495-501); -1.0 stands for message_norm='mean' (DGL fn.mean over the
        in-edges, gvp.py:401-404), which the library expresses as 'divide by the node's in-degree'."""
        if isinstance(self.message_norm, str):
            if self.message_norm == 'sum':
                return 1.0
            if self.message_norm == 'mean':
                return -1.0
            raise ValueError(f"message_norm must be either 'mean', 'sum', or a number, got {self.message_norm}")      # gvp.py:395-396
        if float(self.message_norm) <= 0:
            raise ValueError(f"message_norm must be positive, got {self.message_norm}")
        return float(self.message_norm)

    def update_schedule(self):
        """For each conv index, the updater index run after it or -1.

        Reproduces vector_field.py:320-326: an update follows conv ``i`` iff
        ``i != 0 and (i+1) % convs_per_update == 0``; with separate updaters the
        index is ``i // convs_per_update`` (so updater 0 is dead when
        convs_per_update == 1)."""
        out = []
        for i in range(self.n_convs):
            if i != 0 and (i + 1) % self.convs_per_update == 0:
                out.append(i // self.convs_per_update if self.separate_mol_updaters else 0)
            else:
                out.append(-1)
        return out

    def validate(self) -> "VFConfig":
        # tiles are 256 scalar / 128 edge-feature columns wide; narrower models (configs/dev.yml: 64 / 64) are zero-padded at fm_create
        # ... and the LayerNorm statistics take the mean as sum * (1/n), which equals sum / n bit for bit only for a power-of-two n
        # (fm_device.h: fm_row_stats); the shipped widths (256 / 128, dev.yml 64 / 64) are the tested ones
        pow2 = lambda v: v >= 8 and (v & (v - 1)) == 0
        if not (pow2(self.n_hidden_scalars) and self.n_hidden_scalars <= 256 and pow2(self.n_hidden_edge_feats) and self.n_hidden_edge_feats <= 128) \
                or self.rbf_dim != 32:
            raise NotImplementedError(
                f"HIP kernels hold power-of-two widths up to S=256 scalars and F=128 edge features (R=32); got S={self.n_hidden_scalars} "
                f"F={self.n_hidden_edge_feats} R={self.rbf_dim}")
        if self.n_vec_channels not in (16, 32):
            raise NotImplementedError(f"n_vec_channels must be 16 or 32, got {self.n_vec_channels}")
        if self.n_cp_feats != 4 or self.n_message_gvps != 3 or self.n_update_gvps != 3:
            raise NotImplementedError("only n_cp_feats=4 and 3/3 message/update GVPs are implemented")
        if self.use_dst_feats:
            if self.dst_feat_msg_reduction_factor == 1:
                raise NotImplementedError("use_dst_feats with dst_feat_msg_reduction_factor == 1 (no projection GVP) is not implemented")
            if self.v_dst_feats != self.n_vec_channels // 4 or not 1 <= self.s_dst_feats <= 256:
                raise NotImplementedError(f"use_dst_feats is instantiated for dst_feat_msg_reduction_factor = 4 (v_dst_feats = n_vec_channels / 4, as in "
                                          f"configs/dev.yml); got v={self.v_dst_feats} s={self.s_dst_feats} for {self.n_vec_channels} vector channels")
        if not 1 <= int(self.n_recycles) <= 64:
            raise ValueError(f"n_recycles must be 1..64, got {self.n_recycles}")
        if self.dfm_type not in ('campbell', 'gat'):
            raise ValueError(f"Invalid dfm_type: {self.dfm_type}")           # ctmc_vector_field.py:62-63
        if not isinstance(self.cat_temperature_schedule, (int, float)) and self.cat_temperature_schedule != 'decay':
            raise ValueError(f"Invalid cat_temperature_schedule: {self.cat_temperature_schedule}")
        if not isinstance(self.forward_weight_schedule, (int, float)) and self.forward_weight_schedule != 'beta':
            raise ValueError(f"Invalid forward_weight_schedule: {self.forward_weight_schedule}")
        tok = (self.a_token_dim > 0, self.c_token_dim > 0, self.e_token_dim > 0)
        if len(set(tok)) != 1:
            raise NotImplementedError("token dims must be all zero or all non-zero")
        if self.n_atom_types + 1 > 16 or self.n_charges + 1 > 8 or self.n_bond_types + 1 > 8:
            raise NotImplementedError("categorical widths exceed kernel limits (a<=16, c<=8, e<=8 incl. mask)")
        self.msg_z  # raises for an unknown message_norm
        if self.parameterization not in ('ctmc', 'endpoint'):
            raise NotImplementedError(f"parameterization {self.parameterization!r}: 'ctmc' and 'endpoint' are implemented "
                                      "(the deprecated 'vector-field' / 'dirichlet' families are out of scope)")
        if self.parameterization == 'endpoint':
            if self.a_token_dim or self.c_token_dim or self.e_token_dim:
                raise ValueError('token dims must be 0 for a non-CTMC parameterization (configs/dev.yml:105)')
            if self.self_conditioning:
                raise NotImplementedError('self-conditioning with the endpoint parameterization is not implemented (no such model ships)')
            for k in 'ace':
                if self.prior_types.get(k) not in ('gaussian', 'uniform-simplex', 'barycenter', 'biased-simplex', 'marginal', 'c-given-a'):
                    raise NotImplementedError(f"prior type {self.prior_types.get(k)!r} for {k!r}: implemented for endpoint models: gaussian, "
                                              "uniform-simplex, barycenter, biased-simplex, marginal, c-given-a")
            if self.continuous_inv_temp_schedule not in (None, 'linear'):
                raise ValueError(f'Invalid continuous_inv_temp_schedule: {self.continuous_inv_temp_schedule}')
        for k in 'xace':
            st = self.schedule_type.get(k)
            if st not in ('linear', 'cosine'):
                raise ValueError(f'unsupported schedule_type for {k!r}: {st!r}')           # interpolant_scheduler.py:21-22
            if st == 'cosine' and k not in self.cosine_params:
                raise ValueError(f'must specify cosine_params for feature {k}')          # interpolant_scheduler.py:45-47
        return self

    def to_dict(self):
        return asdict(self)


def from_reference_hparams(hp: dict) -> VFConfig:
    """Build a VFConfig from a Lightning checkpoint's ``hyper_parameters`` dict
    (the kwargs of reference ``FlowMol.__init__``, flowmol.py:29-55,169)."""
    vf = dict(hp.get('vector_field_config', {}))
    cfg = VFConfig(
        atom_type_map=list(hp['atom_type_map']),
        fake_atoms=float(hp.get('fake_atom_p', 0.0)) > 0,
        n_charges=int(hp.get('n_atom_charges', 6)),
        n_bond_types=5 if hp.get('explicit_aromaticity', False) else 4,
        explicit_aromaticity=bool(hp.get('explicit_aromaticity', False)),
        default_n_timesteps=int(hp.get('default_n_timesteps', 250)),
    )
    # defaults of EndpointVectorField/CTMCVectorField.__init__ for keys absent from the YAML
    ref_defaults = dict(
        n_vec_channels=16, n_cp_feats=0, n_hidden_scalars=64, n_hidden_edge_feats=64,
        n_molecule_updates=2, convs_per_update=2, n_message_gvps=3, n_update_gvps=3,
        separate_mol_updaters=False, message_norm=100, update_edge_w_distance=False,
        rbf_dmax=20, rbf_dim=16, time_embedding_dim=1, a_token_dim=0, c_token_dim=0,
        e_token_dim=0, self_conditioning=False, use_dst_feats=False, dst_feat_msg_reduction_factor=4, n_recycles=1,
        stochasticity=0.0, high_confidence_threshold=0.0, dfm_type='campbell',
    )
    for k, dflt in ref_defaults.items():
        setattr(cfg, k, vf.get(k, dflt))
    ct = vf.get('cat_temperature_schedule', 0.05)
    cfg.cat_temperature_schedule = ct
    if isinstance(ct, (int, float)):
        cfg.cat_temperature = float(ct)
    for k, dflt in (('cat_temp_decay_max', 0.8), ('cat_temp_decay_a', 2), ('forward_weight_schedule', 'beta'),
                    ('fw_beta_a', 0.25), ('fw_beta_b', 0.25), ('fw_beta_max', 10.0)):
        setattr(cfg, k, vf.get(k, dflt))
    # keys of the reference's vector_field: block this loader understands; anything else would be silently ignored, so it is an error
    known = set(ref_defaults) | {'cat_temperature_schedule', 'cat_temp_decay_max', 'cat_temp_decay_a', 'forward_weight_schedule', 'fw_beta_a',
                                 'fw_beta_b', 'fw_beta_max', 'attention', 'dropout', 's_message_dim', 'v_message_dim', 'n_heads', 'n_expansion_gvps',
                                 'continuous_inv_temp_schedule', 'continuous_inv_temp_max', 'dst_feat_msg_reduction_factor', 'scprop', 'has_mask',
                                 'exclude_charges', 'n_cp_feats'}
    unknown = sorted(set(vf) - known)
    if unknown:
        raise NotImplementedError(f'vector_field keys {unknown} are not understood by this loader')
    if vf.get('exclude_charges', False):
        raise NotImplementedError('exclude_charges is deprecated in the reference (vector_field.py:80-82) and not implemented')
    # n_heads / n_expansion_gvps only take effect with attention / compressed messaging (rejected below); flowmol3.yml sets n_heads: 32 with attention: False
    for unsupported in ('attention', 'dropout', 's_message_dim', 'v_message_dim'):
        v = vf.get(unsupported)
        if v not in (None, False, 0, 0.0):
            raise NotImplementedError(f"vector_field.{unsupported}={v!r} is not implemented")
    cfg.parameterization = hp.get('parameterization', 'endpoint')           # flowmol.py:40 default
    pc = hp.get('prior_config', {}) or {}
    if cfg.parameterization != 'ctmc':
        cfg.prior_types = {k: (pc.get(k, {}) or {}).get('type') for k in 'ace'}
        cfg.prior_kwargs = {k: dict((pc.get(k, {}) or {}).get('kwargs', {}) or {}) for k in 'ace'}
        cfg.continuous_inv_temp_schedule = vf.get('continuous_inv_temp_schedule')
        cfg.continuous_inv_temp_max = float(vf.get('continuous_inv_temp_max', 10.0))
    isc = hp.get('interpolant_scheduler_config', {}) or {}
    st = isc.get('schedule_type', 'cosine')                  # InterpolantScheduler's own default (interpolant_scheduler.py:9)
    cfg.schedule_type = {k: st for k in 'xace'} if isinstance(st, str) else {k: st[k] for k in 'xace'}
    cfg.cosine_params = {k: float(v) for k, v in (isc.get('cosine_params', {}) or {}).items()}
    nah = str(hp.get('n_atoms_hist_file', ''))
    for name in ('geom_full_kekulized', 'geom_5_kekulized', 'geom_5_aromatic', 'qm9', 'geom'):
        if f'/{name}/' in nah or nah.startswith(f'data/{name}'):
            cfg.n_atoms_hist = name
            break
    return cfg.validate()
