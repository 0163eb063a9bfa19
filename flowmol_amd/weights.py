"""State-dict layout of the reference ``CTMCVectorField`` and a deterministic filler.

No checkpoint ships with the reference (``*.ckpt`` is git-LFS and downloaded at
run time, reference flowmol/__init__.py:42-44,58-77), and there is no network, so
every parity check and benchmark runs on *weights-by-name*: each tensor is filled
from a generator seeded by crc32 of its state-dict key.  The key/shape list below
reproduces the reference module tree; ``oracle/make_golden.py``
verifies it with ``load_state_dict(strict=True)`` on the reference's own module.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import VFConfig


def _gvp_shapes(prefix: str, vin: int, vout: int, sin: int, sout: int, ncp: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameters of one reference ``GVP`` (gvp.py:30-88); hidden = max(vin, vout)."""
    h = max(vin, vout)
    d = OrderedDict()
    d[f'{prefix}.Wh'] = (vin, h)
    if ncp > 0:
        d[f'{prefix}.Wcp'] = (vin, 2 * ncp)
    d[f'{prefix}.Wu'] = (h + ncp, vout)
    d[f'{prefix}.to_feats_out.0.weight'] = (sout, h + ncp + sin)
    d[f'{prefix}.to_feats_out.0.bias'] = (sout,)
    d[f'{prefix}.scalar_to_vector_gates.weight'] = (vout, sout)
    d[f'{prefix}.scalar_to_vector_gates.bias'] = (vout,)
    return d


def state_dict_shapes(cfg: VFConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    S, V, F, R = cfg.n_hidden_scalars, cfg.n_vec_channels, cfg.n_hidden_edge_feats, cfg.rbf_dim
    ncp = cfg.n_cp_feats
    na, nc, ne = cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types
    ta, tc, te = cfg.token_dims
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if cfg.a_token_dim:
        d['token_embeddings.a.weight'] = (na + 1, ta)
        d['token_embeddings.c.weight'] = (nc + 1, tc)
        d['token_embeddings.e.weight'] = (ne + 1, te)
    d['scalar_embedding.0.weight'] = (S, ta + tc + cfg.time_embedding_dim)
    d['scalar_embedding.0.bias'] = (S,)
    d['scalar_embedding.2.weight'] = (S, S)
    d['scalar_embedding.2.bias'] = (S,)
    d['scalar_embedding.4.weight'] = (S,)
    d['scalar_embedding.4.bias'] = (S,)
    d['edge_embedding.0.weight'] = (F, te)
    d['edge_embedding.0.bias'] = (F,)
    d['edge_embedding.2.weight'] = (F, F)
    d['edge_embedding.2.bias'] = (F,)
    d['edge_embedding.4.weight'] = (F,)
    d['edge_embedding.4.bias'] = (F,)
    for i in range(cfg.n_convs):
        p = f'conv_layers.{i}'
        sd_, vd_ = cfg.s_dst_feats, cfg.v_dst_feats
        if cfg.use_dst_feats:        # gvp.py:300-311: GVP(V -> V/r vectors, S -> S/r scalars, no cross-product features)
            d.update(_gvp_shapes(f'{p}.dst_feat_msg_projection', V, vd_, S, sd_, 0))
        for g in range(cfg.n_message_gvps):
            if g == 0:
                d.update(_gvp_shapes(f'{p}.edge_message.{g}', V + 1 + vd_, V, S + R + F + sd_, S, ncp))
            else:
                d.update(_gvp_shapes(f'{p}.edge_message.{g}', V, V, S, S, ncp))
        for g in range(cfg.n_update_gvps):
            d.update(_gvp_shapes(f'{p}.node_update.{g}', V, V, S, S, ncp))
        d[f'{p}.dropout.vector_dropout.dummy_param'] = (0,)
        d[f'{p}.message_layer_norm.feat_norm.weight'] = (S,)
        d[f'{p}.message_layer_norm.feat_norm.bias'] = (S,)
        d[f'{p}.update_layer_norm.feat_norm.weight'] = (S,)
        d[f'{p}.update_layer_norm.feat_norm.bias'] = (S,)
    for u in range(cfg.n_updaters):
        p = f'node_position_updaters.{u}.gvps'
        d.update(_gvp_shapes(f'{p}.0', V, V, S, S, ncp))
        d.update(_gvp_shapes(f'{p}.1', V, V, S, S, ncp))
        d.update(_gvp_shapes(f'{p}.2', V, 1, S, S, ncp))
    for u in range(cfg.n_updaters):
        p = f'edge_updaters.{u}'
        d[f'{p}.edge_update_fn.0.weight'] = (F, 2 * S + F + (R if cfg.update_edge_w_distance else 0))
        d[f'{p}.edge_update_fn.0.bias'] = (F,)
        d[f'{p}.edge_update_fn.2.weight'] = (F, F)
        d[f'{p}.edge_update_fn.2.bias'] = (F,)
        d[f'{p}.edge_norm.weight'] = (F,)
        d[f'{p}.edge_norm.bias'] = (F,)
    d['node_output_head.0.weight'] = (S, S)
    d['node_output_head.0.bias'] = (S,)
    d['node_output_head.2.weight'] = (na + nc, S)
    d['node_output_head.2.bias'] = (na + nc,)
    d['to_edge_logits.0.weight'] = (F, F)
    d['to_edge_logits.0.bias'] = (F,)
    d['to_edge_logits.2.weight'] = (ne, F)
    d['to_edge_logits.2.bias'] = (ne,)
    if cfg.self_conditioning:
        p = 'self_conditioning_residual_layer'
        d[f'{p}.node_residual_mlp.0.weight'] = (S, S + na + nc + R)
        d[f'{p}.node_residual_mlp.0.bias'] = (S,)
        d[f'{p}.node_residual_mlp.2.weight'] = (S, S)
        d[f'{p}.node_residual_mlp.2.bias'] = (S,)
        d[f'{p}.edge_residual_mlp.0.weight'] = (F, F + ne + R)
        d[f'{p}.edge_residual_mlp.0.bias'] = (F,)
        d[f'{p}.edge_residual_mlp.2.weight'] = (F, F)
        d[f'{p}.edge_residual_mlp.2.bias'] = (F,)
    return d


def n_params(cfg: VFConfig) -> int:
    return sum(math.prod(s) for s in state_dict_shapes(cfg).values())


def _is_norm(key: str) -> bool:
    return ('feat_norm' in key or 'edge_norm' in key
            or key.startswith('scalar_embedding.4') or key.startswith('edge_embedding.4'))


def synth_state_dict(cfg: VFConfig, seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic weights-by-name fill.

    * matrices / biases: U(-k, k), k = 1/sqrt(fan_in) (the reference's own init scale,
      gvp.py:49-68 and nn.Linear's default);
    * LayerNorm gain 1 + 0.1 U(-1,1), bias 0.1 U(-1,1) (non-trivial so affine bugs show);
    * embeddings: U(-sqrt(3), sqrt(3)) (unit variance like nn.Embedding's N(0,1)).
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    shapes = state_dict_shapes(cfg)
    for key, shape in shapes.items():
        g = torch.Generator(device='cpu')
        g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
        if len(shape) == 1 and shape[0] == 0:
            sd[key] = torch.empty(0, dtype=dtype)
            continue
        u = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        if _is_norm(key):
            t = (1.0 + 0.1 * u) if key.endswith('weight') else 0.1 * u
        elif key.startswith('token_embeddings'):
            t = u * math.sqrt(3.0)
        elif key.endswith('.bias'):
            # fan_in of the matching weight
            wshape = shapes[key[:-4] + 'weight']
            t = u / math.sqrt(wshape[1])
        elif key.rsplit('.', 1)[-1] in ('Wh', 'Wcp', 'Wu'):
            t = u / math.sqrt(shape[0])
        else:  # nn.Linear weight (out, in)
            t = u / math.sqrt(shape[1])
        sd[key] = t.to(dtype)
    return sd


def check_state_dict(cfg: VFConfig, sd: Dict[str, torch.Tensor], prefix: str = '') -> None:
    """Raise if ``sd`` does not hold exactly the tensors this config needs."""
    want = state_dict_shapes(cfg)
    for key, shape in want.items():
        k = prefix + key
        if k not in sd:
            raise KeyError(f"missing weight {k!r}")
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError(f"weight {k!r}: shape {tuple(sd[k].shape)} != expected {tuple(shape)}")


def scaled_weights(sd: Dict[str, torch.Tensor], scale: float, pos_head_scale: float = 1.0, cat_head_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Every Linear / GVP weight matrix times ``scale`` (biases, LayerNorm affine parameters and embeddings untouched).  Synthetic weights
    as drawn (scale 1) give a network whose position updates are < 0.1 % of the coordinate scale; x2 moves atoms by 3-8 % per evaluation on a
    still well-conditioned trajectory; x3 amplifies rounding differences ~10x per convolution (the ill-conditioned regime of the parity tests).
    ``pos_head_scale``: additionally, the output projection ``Wu`` of the LAST GVP of every NodePositionUpdate (vector_field.py:813-842) times
    this factor -- the position update is linear in it, so x128 turns the 0.04 % per evaluation of unit weights into ~4.6 % while every other
    activation keeps its unit scale: trajectories whose coordinates really depend on 250 network evaluations and stay well-conditioned (a 1-ulp
    perturbation of x_0 moves the result by 2e-7; oracle/make_golden.py LONG_CASES).
    ``cat_head_scale``: the last Linear of the categorical output heads (``node_output_head.2.weight``, ``to_edge_logits.2.weight``: vector_field.py:336-344)
    times this factor.  Weights-by-name give head logits that differ by well under 1, so the tempered probabilities softmax(log p / 0.05)
    (ctmc_vector_field.py:354-356) never hold an exact zero and p never saturates; x256 puts the heads where a TRAINED model lives: most classes of the
    tempered distribution exactly 0, p == 1.0 rows, exact zeros (log 0 = -inf) and denormals in p."""
    out = dict(sd) if scale == 1 else {
        k: (v * scale if ('weight' in k or k.endswith(('Wh', 'Wu', 'Wcp'))) and 'norm' not in k and '.4.' not in k and 'token_embeddings' not in k else v)
        for k, v in sd.items()}
    if pos_head_scale != 1:
        out = {k: (v * pos_head_scale if ('node_position_updaters' in k and k.endswith('gvps.2.Wu')) else v) for k, v in out.items()}
    if cat_head_scale != 1:
        out = {k: (v * cat_head_scale if k.endswith(('node_output_head.2.weight', 'to_edge_logits.2.weight')) else v) for k, v in out.items()}
    return out


def long_fixture_weights(cfg: VFConfig, g) -> Dict[str, torch.Tensor]:
    """The weights a tests/golden/long_*.npz fixture was generated with: weights-by-name (seed 0), scaled as the fixture records."""
    return scaled_weights(synth_state_dict(cfg, 0), float(g['weight_scale']), float(g['pos_head_scale']) if 'pos_head_scale' in g else 1.0,
                          float(g['cat_head_scale']) if 'cat_head_scale' in g else 1.0)
