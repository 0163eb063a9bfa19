"""CPU oracle: an eager-PyTorch, op-for-op restatement of the reference's sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (``flowmol_amd/``) imports
this file; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may use it, and only as the checker / the timed CPU baseline -- never as the product.

What it restates (all citations are into /root/reference):
  graph construction      flowmol/data_processing/utils.py:4-46, flowmol/models/flowmol.py:509-529
  priors                  flowmol/data_processing/priors.py:27-35,101-107,305-316
  alpha schedule          flowmol/models/interpolant_scheduler.py:97-153 (linear and cosine)
  embeddings              flowmol/utils/embedding.py:5-34
  GVP / GVPLayerNorm      flowmol/models/gvp.py:14-21,90-133,169-184
  GVPConv                 flowmol/models/gvp.py:435-543
  NodePositionUpdate      flowmol/models/vector_field.py:813-842
  EdgeUpdate              flowmol/models/vector_field.py:844-880
  self-conditioning       flowmol/models/self_conditioning.py:37-102
  forward / denoise       flowmol/models/vector_field.py:212-386
  CTMC step / integrate   flowmol/models/ctmc_vector_field.py:145-461
  purity sampling         flowmol/utils/ctmc_utils.py:4-34
  molecule extraction     flowmol/analysis/molecule_builder.py:217-265

Third-party arithmetic not under /root/reference (DGL 2.0.0 ``u_sub_v`` / ``copy_e``+``sum`` /
``readout_nodes(mean)``, torch_scatter 2.1.2 ``segment_csr``) is restated as gather /
``index_add_`` / segment sums; DGL's own reduction order is unspecified, so float parity with a
real DGL run is to summation order, not bitwise.

Pinning: the reference has NO tests or golden vectors for this path, so the pins
are outputs of the reference's own modules run in the build container: ``oracle/make_golden.py``
imports ``/root/reference/flowmol`` with the pure-torch stand-ins in ``oracle/ref_standin.py`` and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against them
wherever it runs (the GPU box has no /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import einsum


# --------------------------------------------------------------------------------------
# graph batching  (data_processing/utils.py:4-46, flowmol.py:509-529)
# --------------------------------------------------------------------------------------
def build_edge_idxs(n_atoms: int) -> torch.Tensor:
    """(2, n(n-1)) edge list: upper triangle (src<dst, row-major) then the same pairs swapped."""
    up = torch.triu_indices(n_atoms, n_atoms, offset=1)
    lo = torch.stack((up[1], up[0]))
    return torch.cat((up, lo), dim=1)


@dataclass
class Batch:
    n_atoms: torch.Tensor          # (B,) int64
    src: torch.Tensor              # (E,) int64, global node ids, reference edge order
    dst: torch.Tensor              # (E,)
    upper_edge_mask: torch.Tensor  # (E,) bool
    node_batch_idx: torch.Tensor   # (N,)
    edge_batch_idx: torch.Tensor   # (E,)

    @property
    def B(self): return int(self.n_atoms.shape[0])
    @property
    def N(self): return int(self.n_atoms.sum())
    @property
    def E(self): return int(self.src.shape[0])
    @property
    def batch_num_nodes(self): return self.n_atoms
    @property
    def batch_num_edges(self): return self.n_atoms * (self.n_atoms - 1)


def build_batch(n_atoms: torch.Tensor) -> Batch:
    n_atoms = n_atoms.to(torch.int64).cpu()
    srcs, dsts, masks = [], [], []
    off = 0
    for n in n_atoms.tolist():
        e = build_edge_idxs(n)
        srcs.append(e[0] + off)
        dsts.append(e[1] + off)
        u = n * (n - 1) // 2
        masks.append(torch.cat([torch.ones(u, dtype=torch.bool), torch.zeros(u, dtype=torch.bool)]))
        off += n
    B = n_atoms.shape[0]
    nb = torch.arange(B).repeat_interleave(n_atoms)
    eb = torch.arange(B).repeat_interleave(n_atoms * (n_atoms - 1))
    return Batch(n_atoms, torch.cat(srcs), torch.cat(dsts), torch.cat(masks), nb, eb)


# --------------------------------------------------------------------------------------
# small numeric helpers
# --------------------------------------------------------------------------------------
def norm_no_nan(x, axis=-1, keepdims=False, eps=1e-8, sqrt=True):
    """gvp.py:14-21: clamps the SQUARED norm at eps."""
    out = torch.clamp(torch.sum(torch.square(x), axis, keepdims), min=eps)
    return torch.sqrt(out) if sqrt else out


def rbf(D, D_max, D_count, D_min=0.0):
    """embedding.py:19-34."""
    mu = torch.linspace(D_min, D_max, D_count, device=D.device).view([1, -1])
    sigma = (D_max - D_min) / D_count
    return torch.exp(-((torch.unsqueeze(D, -1) - mu) / sigma) ** 2)


def time_embedding(t, dim, max_positions=1000):
    """embedding.py:5-17."""
    t = t * max_positions
    half = dim // 2
    emb = math.log(max_positions) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32, device=t.device) * -emb)
    emb = t.float()[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1), mode='constant')
    return emb


def alpha_tables(t: torch.Tensor, schedule_type=None, cosine_params=None):
    """alpha_t / alpha_t_prime for x,a,c,e (interpolant_scheduler.py:97-153).  linear: alpha=t, alpha'=1; cosine:
    alpha = 1 - cos^2(pi/2 t^nu), alpha' = pi/2 sin(pi t^nu) nu t^(nu-1) -- whose evaluation clamps ``t`` IN PLACE to >= 1e-9
    (:140-141) after alpha was computed from the unclamped tensor (ctmc_vector_field.py:175-176)."""
    schedule_type = schedule_type or {}
    nus = {k: torch.tensor(v).unsqueeze(0) for k, v in (cosine_params or {}).items()}
    a = []
    for k in 'xace':
        if schedule_type.get(k, 'linear') == 'cosine':
            a.append(1 - torch.cos(torch.pi * 0.5 * torch.pow(t.unsqueeze(-1), nus[k])).square())
        else:
            a.append(t.unsqueeze(-1))
    a = torch.cat(a, dim=1)                 # materialised BEFORE the derivative's in-place clamp, as in the reference (alpha_t returns a cat)
    ap = []
    for k in 'xace':
        if schedule_type.get(k, 'linear') == 'cosine':
            t = torch.clamp_(t, min=1e-9)
            tt = t.unsqueeze(-1)
            ap.append(torch.pi * 0.5 * torch.sin(torch.pi * torch.pow(tt, nus[k])) * nus[k] * torch.pow(tt, nus[k] - 1))
        else:
            ap.append(torch.ones_like(t).unsqueeze(-1))
    return a, torch.cat(ap, dim=1)


# --------------------------------------------------------------------------------------
# priors  (priors.py:27-35,101-107,305-316)
# --------------------------------------------------------------------------------------
def centered_normal_prior(batch: Batch, device='cpu'):
    """x0 ~ N(0, I) minus per-molecule mean; the reference ignores its ``std`` argument."""
    x = torch.randn(batch.N, 3, device=device)
    return x - segment_mean(x, batch.node_batch_idx.to(device), batch.B)[batch.node_batch_idx.to(device)]


def ctmc_masked_prior(n: int, d: int):
    return F.one_hot(torch.full((n,), fill_value=d), num_classes=d + 1).float()


def edge_prior(upper_edge_mask: torch.Tensor, n_bond_types: int):
    nu = int(upper_edge_mask.sum())
    up = ctmc_masked_prior(nu, n_bond_types)
    out = torch.zeros(upper_edge_mask.shape[0], up.shape[1])
    out[upper_edge_mask] = up
    out[~upper_edge_mask] = up
    return out


def segment_mean(x, seg_idx, n_seg):
    """DGL readout_nodes(op='mean') restated as segment sum / count."""
    s = torch.zeros((n_seg,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device).index_add_(0, seg_idx, x)
    cnt = torch.zeros(n_seg, dtype=x.dtype, device=x.device).index_add_(
        0, seg_idx, torch.ones(x.shape[0], dtype=x.dtype, device=x.device))
    return s / cnt.view(-1, *([1] * (x.dim() - 1)))


# --------------------------------------------------------------------------------------
# noise sources: the reference draws from torch's global RNG; the draws are made explicit here
# so the HIP path can be fed the identical numbers.
# Order per step, per modality a,c,e (ctmc_vector_field.py:428; ctmc_utils.py:34 or
# ctmc_vector_field.py:445; ctmc_vector_field.py:450):  Exp(1) of (rows,K)  -> rand(rows) -> rand(rows) [not on last step]
# --------------------------------------------------------------------------------------
class TorchNoise:
    """Draw from torch's global generator exactly as the reference's ops would."""
    def exp_like(self, p):
        return torch.empty_like(p).exponential_(1)

    def rand(self, n, device):
        return torch.rand(n, device=device)


class RecordingNoise(TorchNoise):
    def __init__(self):
        self.tape: List[torch.Tensor] = []

    def exp_like(self, p):
        q = super().exp_like(p)
        self.tape.append(q.clone())
        return q

    def rand(self, n, device):
        u = super().rand(n, device)
        self.tape.append(u.clone())
        return u


class TapeNoise:
    def __init__(self, tape):
        self.tape = list(tape)
        self.pos = 0

    def _next(self):
        t = self.tape[self.pos]
        self.pos += 1
        return t

    def exp_like(self, p):
        q = self._next()
        assert q.shape == p.shape, (q.shape, p.shape)
        return q.to(p.device)

    def rand(self, n, device):
        u = self._next()
        assert u.shape == (n,), (u.shape, n)
        return u.to(device)


# --------------------------------------------------------------------------------------
# the network
# --------------------------------------------------------------------------------------
class OracleVF:
    """Restatement of CTMCVectorField (inference only) on a plain state dict."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], prefix: str = ''):
        self.cfg = cfg
        self.p = {k[len(prefix):]: v.detach().to(torch.float32) for k, v in state_dict.items() if k.startswith(prefix)}
        self.na, self.nc, self.ne = cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types
        self.mask_idx = {'a': self.na, 'c': self.nc, 'e': self.ne}
        self.taps: Optional[Dict[str, torch.Tensor]] = None   # filled when tap recording is on

    def to(self, device):
        self.p = {k: v.to(device) for k, v in self.p.items()}
        return self

    def _tap(self, name, t):
        if self.taps is not None:
            self.taps[name] = t.detach().clone()

    # ---------------- building blocks
    def _lin(self, key, x):
        return F.linear(x, self.p[key + '.weight'], self.p[key + '.bias'])

    def _ln(self, key, x):
        return F.layer_norm(x, (x.shape[-1],), self.p[key + '.weight'], self.p[key + '.bias'], 1e-5)

    def gvp(self, key, feats, vectors, vec_act='sigmoid'):
        """gvp.py:90-133 (vector_gating=True always on this path)."""
        p = self.p
        ncp = self.cfg.n_cp_feats if (key + '.Wcp') in p else 0          # the destination-feature projection GVP has none (gvp.py:309)
        Vh = einsum('b v c, v h -> b h c', vectors, p[key + '.Wh'])
        if ncp > 0:
            Vcp = einsum('b v c, v p -> b p c', vectors, p[key + '.Wcp'])
            cp_src, cp_dst = torch.split(Vcp, ncp, dim=1)
            cp = torch.linalg.cross(cp_src, cp_dst, dim=-1)
            Vh = torch.cat((Vh, cp), dim=1)
        Vu = einsum('b h c, h u -> b u c', Vh, p[key + '.Wu'])
        sh = norm_no_nan(Vh)
        s = torch.cat((feats, sh), dim=1)
        feats_out = F.silu(self._lin(key + '.to_feats_out.0', s))
        gating = self._lin(key + '.scalar_to_vector_gates', feats_out).unsqueeze(-1)
        if vec_act == 'sigmoid':
            gating = torch.sigmoid(gating)
        return feats_out, gating * Vu

    def gvp_layer_norm(self, key, feats, vectors, eps=1e-5):
        """gvp.py:169-184."""
        nf = self._ln(key + '.feat_norm', feats)
        vn = norm_no_nan(vectors, axis=-1, keepdims=True, sqrt=False)
        vn = torch.sqrt(torch.mean(vn, dim=-2, keepdim=True) + eps) + eps
        return nf, vectors / vn

    def distances(self, batch: Batch, x):
        """precompute_distances, vector_field.py:371-386: x_diff = x[src]-x[dst] (DGL u_sub_v)."""
        xd = x[batch.src] - x[batch.dst]
        dij = norm_no_nan(xd, keepdims=True) + 1e-8
        return xd / dij, rbf(dij.squeeze(1), D_max=self.cfg.rbf_dmax, D_count=self.cfg.rbf_dim)

    def conv(self, i, batch: Batch, s, v, ef, x_diff, d, tap_msgs=True):
        """GVPConv.forward / message, gvp.py:435-543 (no attention / compression); with use_dst_feats the projected
        destination-node features join the message inputs (gvp.py:300-316, 472-473, 527-537)."""
        key = f'conv_layers.{i}'
        vec = [x_diff.unsqueeze(1), v[batch.src]]
        sca = [s[batch.src], d, ef]
        if self.cfg.use_dst_feats:
            s_dm, v_dm = self.gvp(f'{key}.dst_feat_msg_projection', s, v)
            vec.append(v_dm[batch.dst])
            sca.append(s_dm[batch.dst])
        vec = torch.cat(vec, dim=1)
        sca = torch.cat(sca, dim=1)
        for g in range(self.cfg.n_message_gvps):
            sca, vec = self.gvp(f'{key}.edge_message.{g}', sca, vec)
            if i == 0 and tap_msgs:
                self._tap(f'conv0.msg{g}.s', sca)
                self._tap(f'conv0.msg{g}.v', vec)
        N = s.shape[0]
        ms = torch.zeros(N, sca.shape[1], dtype=s.dtype, device=s.device).index_add_(0, batch.dst, sca)
        mv = torch.zeros(N, vec.shape[1], 3, dtype=s.dtype, device=s.device).index_add_(0, batch.dst, vec)
        if self.cfg.message_norm == 'mean':        # DGL fn.mean over the in-edges (gvp.py:401-404); nodes without in-edges get 0
            deg = torch.zeros(N, dtype=s.dtype, device=s.device).index_add_(0, batch.dst, torch.ones_like(batch.dst, dtype=s.dtype))
            deg = deg.clamp(min=1.0)
            ms = ms / deg[:, None]
            mv = mv / deg[:, None, None]
        else:
            z = self.cfg.msg_z
            ms = ms / z
            mv = mv / z
        self._tap(f'conv{i}.agg.s', ms)
        self._tap(f'conv{i}.agg.v', mv)
        s1, v1 = self.gvp_layer_norm(f'{key}.message_layer_norm', s + ms, v + mv)
        rs, rv = s1, v1
        for g in range(self.cfg.n_update_gvps):
            rs, rv = self.gvp(f'{key}.node_update.{g}', rs, rv)
        s2, v2 = self.gvp_layer_norm(f'{key}.update_layer_norm', s1 + rs, v1 + rv)
        return s2, v2

    def position_update(self, u, s, x, v):
        """NodePositionUpdate, vector_field.py:813-842 (last GVP: 1 vector out, identity gate activation)."""
        key = f'node_position_updaters.{u}.gvps'
        fs, fv = self.gvp(f'{key}.0', s, v)
        fs, fv = self.gvp(f'{key}.1', fs, fv)
        _, upd = self.gvp(f'{key}.2', fs, fv, vec_act='identity')
        return x + upd.squeeze(1)

    def edge_update(self, u, batch: Batch, s, ef, d):
        """EdgeUpdate, vector_field.py:844-880."""
        key = f'edge_updaters.{u}'
        parts = [s[batch.src], s[batch.dst], ef]
        if self.cfg.update_edge_w_distance:           # vector_field.py:876-877
            parts.append(d)
        inp = torch.cat(parts, dim=-1)
        h = F.silu(self._lin(f'{key}.edge_update_fn.0', inp))
        h = F.silu(self._lin(f'{key}.edge_update_fn.2', h))
        return self._ln(f'{key}.edge_norm', ef + h)

    def self_conditioning(self, batch: Batch, s, x, ef, prev):
        """SelfConditioningResidualLayer.forward, self_conditioning.py:37-85."""
        key = 'self_conditioning_residual_layer'
        cfg = self.cfg
        d_node = rbf(norm_no_nan(x - prev['x']), D_max=cfg.rbf_dmax, D_count=cfg.rbf_dim)
        inp = torch.cat([s, prev['a'], prev['c'], d_node], dim=-1)
        res = F.silu(self._lin(f'{key}.node_residual_mlp.0', inp))
        res = F.silu(self._lin(f'{key}.node_residual_mlp.2', res))
        m = batch.upper_edge_mask

        def edge_d(pos):
            xd = pos[batch.src] - pos[batch.dst]
            dij = norm_no_nan(xd, keepdims=True) + 1e-8
            return rbf(dij.squeeze(1), D_max=cfg.rbf_dmax, D_count=cfg.rbf_dim)

        d_t = edge_d(x)[m]
        d_1 = edge_d(prev['x'])[m]
        einp = torch.cat([ef[m], prev['e'], d_1 - d_t], dim=-1)
        eres = F.silu(self._lin(f'{key}.edge_residual_mlp.0', einp))
        eres = F.silu(self._lin(f'{key}.edge_residual_mlp.2', eres))
        ef_out = torch.zeros_like(ef)
        one = ef[m] + eres
        ef_out[m] = one
        ef_out[~m] = one
        return s + res, ef_out

    def denoise(self, batch: Batch, s, v, x, ef, apply_softmax=False, remove_com=False):
        """denoise_graph, vector_field.py:296-369."""
        cfg = self.cfg
        x_diff, d = self.distances(batch, x)
        sched = cfg.update_schedule()
        for it in range(cfg.n_convs * cfg.n_recycles):       # vector_field.py:307: the stack is repeated n_recycles times over the same weights
            i = it % cfg.n_convs
            s, v = self.conv(i, batch, s, v, ef, x_diff, d, tap_msgs=it == 0)
            self._tap(f'conv{i}.s', s)
            self._tap(f'conv{i}.v', v)
            u = sched[i]
            if u >= 0:
                x = self.position_update(u, s, x, v)
                x_diff, d = self.distances(batch, x)
                ef = self.edge_update(u, batch, s, ef, d)
                self._tap(f'upd{i}.x', x)
                self._tap(f'upd{i}.ef', ef)
        h = self._lin('node_output_head.2', F.silu(self._lin('node_output_head.0', s)))
        a_logits = h[:, :self.na]
        c_logits = h[:, self.na:]
        m = batch.upper_edge_mask
        e_in = ef[m] + ef[~m]
        e_logits = self._lin('to_edge_logits.2', F.silu(self._lin('to_edge_logits.0', e_in)))
        if remove_com:
            x = x - segment_mean(x, batch.node_batch_idx, batch.B)[batch.node_batch_idx]
        out = {'x': x, 'a': a_logits, 'c': c_logits, 'e': e_logits}
        if apply_softmax:
            for k in ('a', 'c', 'e'):
                out[k] = torch.softmax(out[k], dim=-1)
        return out

    def embed(self, batch: Batch, a_t, c_t, e_t, t):
        """Input embeddings, vector_field.py:226-261.  ``t`` has shape (B,)."""
        cfg, p = self.cfg, self.p
        feats = []
        if cfg.a_token_dim:
            feats.append(F.embedding(a_t.argmax(dim=-1), p['token_embeddings.a.weight']))
            feats.append(F.embedding(c_t.argmax(dim=-1), p['token_embeddings.c.weight']))
        else:
            feats.append(a_t)
            feats.append(c_t)
        if cfg.time_embedding_dim == 1:
            feats.append(t[batch.node_batch_idx].unsqueeze(-1))
        else:
            feats.append(time_embedding(t, cfg.time_embedding_dim)[batch.node_batch_idx])
        s = torch.cat(feats, dim=-1)
        s = F.silu(self._lin('scalar_embedding.0', s))
        s = F.silu(self._lin('scalar_embedding.2', s))
        s = self._ln('scalar_embedding.4', s)
        v = torch.zeros((a_t.shape[0], cfg.n_vec_channels, 3), device=s.device)
        if cfg.e_token_dim:
            ef = F.embedding(e_t.argmax(dim=-1), p['token_embeddings.e.weight'])
        else:
            ef = e_t
        ef = F.silu(self._lin('edge_embedding.0', ef))
        ef = F.silu(self._lin('edge_embedding.2', ef))
        ef = self._ln('edge_embedding.4', ef)
        return s, v, ef

    def forward(self, batch: Batch, x_t, a_t, c_t, e_t, t, prev=None, apply_softmax=False, remove_com=False):
        """EndpointVectorField.forward (eval mode), vector_field.py:212-293."""
        s, v, ef = self.embed(batch, a_t, c_t, e_t, t)
        self._tap('embed.s', s)
        self._tap('embed.ef', ef)
        x = x_t
        if self.cfg.self_conditioning and prev is None:
            if bool((t == 0).all().item()):
                taps, self.taps = self.taps, None
                prev = self.denoise(batch, s.clone(), v.clone(), x.clone(), ef.clone(),
                                    apply_softmax=True, remove_com=False)
                self.taps = taps
                if self.taps is not None:
                    for k in 'xace':
                        self.taps[f'boot.{k}'] = prev[k].detach().clone()
        if self.cfg.self_conditioning and prev is not None:
            s, ef = self.self_conditioning(batch, s, x, ef, prev)
            self._tap('sc.s', s)
            self._tap('sc.ef', ef)
        return self.denoise(batch, s, v, x, ef, apply_softmax, remove_com)

    # ---------------- CTMC update
    @staticmethod
    def purity_sampling(xt, x1_probs, unmask_prob, mask_index, batch_size, node_batch_idx, hc_thresh, u):
        """ctmc_utils.py:4-34 with the uniform draw ``u`` made explicit; segment_csr -> index_add."""
        masked = xt == mask_index
        purities = x1_probs.max(-1)[0]
        hc_mask = (purities >= hc_thresh) * masked
        hc_per = torch.zeros(batch_size, dtype=torch.long, device=xt.device).index_add_(0, node_batch_idx, hc_mask.long())
        m_per = torch.zeros(batch_size, dtype=torch.long, device=xt.device).index_add_(0, node_batch_idx, masked.long())
        ph_max = unmask_prob * m_per / hc_per
        ph_max[hc_per == 0] = torch.inf
        ph = torch.minimum(ph_max, torch.full_like(ph_max, 1.0))
        pl = (unmask_prob * m_per - ph * hc_per) / (m_per - hc_per)
        prob = torch.zeros_like(xt).float()
        prob[hc_mask] = ph[node_batch_idx[hc_mask]]
        lc_mask = (purities < hc_thresh) * masked
        prob[lc_mask] = pl[node_batch_idx[lc_mask]]
        return u < prob

    def campbell_step(self, p_1_given_t, xt, eta, hc_thresh, alpha_t, alpha_t_prime, dt, batch_size,
                      n_classes, mask_index, last_step, batch_idx, noise):
        """ctmc_vector_field.py:414-461.  Categorical(p).sample() == argmax((p/sum p)/q), q~Exp(1)."""
        pn = p_1_given_t / p_1_given_t.sum(-1, keepdim=True)
        q = noise.exp_like(pn)
        x1 = torch.argmax(pn / q, dim=-1)
        unmask_prob = torch.clamp(dt * (alpha_t_prime + eta * alpha_t) / (1 - alpha_t), min=0, max=1)
        mask_prob = torch.clamp(dt * eta, min=0, max=1)
        if hc_thresh > 0:
            u1 = noise.rand(xt.shape[0], xt.device)
            will_unmask = self.purity_sampling(xt, p_1_given_t, unmask_prob, mask_index, batch_size,
                                               batch_idx, hc_thresh, u1)
        else:
            u1 = noise.rand(xt.shape[0], xt.device)
            will_unmask = (u1 < unmask_prob) * (xt == mask_index)
        xt = xt.clone()
        if not last_step:
            u2 = noise.rand(xt.shape[0], xt.device)
            will_mask = (u2 < mask_prob) * (xt != mask_index)
            xt[will_mask] = mask_index
        xt[will_unmask] = x1[will_unmask]
        return F.one_hot(xt, num_classes=n_classes).float(), F.one_hot(x1, num_classes=n_classes).float()

    def gat_step(self, p_1_given_t, xt, alpha_t, alpha_t_prime, forward_weight, dt, n_classes, mask_index, noise):
        """ctmc_vector_field.py:463-510: one draw from the clamped transition distribution over K+1 classes."""
        p_1_given_t = torch.cat([p_1_given_t, torch.zeros_like(p_1_given_t[:, :1])], dim=-1)
        delta_xt = F.one_hot(xt, num_classes=n_classes).float()
        u_forward = alpha_t_prime / (1 - alpha_t) * (p_1_given_t - delta_xt)
        delta_mask = torch.zeros_like(delta_xt)
        delta_mask[:, mask_index] = 1
        u_backward = alpha_t_prime / (alpha_t + 1e-8) * (delta_xt - delta_mask)
        backward_weight = forward_weight - 1
        pvel = forward_weight * u_forward - backward_weight * u_backward
        p_step = torch.clamp(delta_xt + dt * pvel, min=1.0e-9, max=1)
        pn = p_step / p_step.sum(-1, keepdim=True)            # Categorical normalises; sample == argmax(pn/q)
        q = noise.exp_like(pn)
        x_dt = torch.argmax(pn / q, dim=-1)
        return F.one_hot(x_dt, num_classes=n_classes).float()

    def step(self, batch: Batch, state: Dict[str, torch.Tensor], s_i, t_i, alpha_t_i, alpha_t_prime_i,
             prev, eta, hc_thresh, last_step, noise, cat_temp=None, dfm_type='campbell', forward_weight=None,
             inv_temp=1.0):
        """CTMCVectorField.step, ctmc_vector_field.py:287-411.  cat_temp / forward_weight / inv_temp are the VALUES
        of cat_temp_func(t_i) / forward_weight_func(t_i) / inv_temp_func(t_i)."""
        cfg = self.cfg
        dev = state['x_t'].device
        dst = self.forward(batch, state['x_t'], state['a_t'], state['c_t'], state['e_t'],
                           t=torch.full((batch.B,), float(t_i), device=dev) if not torch.is_tensor(t_i)
                           else torch.full((batch.B,), t_i, device=dev),
                           prev=prev, apply_softmax=True, remove_com=True)
        dt = s_i - t_i
        x_1 = dst['x']
        x_t = state['x_t']
        vf = alpha_t_prime_i[0] / (1 - alpha_t_i[0]) * (x_1 - x_t)
        new = {'x_t': x_t + dt * vf * inv_temp, 'x_1_pred': x_1.detach().clone()}
        m = batch.upper_edge_mask
        temperature = cfg.cat_temperature if cat_temp is None else cat_temp
        for fi, feat in enumerate(['x', 'a', 'c', 'e']):
            if feat == 'x':
                continue
            xt = state[f'{feat}_t'].argmax(-1)
            if feat == 'e':
                xt = xt[m]
            p = F.softmax(torch.log(dst[feat]) / temperature, dim=-1)
            bidx = batch.edge_batch_idx[m] if feat == 'e' else batch.node_batch_idx
            n_cls = {'a': self.na, 'c': self.nc, 'e': self.ne}[feat] + 1
            if dfm_type == 'campbell':
                xt1h, x11h = self.campbell_step(p, xt, eta, hc_thresh, alpha_t_i[fi], alpha_t_prime_i[fi], dt,
                                                batch.B, n_cls, self.mask_idx[feat], last_step, bidx, noise)
            else:
                x11h = torch.cat([p, torch.zeros_like(p[:, :1])], dim=-1)      # ctmc_vector_field.py:375
                xt1h = self.gat_step(p, xt, alpha_t_i[fi], alpha_t_prime_i[fi], forward_weight, dt, n_cls,
                                     self.mask_idx[feat], noise)
            if feat == 'e':
                e_t = torch.zeros_like(state['e_t'])
                e_t[m] = xt1h
                e_t[~m] = xt1h
                e_1 = torch.zeros_like(state['e_t'])
                e_1[m] = x11h
                e_1[~m] = x11h
                xt1h, x11h = e_t, e_1
            new[f'{feat}_t'] = xt1h
            new[f'{feat}_1_pred'] = x11h
        return new, dst

    def integrate(self, batch: Batch, prior: Dict[str, torch.Tensor], n_timesteps: int, eta=None, hc_thresh=None,
                  noise=None, visualize=False, tspan=None, step_hook=None, dfm_type='campbell', cat_temp_func=None,
                  forward_weight_func=None, inv_temp_func=None):
        """CTMCVectorField.integrate, ctmc_vector_field.py:145-285.  The *_func arguments are callables of the
        0-dim tensor t_i like the reference's (defaults: the configured constant temperature, forward weight 1,
        inverse temperature 1)."""
        cfg = self.cfg
        eta = cfg.stochasticity if eta is None else eta
        hc_thresh = cfg.high_confidence_threshold if hc_thresh is None else hc_thresh
        noise = noise or TorchNoise()
        dev = prior['x_0'].device
        t = torch.linspace(0, 1, n_timesteps, device=dev) if tspan is None else tspan
        alpha_t, alpha_tp = alpha_tables(t, getattr(cfg, 'schedule_type', None), getattr(cfg, 'cosine_params', None))
        state = {'x_t': prior['x_0'], 'a_t': prior['a_0'], 'c_t': prior['c_0'], 'e_t': prior['e_0']}
        frames = None
        if visualize:
            frames = {k: [state[f'{k}_t'].clone()] for k in 'xace'}
            frames.update({f'{k}_1_pred': [] for k in 'xace'})
        dst = None
        for s_idx in range(1, t.shape[0]):
            last = s_idx == t.shape[0] - 1
            t_i = t[s_idx - 1]
            new, dst = self.step(batch, state, t[s_idx], t_i, alpha_t[s_idx - 1], alpha_tp[s_idx - 1],
                                 prev=dst, eta=eta, hc_thresh=hc_thresh, last_step=last, noise=noise,
                                 cat_temp=None if cat_temp_func is None else cat_temp_func(t_i), dfm_type=dfm_type,
                                 forward_weight=1.0 if forward_weight_func is None else forward_weight_func(t_i),
                                 inv_temp=1.0 if inv_temp_func is None else inv_temp_func(t_i))
            state = {k: new[k] for k in ('x_t', 'a_t', 'c_t', 'e_t')}
            if step_hook is not None:
                step_hook(s_idx, new, dst)
            if visualize:
                for k in 'xace':
                    frames[k].append(new[f'{k}_t'].clone())
                    frames[f'{k}_1_pred'].append(new[f'{k}_1_pred'].clone())
        out = {'x_1': state['x_t'], 'a_1': state['a_t'], 'c_1': state['c_t'], 'e_1': state['e_t']}
        if visualize:
            return out, frames
        return out

    # ---------------- endpoint parameterization (EndpointVectorField.step / integrate, vector_field.py:388-569)
    def step_endpoint(self, batch: Batch, state, s_i, t_i, alpha_t_i, alpha_tp_i, prev=None, inv_temp=1.0):
        """One Euler step of all four modalities: x_s = x_t + (alpha'/(1-alpha) (x_1 - x_t)) * inv_temp * (s - t); the categorical
        features are continuous vectors (no mask state); edge state lives on the upper triangle and is mirrored."""
        dst = self.forward(batch, state['x_t'], state['a_t'], state['c_t'], state['e_t'], torch.full((batch.B,), float(t_i)), prev=prev,
                           apply_softmax=True, remove_com=True)
        m = batch.upper_edge_mask
        new = {}
        for idx, k in enumerate('xace'):
            x_t = state[f'{k}_t']
            x_1 = dst[k]
            if k == 'e':
                x_t = x_t[m]
            vf = alpha_tp_i[idx] / (1 - alpha_t_i[idx]) * (x_1 - x_t)
            vf = vf * inv_temp
            x_s = x_t + vf * (s_i - t_i)
            if k == 'e':
                full = torch.zeros_like(state['e_t'])
                full[m] = x_s
                full[~m] = x_s
                x_s = full
                one = torch.zeros_like(state['e_t'])
                one[m] = x_1
                one[~m] = x_1
                x_1 = one
            new[f'{k}_t'] = x_s
            new[f'{k}_1_pred'] = x_1.detach().clone()
        return new, dst

    def integrate_endpoint(self, batch: Batch, prior, n_timesteps: int, inv_temp_func=None):
        cfg = self.cfg
        t = torch.linspace(0, 1, n_timesteps)
        alpha_t, alpha_tp = alpha_tables(t, getattr(cfg, 'schedule_type', None), getattr(cfg, 'cosine_params', None))
        if inv_temp_func is None:
            if cfg.continuous_inv_temp_schedule == 'linear':
                inv_temp_func = lambda tt: cfg.continuous_inv_temp_max * (1 - tt)           # vector_field.py:203-204
            else:
                inv_temp_func = lambda tt: 1.0
        state = {'x_t': prior['x_0'], 'a_t': prior['a_0'], 'c_t': prior['c_0'], 'e_t': prior['e_0']}
        dst = None
        for s_idx in range(1, t.shape[0]):
            new, dst = self.step_endpoint(batch, state, t[s_idx], t[s_idx - 1], alpha_t[s_idx - 1], alpha_tp[s_idx - 1], prev=dst,
                                          inv_temp=inv_temp_func(t[s_idx - 1]))
            state = {k: new[k] for k in ('x_t', 'a_t', 'c_t', 'e_t')}
        return {'x_1': state['x_t'], 'a_1': state['a_t'], 'c_1': state['c_t'], 'e_1': state['e_t']}

    def sample_prior(self, batch: Batch, device='cpu'):
        """FlowMol.sample_prior, flowmol.py:417-448 (RNG use: one randn(N,3))."""
        return {
            'x_0': centered_normal_prior(batch, device),
            'a_0': ctmc_masked_prior(batch.N, self.na).to(device),
            'c_0': ctmc_masked_prior(batch.N, self.nc).to(device),
            'e_0': edge_prior(batch.upper_edge_mask, self.ne).to(device),
        }


# --------------------------------------------------------------------------------------
# result extraction  (molecule_builder.py:217-265 minus RDKit)
# --------------------------------------------------------------------------------------
def extract_moldata(x_1, a_1, c_1, e_1, n_atoms: int, atom_type_map: List[str], fake_atoms: bool,
                    n_bond_types: int = 4):
    """Per-molecule tensors in reference edge order -> (positions, symbols, charges, bond_types, bond_src, bond_dst).

    Fake atoms (type index len(atom_type_map)) are dropped and bonds re-indexed, as DGL
    ``remove_nodes`` does; masked bonds (index n_bond_types) count as no bond."""
    amap = list(atom_type_map) + (['Sn'] if fake_atoms else []) + ['Se']
    e = build_edge_idxs(n_atoms)
    u = n_atoms * (n_atoms - 1) // 2
    a_idx = a_1.argmax(dim=1)
    keep = torch.ones(n_atoms, dtype=torch.bool)
    if fake_atoms:
        keep = a_idx != len(atom_type_map)
    new_id = torch.cumsum(keep.long(), 0) - 1
    positions = x_1[keep]
    symbols = [amap[int(i)] for i in a_idx[keep]]
    charges = c_1.argmax(dim=1)[keep] - 2
    bt = e_1.argmax(dim=1)[:u].clone()
    bt[bt == n_bond_types] = 0
    src, dst = e[0, :u], e[1, :u]
    ok = keep[src] & keep[dst] & (bt != 0)
    return positions, symbols, charges, bt[ok], new_id[src[ok]], new_id[dst[ok]]


# --------------------------------------------------------------------------------------
# valence stability: flowmol/analysis/molecule_builder.py:138-157 and
# flowmol/analysis/metrics.py:333-363, on the output of extract_moldata
# --------------------------------------------------------------------------------------
def compute_valencies(n_atoms: int, bond_types, bond_src, bond_dst, arom_dependent: bool = False):
    """SampledMolecule.compute_valencies (molecule_builder.py:138-157)."""
    adj = torch.zeros((n_atoms, n_atoms)).float()
    bt = bond_types.clone().float()
    bt[bt == 4] = 1.5
    adj[bond_src, bond_dst] = bt
    adj[bond_dst, bond_src] = bt
    val = torch.sum(adj, dim=-1)
    if arom_dependent:
        n_arom = (adj == 1.5).sum(dim=-1)
        val = torch.stack([n_arom, (val - n_arom * 1.5).long()], dim=1)
    return val


def check_stability(atom_types: List[str], valencies, charges, valid_valency_table: dict, explicit_aromaticity: bool = False):
    """metrics.py:333-363 for a molecule whose fake atoms are already removed (extract_moldata drops them; the
    reference's 'Sn' branch therefore never fires on sampled molecules).  Returns (n_stable_atoms, mol_stable)."""
    n_stable = 0
    for atom_type, valency, charge in zip(atom_types, valencies.tolist(), charges):
        if not explicit_aromaticity:
            valency = int(valency)
        charge = int(charge)
        if atom_type not in valid_valency_table:
            continue                      # e.g. a surviving mask atom 'Se' (the reference would raise KeyError)
        by_charge = valid_valency_table[atom_type]
        if charge not in by_charge:
            continue
        if valency in by_charge[charge]:
            n_stable += 1
    return n_stable, n_stable == len(atom_types)


def bond_graph_components(n_atoms: int, bond_src, bond_dst):
    """Connected components of the bond graph: what Chem.GetMolFrags reports for the same bonds
    (metrics.py:172-186).  Returns (number of components, size of the largest)."""
    parent = list(range(n_atoms))

    def find(i):
        while parent[i] != i:
            parent[i] = parent[parent[i]]
            i = parent[i]
        return i
    for s, d in zip(bond_src.tolist(), bond_dst.tolist()):
        parent[find(s)] = find(d)
    sizes: Dict[int, int] = {}
    for i in range(n_atoms):
        r = find(i)
        sizes[r] = sizes.get(r, 0) + 1
    return len(sizes), (max(sizes.values()) if sizes else 0)
