"""Build-container-only tooling: import the reference's OWN sampling modules from
/root/reference with pure-torch stand-ins for the third-party packages that are not installed
(``dgl`` 2.0.0, ``torch_scatter`` 2.1.2).  Used by ``oracle/make_golden.py`` to generate the
golden vectors under ``tests/golden/`` and by ``tools/cpu_ref_vs_oracle_timing.py`` (all of them run in the build container only: /root/reference does not exist on the GPU
box; the tests compare against the committed fixtures, ``tests/test_oracle_golden.py``).  Nothing from the reference is copied; the
stand-ins implement only the documented semantics of the handful of DGL calls on the path
(SURVEY.md §2.2 K1,K3,K5,K10,K18): gather, segmented sum/mean, subtraction.

TEST INFRASTRUCTURE ONLY -- never imported by ``flowmol_amd``.
"""
from __future__ import annotations

import contextlib
import importlib
import sys
import types
from pathlib import Path

import torch

REFERENCE_ROOT = Path('/root/reference')


def reference_available() -> bool:
    return (REFERENCE_ROOT / 'flowmol' / 'models' / 'ctmc_vector_field.py').exists()


# ----------------------------------------------------------------------------- fake dgl
class _EdgeBatch:
    def __init__(self, g):
        self.data = g._edata
        self.src = {k: v[g._src] for k, v in g._ndata.items()}
        self.dst = {k: v[g._dst] for k, v in g._ndata.items()}


class FakeGraph:
    def __init__(self, src, dst, num_nodes, device=None, batch_num_nodes=None, batch_num_edges=None):
        dev = torch.device(device) if device is not None else src.device
        self._src = src.to(dev).long()
        self._dst = dst.to(dev).long()
        self._n = int(num_nodes)
        self._ndata, self._edata = {}, {}
        self._bnn = batch_num_nodes if batch_num_nodes is not None else torch.tensor([self._n])
        self._bne = batch_num_edges if batch_num_edges is not None else torch.tensor([self._src.shape[0]])
        self._bnn = self._bnn.to(dev)
        self._bne = self._bne.to(dev)

    ndata = property(lambda self: self._ndata)
    edata = property(lambda self: self._edata)
    device = property(lambda self: self._src.device)
    batch_size = property(lambda self: int(self._bnn.shape[0]))

    def num_nodes(self): return self._n
    def num_edges(self): return int(self._src.shape[0])
    def batch_num_nodes(self): return self._bnn
    def batch_num_edges(self): return self._bne
    def edges(self, form='uv'): return self._src, self._dst

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self._ndata), dict(self._edata)
        try:
            yield
        finally:
            self._ndata.clear(); self._ndata.update(nd)
            self._edata.clear(); self._edata.update(ed)

    def apply_edges(self, func):
        out = func(self) if getattr(func, '_builtin', False) else func(_EdgeBatch(self))
        self._edata.update(out)

    def update_all(self, msg, red):
        m = msg(self)
        self._ndata.update(red(self, m))

    def remove_nodes(self, nids):
        """dgl.DGLGraph.remove_nodes on an unbatched graph (documented semantics: the nodes and every edge incident to them
        are removed, the remaining nodes keep their relative order and are relabelled 0..n'-1, node/edge features are
        sliced accordingly, edge order is preserved)."""
        nids = torch.as_tensor(nids, device=self._src.device).long().reshape(-1)
        keep = torch.ones(self._n, dtype=torch.bool, device=self._src.device)
        keep[nids] = False
        new_id = torch.cumsum(keep.long(), 0) - 1
        ek = keep[self._src] & keep[self._dst]
        self._ndata = {k: v[keep] for k, v in self._ndata.items()}
        self._edata = {k: v[ek] for k, v in self._edata.items()}
        self._src, self._dst = new_id[self._src[ek]], new_id[self._dst[ek]]
        self._n = int(keep.sum())
        self._bnn = torch.tensor([self._n], device=self._src.device)
        self._bne = torch.tensor([int(ek.sum())], device=self._src.device)

    def to(self, device):
        g = FakeGraph(self._src, self._dst, self._n, device, self._bnn, self._bne)
        g._ndata = {k: v.to(device) for k, v in self._ndata.items()}
        g._edata = {k: v.to(device) for k, v in self._edata.items()}
        return g


def _builtin(f):
    f._builtin = True
    return f


def _u_sub_v(a, b, out):
    return _builtin(lambda g: {out: g._ndata[a][g._src] - g._ndata[b][g._dst]})


def _copy_e(e, out):
    return lambda g: {out: g._edata[e]}


def _reduce(kind):
    def make(m, out):
        def red(g, msgs):
            x = msgs[m]
            acc = torch.zeros((g._n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device).index_add_(0, g._dst, x)
            if kind == 'mean':
                deg = torch.zeros(g._n, dtype=x.dtype, device=x.device).index_add_(
                    0, g._dst, torch.ones(x.shape[0], dtype=x.dtype, device=x.device)).clamp(min=1)
                acc = acc / deg.view(-1, *([1] * (x.dim() - 1)))
            return {out: acc}
        return red
    return make


def _graph(data, num_nodes=None, device=None):
    src, dst = data
    return FakeGraph(torch.as_tensor(src), torch.as_tensor(dst), int(num_nodes), device)


def _batch(graphs):
    srcs, dsts, off = [], [], 0
    for g in graphs:
        srcs.append(g._src + off)
        dsts.append(g._dst + off)
        off += g._n
    bnn = torch.tensor([g._n for g in graphs])
    bne = torch.tensor([g.num_edges() for g in graphs])
    return FakeGraph(torch.cat(srcs), torch.cat(dsts), off, graphs[0].device, bnn, bne)


def _readout_nodes(g, feat, op='sum'):
    x = g._ndata[feat]
    B = g.batch_size
    idx = torch.arange(B, device=x.device).repeat_interleave(g._bnn)
    s = torch.zeros((B,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device).index_add_(0, idx, x)
    if op == 'mean':
        cnt = torch.zeros(B, dtype=x.dtype, device=x.device).index_add_(
            0, idx, torch.ones(x.shape[0], dtype=x.dtype, device=x.device))
        s = s / cnt.view(-1, *([1] * (x.dim() - 1)))
    return s


def _segment_csr(src, indptr, out=None, reduce='sum'):
    assert reduce == 'sum'
    cs = torch.cat([torch.zeros(1, dtype=src.dtype, device=src.device), src.cumsum(0)])
    return cs[indptr[1:]] - cs[indptr[:-1]]


def install_standins():
    if 'dgl' not in sys.modules:
        dgl = types.ModuleType('dgl')
        dgl.DGLGraph = FakeGraph
        dgl.graph = _graph
        dgl.batch = _batch
        dgl.readout_nodes = _readout_nodes
        fn = types.ModuleType('dgl.function')
        fn.u_sub_v, fn.copy_e, fn.sum, fn.mean = _u_sub_v, _copy_e, _reduce('sum'), _reduce('mean')
        dgl.function = fn
        nn_mod = types.ModuleType('dgl.nn')
        nnf = types.ModuleType('dgl.nn.functional')
        nnf.edge_softmax = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError('edge_softmax stand-in'))
        nn_mod.functional = nnf
        dgl.nn = nn_mod
        sys.modules.update({'dgl': dgl, 'dgl.function': fn, 'dgl.nn': nn_mod, 'dgl.nn.functional': nnf})
    if 'torch_scatter' not in sys.modules:
        ts = types.ModuleType('torch_scatter')
        ts.segment_csr = _segment_csr
        sys.modules['torch_scatter'] = ts


def import_reference():
    """Returns a namespace with the reference's sampling-path classes/functions."""
    if not reference_available():
        raise RuntimeError('/root/reference is not present')
    install_standins()
    if 'flowmol' not in sys.modules or not getattr(sys.modules['flowmol'], '_standin_pkg', False):
        pkg = types.ModuleType('flowmol')            # skip flowmol/__init__.py (needs Lightning/RDKit)
        pkg.__path__ = [str(REFERENCE_ROOT / 'flowmol')]
        pkg._standin_pkg = True
        sys.modules['flowmol'] = pkg
    ns = types.SimpleNamespace()
    ns.CTMCVectorField = importlib.import_module('flowmol.models.ctmc_vector_field').CTMCVectorField
    ns.InterpolantScheduler = importlib.import_module('flowmol.models.interpolant_scheduler').InterpolantScheduler
    gvp = importlib.import_module('flowmol.models.gvp')
    ns.GVP, ns.GVPConv, ns.GVPLayerNorm = gvp.GVP, gvp.GVPConv, gvp.GVPLayerNorm
    vf = importlib.import_module('flowmol.models.vector_field')
    ns.NodePositionUpdate, ns.EdgeUpdate = vf.NodePositionUpdate, vf.EdgeUpdate
    ns.EndpointVectorField = vf.EndpointVectorField
    ns.purity_sampling = importlib.import_module('flowmol.utils.ctmc_utils').purity_sampling
    emb = importlib.import_module('flowmol.utils.embedding')
    ns.get_time_embedding, ns._rbf = emb.get_time_embedding, emb._rbf
    du = importlib.import_module('flowmol.data_processing.utils')
    ns.build_edge_idxs, ns.get_upper_edge_mask, ns.get_batch_idxs = du.build_edge_idxs, du.get_upper_edge_mask, du.get_batch_idxs
    pr = importlib.import_module('flowmol.data_processing.priors')
    ns.ctmc_masked_prior, ns.edge_prior = pr.ctmc_masked_prior, pr.edge_prior
    ns.centered_normal_prior_batched_graph = pr.centered_normal_prior_batched_graph
    ns.inference_prior_register = pr.inference_prior_register
    ns.dgl = sys.modules['dgl']
    return ns


def build_reference_vf(ns, cfg, state_dict, **overrides):
    """Instantiate the reference CTMCVectorField for ``cfg`` and load ``state_dict`` (strict)."""
    sched = ns.InterpolantScheduler(canonical_feat_order=['x', 'a', 'c', 'e'],
                                    schedule_type=dict(getattr(cfg, 'schedule_type', None) or {k: 'linear' for k in 'xace'}),
                                    cosine_params=dict(getattr(cfg, 'cosine_params', None) or {}))
    common = {}
    if getattr(cfg, 'parameterization', 'ctmc') == 'endpoint':
        return _build_endpoint_vf(ns, cfg, state_dict, sched, overrides)
    vf = ns.CTMCVectorField(
        n_atom_types=cfg.n_atom_types, canonical_feat_order=['x', 'a', 'c', 'e'],
        interpolant_scheduler=sched, n_charges=cfg.n_charges, n_bond_types=cfg.n_bond_types,
        exclude_charges=False, fake_atoms=cfg.fake_atoms,
        self_conditioning=cfg.self_conditioning, stochasticity=cfg.stochasticity,
        high_confidence_threshold=cfg.high_confidence_threshold,
        n_vec_channels=cfg.n_vec_channels, update_edge_w_distance=cfg.update_edge_w_distance,
        n_hidden_scalars=cfg.n_hidden_scalars, n_hidden_edge_feats=cfg.n_hidden_edge_feats,
        n_recycles=cfg.n_recycles, separate_mol_updaters=cfg.separate_mol_updaters,
        n_molecule_updates=cfg.n_molecule_updates, convs_per_update=cfg.convs_per_update,
        n_cp_feats=cfg.n_cp_feats, n_message_gvps=cfg.n_message_gvps, n_update_gvps=cfg.n_update_gvps,
        message_norm=cfg.message_norm, rbf_dmax=cfg.rbf_dmax, rbf_dim=cfg.rbf_dim,
        time_embedding_dim=cfg.time_embedding_dim, a_token_dim=cfg.a_token_dim,
        c_token_dim=cfg.c_token_dim, e_token_dim=cfg.e_token_dim,
        use_dst_feats=cfg.use_dst_feats, dst_feat_msg_reduction_factor=cfg.dst_feat_msg_reduction_factor,
        **{'cat_temperature_schedule': cfg.cat_temperature, **overrides},
    )
    vf.load_state_dict(state_dict, strict=True)
    vf.eval()
    return vf


def build_reference_graph(ns, n_atoms: torch.Tensor, device='cpu'):
    """The graph-construction lines of reference FlowMol.sample (flowmol.py:509-529), which itself
    cannot be imported (Lightning/RDKit)."""
    graphs = []
    for n in n_atoms.tolist():
        e = ns.build_edge_idxs(n)
        graphs.append(ns.dgl.graph((e[0], e[1]), num_nodes=n, device=device))
    g = ns.dgl.batch(graphs)
    upper = ns.get_upper_edge_mask(g)
    nb, eb = ns.get_batch_idxs(g)
    return g, upper, nb, eb


def _build_endpoint_vf(ns, cfg, state_dict, sched, overrides):
    """The reference's EndpointVectorField (vector_field.py:15-211) for an endpoint-parameterised config, as FlowMol.__init__
    builds it (flowmol.py:137-153: n_atom_types incl. the fake atom, has_mask left False)."""
    vf = ns.EndpointVectorField(
        n_atom_types=cfg.n_atom_types, canonical_feat_order=['x', 'a', 'c', 'e'], interpolant_scheduler=sched,
        n_charges=cfg.n_charges, n_bond_types=cfg.n_bond_types, exclude_charges=False,
        self_conditioning=cfg.self_conditioning, n_vec_channels=cfg.n_vec_channels, update_edge_w_distance=cfg.update_edge_w_distance,
        n_hidden_scalars=cfg.n_hidden_scalars, n_hidden_edge_feats=cfg.n_hidden_edge_feats, n_recycles=cfg.n_recycles,
        separate_mol_updaters=cfg.separate_mol_updaters, n_molecule_updates=cfg.n_molecule_updates, convs_per_update=cfg.convs_per_update,
        n_cp_feats=cfg.n_cp_feats, n_message_gvps=cfg.n_message_gvps, n_update_gvps=cfg.n_update_gvps, message_norm=cfg.message_norm,
        rbf_dmax=cfg.rbf_dmax, rbf_dim=cfg.rbf_dim, time_embedding_dim=cfg.time_embedding_dim, a_token_dim=0, c_token_dim=0, e_token_dim=0,
        use_dst_feats=cfg.use_dst_feats, dst_feat_msg_reduction_factor=cfg.dst_feat_msg_reduction_factor,
        continuous_inv_temp_schedule=cfg.continuous_inv_temp_schedule, continuous_inv_temp_max=cfg.continuous_inv_temp_max, **overrides)
    vf.load_state_dict(state_dict, strict=True)
    vf.eval()
    return vf
