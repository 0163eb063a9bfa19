"""Generate tests/golden/*.npz by running the REFERENCE's own modules (imported from
/root/reference with the stand-ins of oracle/ref_standin.py) on seeded inputs.

Run in the build container only:   python -m oracle.make_golden
The fixtures hold inputs + expected outputs (data, kilobytes to a few MB); weights are not
stored -- they are re-created by name from ``flowmol_amd.weights.synth_state_dict(cfg, seed)``.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from flowmol_amd import presets, weights          # noqa: E402
from oracle import cpu_ref, ref_standin           # noqa: E402

OUT = ROOT / 'tests' / 'golden'


def _np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def _rand_state(cfg, g, upper, gen, frac_masked=0.5):
    """A mid-trajectory-like state: random tokens with a share of mask tokens."""
    N, E = g.num_nodes(), g.num_edges()
    U = E // 2

    def toks(n, k):
        t = torch.randint(0, k, (n,), generator=gen)
        t[torch.rand(n, generator=gen) < frac_masked] = k
        return t
    a, c, eu = toks(N, cfg.n_atom_types), toks(N, cfg.n_charges), toks(U, cfg.n_bond_types)
    x = torch.randn(N, 3, generator=gen) * 1.5
    oh = torch.nn.functional.one_hot
    e = torch.zeros(E, cfg.n_bond_types + 1)
    e[upper] = oh(eu, cfg.n_bond_types + 1).float()
    e[~upper] = oh(eu, cfg.n_bond_types + 1).float()
    return x, oh(a, cfg.n_atom_types + 1).float(), oh(c, cfg.n_charges + 1).float(), e, a, c, eu


def gen_forward(ns, name, cfg, sd):
    vf = ref_standin.build_reference_vf(ns, cfg, sd)
    n_atoms = torch.tensor([5, 9, 12, 3, 2])
    g, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    gen = torch.Generator().manual_seed(11)
    out = {'n_atoms': n_atoms}
    with torch.no_grad():
        # (1) t == 0 : bootstrap pass for self-conditioned models (2 network evaluations)
        x, a1h, c1h, e1h, a, c, eu = _rand_state(cfg, g, upper, gen, frac_masked=1.0)
        g.ndata['x_t'], g.ndata['a_t'], g.ndata['c_t'], g.edata['e_t'] = x, a1h, c1h, e1h
        d0 = vf(g, t=torch.zeros(g.batch_size), node_batch_idx=nb, upper_edge_mask=upper,
                apply_softmax=True, remove_com=True, prev_dst_dict=None)
        out.update({'t0.x_t': x, 't0.a': a, 't0.c': c, 't0.e_upper': eu})
        out.update({f't0.out.{k}': v for k, v in d0.items()})
        # (2) t = 0.5 with a previous endpoint
        x, a1h, c1h, e1h, a, c, eu = _rand_state(cfg, g, upper, gen, frac_masked=0.4)
        g.ndata['x_t'], g.ndata['a_t'], g.ndata['c_t'], g.edata['e_t'] = x, a1h, c1h, e1h
        prev = None
        if cfg.self_conditioning:
            prev = {'x': x + 0.3 * torch.randn(x.shape, generator=gen),
                    'a': torch.softmax(torch.randn(x.shape[0], cfg.n_atom_types, generator=gen), -1),
                    'c': torch.softmax(torch.randn(x.shape[0], cfg.n_charges, generator=gen), -1),
                    'e': torch.softmax(torch.randn(int(upper.sum()), cfg.n_bond_types, generator=gen), -1)}
            out.update({f'th.prev.{k}': v for k, v in prev.items()})
        d1 = vf(g, t=torch.full((g.batch_size,), 0.5), node_batch_idx=nb, upper_edge_mask=upper,
                apply_softmax=True, remove_com=True, prev_dst_dict=prev)
        out.update({'th.x_t': x, 'th.a': a, 'th.c': c, 'th.e_upper': eu})
        out.update({f'th.out.{k}': v for k, v in d1.items()})
    np.savez_compressed(OUT / f'forward_{name}.npz', **_np(out))


def gen_modules(ns, name, cfg, sd):
    """Module-level vectors from the reference's submodules: GVPConv 0, EdgeUpdate 1, NodePositionUpdate 1, SC layer."""
    vf = ref_standin.build_reference_vf(ns, cfg, sd)
    n_atoms = torch.tensor([6, 9])
    g, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    gen = torch.Generator().manual_seed(5)
    N, E = g.num_nodes(), g.num_edges()
    S, V, F_ = cfg.n_hidden_scalars, cfg.n_vec_channels, cfg.n_hidden_edge_feats
    s = torch.randn(N, S, generator=gen)
    v = torch.randn(N, V, 3, generator=gen) * 0.5
    x = torch.randn(N, 3, generator=gen) * 2
    ef = torch.randn(E, F_, generator=gen)
    out = {'n_atoms': n_atoms, 's': s, 'v': v, 'x': x, 'ef': ef}
    with torch.no_grad():
        g.ndata['x_t'] = x
        x_diff, d = vf.precompute_distances(g)
        out['x_diff'], out['d'] = x_diff, d
        s2, v2 = vf.conv_layers[0](g, scalar_feats=s, coord_feats=x, vec_feats=v, edge_feats=ef, x_diff=x_diff, d=d)
        out['conv0.s'], out['conv0.v'] = s2, v2
        out['pos1.x'] = vf.node_position_updaters[1](s, x, v)
        out['edge1.ef'] = vf.edge_updaters[1](g, s, ef, d=d)
        gvp0 = vf.conv_layers[0].edge_message[0]
        R = cfg.rbf_dim
        fs = torch.randn(7, S + R + F_ + cfg.s_dst_feats, generator=gen)
        fv = torch.randn(7, V + 1 + cfg.v_dst_feats, 3, generator=gen)
        o_s, o_v = gvp0((fs, fv))
        out['gvp0.in_s'], out['gvp0.in_v'], out['gvp0.out_s'], out['gvp0.out_v'] = fs, fv, o_s, o_v
        if cfg.self_conditioning:
            prev = {'x': x + 0.2 * torch.randn(x.shape, generator=gen),
                    'a': torch.softmax(torch.randn(N, cfg.n_atom_types, generator=gen), -1),
                    'c': torch.softmax(torch.randn(N, cfg.n_charges, generator=gen), -1),
                    'e': torch.softmax(torch.randn(E // 2, cfg.n_bond_types, generator=gen), -1)}
            efs = ef.clone()
            efs[~upper] = efs[upper]
            so, _, _, eo = vf.self_conditioning_residual_layer(g, s, x, v, efs, prev, nb, upper)
            out.update({f'sc.prev.{k}': t for k, t in prev.items()})
            out['sc.ef_in'], out['sc.s'], out['sc.ef'] = efs, so, eo
    np.savez_compressed(OUT / f'modules_{name}.npz', **_np(out))


class _Tape:
    """Record the RNG draws the reference makes during integrate, in order.

    ``Categorical.sample`` -> ``torch.multinomial(p, 1, True)`` draws its Exp(1) tensor inside ATen
    (q = empty_like(p).exponential_(); argmax(p/q)), invisible to Python; so ``torch.multinomial`` is
    wrapped: save the RNG state, draw the same-shaped exponential (recorded), restore the state and
    let the real multinomial consume the identical numbers.  ``torch.rand`` is wrapped directly."""
    def __init__(self):
        self.tape = []

    def __enter__(self):
        self._mn = torch.multinomial
        self._rand = torch.rand
        tape = self.tape
        mn_, rand_ = self._mn, self._rand

        def multinomial(p, num_samples, replacement=False, **k):
            assert num_samples == 1
            st = torch.get_rng_state()
            tape.append(torch.empty_like(p).exponential_(1).detach().clone())
            torch.set_rng_state(st)
            return mn_(p, num_samples, replacement, **k)

        def rand(*a, **k):
            r = rand_(*a, **k)
            tape.append(r.detach().clone())
            return r
        torch.multinomial = multinomial
        torch.rand = rand
        return self

    def __exit__(self, *exc):
        torch.multinomial = self._mn
        torch.rand = self._rand


def gen_integrate(ns, name, cfg, sd, sizes, T, tag):
    vf = ref_standin.build_reference_vf(ns, cfg, sd)
    n_atoms = torch.tensor(sizes)
    g, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    torch.manual_seed(1)
    x0 = ns.centered_normal_prior_batched_graph(g, nb)
    g.ndata['x_0'] = x0
    g.ndata['a_0'] = ns.ctmc_masked_prior(g.num_nodes(), cfg.n_atom_types)
    g.ndata['c_0'] = ns.ctmc_masked_prior(g.num_nodes(), cfg.n_charges)
    g.edata['e_0'] = ns.edge_prior(upper, {'type': 'ctmc', 'kwargs': {}}, explicit_aromaticity=cfg.explicit_aromaticity)
    torch.manual_seed(2)
    with torch.no_grad(), _Tape() as tp:
        gout, frames = vf.integrate(g, nb, upper_edge_mask=upper, n_timesteps=T, visualize=True,
                                    stochasticity=None, high_confidence_threshold=None)
    out = {'n_atoms': n_atoms, 'T': T, 'x_0': x0,
           'x_1': gout.ndata['x_1'], 'a_1': gout.ndata['a_1'].argmax(-1), 'c_1': gout.ndata['c_1'].argmax(-1),
           'e_1_upper': gout.edata['e_1'][upper].argmax(-1),
           'e_1_sym': torch.equal(gout.edata['e_1'][upper], gout.edata['e_1'][~upper])}
    for i, t in enumerate(tp.tape):
        out[f'noise.{i:05d}'] = t
    # per-step trajectories of molecule 0 and per-step norms of x_t for all
    out['traj0.x'] = frames[0]['x']
    out['traj0.a'] = frames[0]['a'].argmax(-1)
    out['traj0.x_1_pred'] = frames[0]['x_1_pred']
    np.savez_compressed(OUT / f'integrate_{name}_{tag}.npz', **_np(out))


# integrator variants: non-uniform tspan, 'decay' temperature schedule, an inverse-temperature
# function for the position step, and dfm_type='gat' with the 'beta' forward-weight schedule
VARIANT_TSPAN = [0.0, 0.04, 0.12, 0.25, 0.4, 0.55, 0.7, 0.82, 0.92, 0.97, 1.0]


def variant_inv_temp(t):
    return 1 - 0.25 * t


def gen_integrate_variant(ns, name, cfg, sd, sizes, tag, dfm_type):
    vf = ref_standin.build_reference_vf(ns, cfg, sd, cat_temperature_schedule='decay', cat_temp_decay_max=0.8,
                                        cat_temp_decay_a=2, forward_weight_schedule='beta')
    n_atoms = torch.tensor(sizes)
    g, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    torch.manual_seed(5)
    x0 = ns.centered_normal_prior_batched_graph(g, nb)
    g.ndata['x_0'] = x0
    g.ndata['a_0'] = ns.ctmc_masked_prior(g.num_nodes(), cfg.n_atom_types)
    g.ndata['c_0'] = ns.ctmc_masked_prior(g.num_nodes(), cfg.n_charges)
    g.edata['e_0'] = ns.edge_prior(upper, {'type': 'ctmc', 'kwargs': {}}, explicit_aromaticity=False)
    tspan = torch.tensor(VARIANT_TSPAN)
    torch.manual_seed(6)
    with torch.no_grad(), _Tape() as tp:
        gout, frames = vf.integrate(g, nb, upper_edge_mask=upper, n_timesteps=len(VARIANT_TSPAN), visualize=True,
                                    dfm_type=dfm_type, stochasticity=None, high_confidence_threshold=None,
                                    tspan=tspan, inv_temp_func=variant_inv_temp)
    out = {'n_atoms': n_atoms, 'tspan': tspan, 'x_0': x0,
           'x_1': gout.ndata['x_1'], 'a_1': gout.ndata['a_1'].argmax(-1), 'c_1': gout.ndata['c_1'].argmax(-1),
           'e_1_upper': gout.edata['e_1'][upper].argmax(-1)}
    for i, t in enumerate(tp.tape):
        out[f'noise.{i:05d}'] = t
    out['traj0.x'] = frames[0]['x']
    out['traj0.a'] = frames[0]['a'].argmax(-1)
    out['traj0.a_1_pred'] = frames[0]['a_1_pred'].argmax(-1)
    np.savez_compressed(OUT / f'integrate_{name}_{tag}.npz', **_np(out))


def gen_integrate_cosine(ns, cfg, sd, sizes):
    """Free-running reference trajectory under the COSINE interpolant schedule (interpolant_scheduler.py:131-146; the
    commented example of configs/flowmol3.yml:114-118): x cosine nu=1, a and c cosine nu=2, e linear -- incl. the reference's
    in-place clamp of t[0] to 1e-9, which removes the bootstrap evaluation."""
    import dataclasses
    ccfg = dataclasses.replace(cfg, schedule_type={'x': 'cosine', 'a': 'cosine', 'c': 'cosine', 'e': 'linear'},
                               cosine_params={'x': 1, 'a': 2, 'c': 2})
    vf = ref_standin.build_reference_vf(ns, ccfg, sd)
    n_atoms = torch.tensor(sizes)
    g, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    torch.manual_seed(8)
    x0 = ns.centered_normal_prior_batched_graph(g, nb)
    g.ndata['x_0'] = x0
    g.ndata['a_0'] = ns.ctmc_masked_prior(g.num_nodes(), cfg.n_atom_types)
    g.ndata['c_0'] = ns.ctmc_masked_prior(g.num_nodes(), cfg.n_charges)
    g.edata['e_0'] = ns.edge_prior(upper, {'type': 'ctmc', 'kwargs': {}}, explicit_aromaticity=False)
    T = 12
    torch.manual_seed(9)
    with torch.no_grad(), _Tape() as tp:
        gout, frames = vf.integrate(g, nb, upper_edge_mask=upper, n_timesteps=T, visualize=True,
                                    stochasticity=None, high_confidence_threshold=None)
    out = {'n_atoms': n_atoms, 'T': T, 'x_0': x0,
           'x_1': gout.ndata['x_1'], 'a_1': gout.ndata['a_1'].argmax(-1), 'c_1': gout.ndata['c_1'].argmax(-1),
           'e_1_upper': gout.edata['e_1'][upper].argmax(-1)}
    for i, t in enumerate(tp.tape):
        out[f'noise.{i:05d}'] = t
    out['traj0.x'] = frames[0]['x']
    out['traj0.a'] = frames[0]['a'].argmax(-1)
    tt = torch.linspace(0, 1, T)
    out['alpha.a'] = vf.interpolant_scheduler.alpha_t(tt)
    out['alpha.ap'] = vf.interpolant_scheduler.alpha_t_prime(tt)
    out['alpha.t_after'] = tt
    np.savez_compressed(OUT / 'integrate_qm9_cosine.npz', **_np(out))


def endpoint_prior(cfg, n_atoms, seed):
    """Priors of an endpoint-parameterised model drawn with the reference's own prior functions (priors.py:8-68,305-316)."""
    import importlib
    pr = importlib.import_module('flowmol.data_processing.priors')
    N = int(n_atoms.sum())
    U = int((n_atoms * (n_atoms - 1) // 2).sum())
    torch.manual_seed(seed)
    out = {}
    for k, rows, d in (('a', N, cfg.n_atom_types), ('c', N, cfg.n_charges), ('e', U, cfg.n_bond_types)):
        out[k] = pr.train_prior_register[cfg.prior_types[k]](rows, d, **cfg.prior_kwargs.get(k, {}))
    return out


def gen_integrate_endpoint(ns):
    """Free-running trajectory of the reference's EndpointVectorField (vector_field.py:388-569): Euler steps of x, a, c, e as
    continuous features, 'linear' inverse-temperature schedule of the vector field, priors from the reference's prior functions."""
    import dataclasses
    cfg = dataclasses.replace(presets.endpoint_small(), continuous_inv_temp_schedule='linear', continuous_inv_temp_max=1.5)
    sd = weights.synth_state_dict(cfg, 0)
    vf = ref_standin.build_reference_vf(ns, cfg, sd)
    n_atoms = torch.tensor([6, 3, 9, 2])
    g, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    torch.manual_seed(21)
    x0 = ns.centered_normal_prior_batched_graph(g, nb)
    pri = endpoint_prior(cfg, n_atoms, 22)
    e0 = torch.zeros(g.num_edges(), cfg.n_bond_types)
    e0[upper] = pri['e']; e0[~upper] = pri['e']
    g.ndata['x_0'], g.ndata['a_0'], g.ndata['c_0'], g.edata['e_0'] = x0, pri['a'], pri['c'], e0
    T = 9
    with torch.no_grad():
        gout, frames = vf.integrate(g, nb, upper_edge_mask=upper, n_timesteps=T, visualize=True)
        # one forward pass of the prior state on its own (the dense-embedding path of the network)
        g2, _, _, _ = ref_standin.build_reference_graph(ns, n_atoms)
        g2.ndata['x_t'], g2.ndata['a_t'], g2.ndata['c_t'], g2.edata['e_t'] = x0, pri['a'], pri['c'], e0
        d0 = vf(g2, t=torch.full((4,), 0.25), node_batch_idx=nb, upper_edge_mask=upper, apply_softmax=True, remove_com=True, prev_dst_dict=None)
    out = {'n_atoms': n_atoms, 'T': T, 'x_0': x0, 'a_0': pri['a'], 'c_0': pri['c'], 'e_0_upper': pri['e'],
           'x_1': gout.ndata['x_1'], 'a_1': gout.ndata['a_1'], 'c_1': gout.ndata['c_1'], 'e_1_upper': gout.edata['e_1'][upper],
           'e_1_sym': torch.equal(gout.edata['e_1'][upper], gout.edata['e_1'][~upper]),
           'traj0.x': frames[0]['x'], 'traj0.a': frames[0]['a'], 'traj0.x_1_pred': frames[0]['x_1_pred']}
    out.update({f'fwd.{k}': v for k, v in d0.items()})
    np.savez_compressed(OUT / 'integrate_endpoint.npz', **_np(out))


def _ref_function(path, name, cls=None):
    """Source-level import of ONE function (or method of class ``cls``) of a reference module that cannot be
    imported here as a whole (rdkit at module scope): parsed with ast, compiled and executed in this container
    only, to produce fixtures."""
    import ast
    tree = ast.parse(Path(path).read_text())
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fn = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    fn.decorator_list = []
    for a in fn.args.args + fn.args.kwonlyargs:
        a.annotation = None
    fn.returns = None
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    env = {'torch': torch}
    exec(compile(mod, str(path), 'exec'), env)
    return env[name]


def gen_stability():
    """Random token molecules -> the reference's own compute_valencies + check_stability verdicts."""
    import json
    from types import SimpleNamespace
    ref = Path('/root/reference/flowmol')
    check = _ref_function(ref / 'analysis' / 'metrics.py', 'check_stability')
    comp_val = _ref_function(ref / 'analysis' / 'molecule_builder.py', 'compute_valencies', cls='SampledMolecule')
    out = {}
    gen = torch.Generator().manual_seed(11)
    for tag, dataset, arom, fake in (('kek', 'geom_full_kekulized', False, True), ('arom', 'geom_5_aromatic', True, False)):
        fn = 'train_data_valencies_aromatic.json' if arom else 'train_data_valencies_kekulized.json'
        raw = json.loads((Path('/root/reference/data') / dataset / fn).read_text())
        table = {a: {int(c): v for c, v in cs.items()} for a, cs in raw.items()}      # metrics.py:76-79
        atom_map = ['C', 'H', 'N', 'O', 'F', 'P', 'S', 'Cl', 'Br', 'I']
        nb = 5 if arom else 4
        sizes = [2, 5, 9, 14, 23, 31, 6, 7]
        A, C_, E, res = [], [], [], []
        for n in sizes:
            u = n * (n - 1) // 2
            if n == 6:      # methane + one extra atom (the fake atom, bonded to C, where the model has one; else a lone F-)
                a = torch.tensor([0, 1, 1, 1, 1, 10 if fake else 4])
                c = torch.tensor([2, 2, 2, 2, 2, 2 if fake else 1])
                e = torch.zeros(u, dtype=torch.long)
                e[0:4] = 1
                e[4] = 1 if fake else nb         # C-fake single bond (dropped with the atom) / a masked pair
            elif n == 7:    # ammonium + hydroxide: two fragments, every atom stable
                a = torch.tensor([2, 1, 1, 1, 1, 3, 1]) if not arom else torch.tensor([2, 1, 1, 1, 1, 3, 1])
                c = torch.tensor([3, 2, 2, 2, 2, 1, 2])
                e = torch.zeros(u, dtype=torch.long)
                e[0:4] = 1                       # N-H x4
                e[5] = 0                         # N-O none
                pair = lambda i, j: i * (2 * n - i - 1) // 2 + (j - i - 1)
                e[pair(5, 6)] = 1; e[pair(4, 5)] = 0
            else:
                # chemistry-like sparsity so that a fair share of atoms is stable
                a = torch.multinomial(torch.tensor([4., 5, 1, 1, .3, .1, .2, .2, .1, .1] + ([1.0] if fake else [])), n, True, generator=gen)
                c = torch.multinomial(torch.tensor([.02, .05, 1., .08, .02, .01]), n, True, generator=gen)
                pe = torch.tensor([1 - 2.2 / n if n > 3 else .3, 2.0 / n, .25 / n, .05 / n] + ([.5 / n] if arom else []) + [.1 / n]).clamp(min=1e-3)
                e = torch.multinomial(pe, u, True, generator=gen)       # last index = mask token
            A.append(a); C_.append(c); E.append(e)
            x = torch.zeros(n, 3)
            oh = lambda t, k: torch.nn.functional.one_hot(t, k).float()
            pos, sym, chg, bt, bs, bd = cpu_ref.extract_moldata(x, oh(a, len(atom_map) + (2 if fake else 1)), oh(c, 6),
                                                                torch.cat([oh(e, nb + 1)] * 2), n, atom_map, fake, nb)
            mol = SimpleNamespace(num_atoms=len(sym), bond_types=bt, bond_src_idxs=bs, bond_dst_idxs=bd, atom_types=sym,
                                  atom_charges=chg, fake_atoms=fake)
            mol.valencies = comp_val(mol, arom_dependent=arom)
            n_stable, mol_stable, n_fake = check(mol, table, explicit_aromaticity=arom)
            res.append([n_stable, int(mol_stable), len(sym)])
        out[f'{tag}.n_atoms'] = torch.tensor(sizes)
        out[f'{tag}.a'] = torch.cat(A); out[f'{tag}.c'] = torch.cat(C_); out[f'{tag}.e'] = torch.cat(E)
        out[f'{tag}.expect'] = torch.tensor(res)            # n_stable_atoms, mol_stable, real atoms
    np.savez_compressed(OUT / 'stability.npz', **_np(out))


def gen_moldata():
    """Token molecules -> the reference's OWN extract_moldata_from_graph (molecule_builder.py:217-265) run on a per-molecule
    graph carrying the float one-hots the reference keeps (x_1, a_1, c_1, e_1, ue_mask); the function is AST-imported like
    check_stability because its module imports RDKit.  Pins cpu_ref.extract_moldata and flowmol_amd.molecule."""
    ref = Path('/root/reference/flowmol')
    extract = _ref_function(ref / 'analysis' / 'molecule_builder.py', 'extract_moldata_from_graph')
    ns = ref_standin.import_reference()
    g_st = {k: torch.from_numpy(v) for k, v in np.load(OUT / 'stability.npz').items()}
    oh = lambda t, k: torch.nn.functional.one_hot(t, k).float()
    out = {}
    gen = torch.Generator().manual_seed(23)
    base_map = ['C', 'H', 'N', 'O', 'F', 'P', 'S', 'Cl', 'Br', 'I']
    for tag, arom, fake in (('kek', False, True), ('arom', True, False)):
        nb = 5 if arom else 4
        amap = base_map + (['Sn'] if fake else []) + ['Se']            # what SampledMolecule.__init__ passes (molecule_builder.py:40-44)
        sizes = g_st[f'{tag}.n_atoms'].tolist()
        X, P, SYM, CHG, BT, BS, BD, cnt = [], [], [], [], [], [], [], []
        no = po = 0
        for n in sizes:
            u = n * (n - 1) // 2
            a, c, e = g_st[f'{tag}.a'][no:no + n], g_st[f'{tag}.c'][no:no + n], g_st[f'{tag}.e'][po:po + u]
            no += n; po += u
            x = torch.randn(n, 3, generator=gen)
            ei = ns.build_edge_idxs(n)
            g = ns.dgl.graph((ei[0], ei[1]), num_nodes=n)
            g.ndata['x_1'], g.ndata['a_1'], g.ndata['c_1'] = x, oh(a, len(amap)), oh(c, 6)
            g.edata['e_1'] = torch.cat([oh(e, nb + 1)] * 2)
            g.edata['ue_mask'] = ns.get_upper_edge_mask(g)
            pos, sym, chg, bt, bs, bd = extract(g, amap, exclude_charges=False, ctmc_mol=True, fake_atoms=fake,
                                                show_fake_atoms=False, explicit_aromaticity=arom)
            X.append(x); P.append(pos); SYM.append(torch.tensor([amap.index(s_) for s_ in sym], dtype=torch.long))
            CHG.append(chg); BT.append(bt); BS.append(bs); BD.append(bd)
            cnt.append([pos.shape[0], bt.shape[0]])
        out[f'{tag}.x'] = torch.cat(X)
        out[f'{tag}.pos'] = torch.cat(P); out[f'{tag}.sym'] = torch.cat(SYM); out[f'{tag}.chg'] = torch.cat(CHG)
        out[f'{tag}.bt'] = torch.cat(BT); out[f'{tag}.bs'] = torch.cat(BS); out[f'{tag}.bd'] = torch.cat(BD)
        out[f'{tag}.counts'] = torch.tensor(cnt)                       # per molecule: atoms kept, bonds kept
    np.savez_compressed(OUT / 'moldata.npz', **_np(out))


def gen_misc(ns):
    out = {}
    t = torch.tensor([0.0, 0.004016064, 0.5, 0.9959839, 1.0])
    out['temb.t'] = t
    out['temb.out'] = ns.get_time_embedding(t, embedding_dim=64)
    d = torch.tensor([0.0, 1e-4, 0.7, 1.5, 3.3, 9.99, 14.0])
    out['rbf.d'] = d
    out['rbf.out10'] = ns._rbf(d, D_max=10, D_count=32)
    out['rbf.out12'] = ns._rbf(d, D_max=12, D_count=32)
    for n in (2, 3, 7):
        out[f'edges.{n}'] = ns.build_edge_idxs(n)
    sched = ns.InterpolantScheduler(canonical_feat_order=['x', 'a', 'c', 'e'], schedule_type={k: 'linear' for k in 'xace'})
    tt = torch.linspace(0, 1, 9)
    out['alpha.t'] = tt
    out['alpha.a'] = sched.alpha_t(tt)
    out['alpha.ap'] = sched.alpha_t_prime(tt)
    # purity sampling + campbell step through the reference's CTMCVectorField methods
    cfg = presets.flowmol3()
    vf = ref_standin.build_reference_vf(ns, cfg, weights.synth_state_dict(cfg, 0))
    gen = torch.Generator().manual_seed(3)
    sizes = torch.tensor([4, 7, 1, 9, 5])           # rows per "molecule"
    rows = int(sizes.sum())
    bidx = torch.arange(5).repeat_interleave(sizes)
    K = 11
    p = torch.softmax(torch.randn(rows, K, generator=gen) * 3, -1)
    p[4:11] = torch.softmax(torch.randn(7, K, generator=gen) * 0.3, -1)   # molecule 1: no high-confidence rows (h=0)
    p[12:21] = torch.softmax(torch.randn(9, K, generator=gen) * 30, -1)   # molecule 3: all high-confidence (m=h)
    xt = torch.full((rows,), K)
    xt[torch.rand(rows, generator=gen) < 0.3] = 2
    xt[12:21] = K
    for case, (hc, last, eta, alpha) in enumerate([(0.9, False, 30.0, 0.3), (0.9, True, 30.0, 0.996), (0.0, False, 10.0, 0.5),
                                                   (0.9, False, 30.0, 0.05)]):
        torch.manual_seed(100 + case)
        with _Tape() as tp:
            xt_new, x1 = vf.campbell_step(p_1_given_t=p, xt=xt.clone(), stochasticity=eta, hc_thresh=hc,
                                          alpha_t=torch.tensor(alpha), alpha_t_prime=torch.tensor(1.0),
                                          dt=torch.tensor(1.0 / 249), batch_size=5, batch_num_nodes=sizes,
                                          n_classes=K + 1, mask_index=K, last_step=last, batch_idx=bidx)
        out[f'ctmc.{case}.params'] = np.array([hc, float(last), eta, alpha, 1.0 / 249])
        out[f'ctmc.{case}.xt_new'] = xt_new.argmax(-1)
        out[f'ctmc.{case}.x1'] = x1.argmax(-1)
        for i, t_ in enumerate(tp.tape):
            out[f'ctmc.{case}.noise{i}'] = t_
    out['ctmc.p'], out['ctmc.xt'], out['ctmc.sizes'] = p, xt, sizes
    np.savez_compressed(OUT / 'misc.npz', **_np(out))


def gen_ctmc_step(ns):
    """The reference's own CTMCVectorField.step (ctmc_vector_field.py:287-411: Euler step, tempering softmax(log p / T),
    campbell_step, purity_sampling, edge mirroring) with its network evaluation replaced by a FIXED endpoint prediction, on a
    batch crafted to hit the purity-sampling edge cases by construction: a molecule with no high-confidence masked rows (h = 0),
    one whose masked rows are all high-confidence (m = h), a fully unmasked molecule (m = 0), a 2-atom molecule (one pair),
    the hc = 0 branch and the last step.  Inputs, the recorded RNG draws and the outputs go to tests/golden/ctmc_step.npz;
    the HIP fm_ctmc_step must reproduce the outputs bit for bit (tests/test_gpu_parity.py)."""
    cfg = presets.flowmol3()
    vf = ref_standin.build_reference_vf(ns, cfg, weights.synth_state_dict(cfg, 0))
    n_atoms = torch.tensor([4, 7, 2, 9, 5])
    g0, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    N, E = g0.num_nodes(), g0.num_edges()
    U = E // 2
    pairs = n_atoms * (n_atoms - 1) // 2
    pb = torch.arange(5).repeat_interleave(pairs)
    gen = torch.Generator().manual_seed(17)
    oh = torch.nn.functional.one_hot
    out = {'n_atoms': n_atoms}
    T = 250
    t = torch.linspace(0, 1, T)
    sched = ns.InterpolantScheduler(canonical_feat_order=['x', 'a', 'c', 'e'], schedule_type={k: 'linear' for k in 'xace'})
    alpha_t, alpha_tp = sched.alpha_t(t), sched.alpha_t_prime(t)

    def probs(rows, K, idx):
        p = torch.softmax(torch.randn(rows, K, generator=gen) * 3, -1)
        flat = torch.softmax(torch.randn(rows, K, generator=gen) * 0.05, -1)     # tempered max prob stays far below 0.9
        sharp = torch.softmax(torch.randn(rows, K, generator=gen) * 40, -1)      # tempered max prob ~ 1
        p[idx == 1] = flat[idx == 1]
        p[idx == 3] = sharp[idx == 3]
        return p

    cases = [(0.9, False, 30.0, 75), (0.9, True, 30.0, T - 1), (0.0, False, 10.0, 120), (0.9, False, 30.0, 12), (0.5, False, 0.0, 200)]
    for case, (hc, last, eta, s_idx) in enumerate(cases):
        def toks(rows, K, idx, frac):
            tk = torch.randint(0, K, (rows,), generator=gen)
            tk[torch.rand(rows, generator=gen) < frac] = K
            tk[idx == 3] = K                                    # molecule 3: every row masked  -> m = h
            tk[idx == 4] = tk[idx == 4].clamp(max=K - 1)        # molecule 4: nothing masked    -> m = 0, h = 0
            if case == 3:
                tk[idx == 1] = K                                # all masked, none high-confidence -> h = 0 with m > 0
            return tk
        a, c = toks(N, cfg.n_atom_types, nb, 0.5), toks(N, cfg.n_charges, nb, 0.5)
        eu = toks(U, cfg.n_bond_types, pb, 0.5)
        x_t = torch.randn(N, 3, generator=gen)
        dst = {'x': torch.randn(N, 3, generator=gen), 'a': probs(N, cfg.n_atom_types, nb), 'c': probs(N, cfg.n_charges, nb),
               'e': probs(U, cfg.n_bond_types, pb)}
        if case == 0:
            dst['a'][0, 3] = 0.0                                 # an exact zero probability: log -> -inf -> 0
            dst['a'][0] = dst['a'][0] / dst['a'][0].sum()
        g, _, _, _ = ref_standin.build_reference_graph(ns, n_atoms)
        e1h = torch.zeros(E, cfg.n_bond_types + 1)
        e1h[upper] = oh(eu, cfg.n_bond_types + 1).float(); e1h[~upper] = oh(eu, cfg.n_bond_types + 1).float()
        g.ndata['x_t'], g.ndata['a_t'], g.ndata['c_t'], g.edata['e_t'] = x_t.clone(), oh(a, cfg.n_atom_types + 1).float(), oh(c, cfg.n_charges + 1).float(), e1h
        vf.forward = lambda *a_, **k_: {k: v.clone() for k, v in dst.items()}       # the network evaluation is not under test here
        torch.manual_seed(300 + case)
        with torch.no_grad(), _Tape() as tp:
            gout, _ = vf.step(g, t[s_idx], t[s_idx - 1], alpha_t[s_idx - 1], alpha_t[s_idx], alpha_tp[s_idx - 1], nb, eb, upper,
                              cat_temp_func=vf.cat_temp_func, forward_weight_func=vf.forward_weight_func, prev_dst_dict=None,
                              dfm_type='campbell', stochasticity=eta, high_confidence_threshold=hc, last_step=last)
        del vf.forward
        pre = f'{case}.'
        out[pre + 'params'] = np.array([hc, float(last), eta, float(s_idx), float(T)])
        out.update({pre + 'x_t': x_t, pre + 'a_t': a, pre + 'c_t': c, pre + 'e_t': eu})
        out.update({pre + f'dst.{k}': v for k, v in dst.items()})
        out.update({pre + 'x_new': gout.ndata['x_t'], pre + 'a_new': gout.ndata['a_t'].argmax(-1), pre + 'c_new': gout.ndata['c_t'].argmax(-1),
                    pre + 'e_new': gout.edata['e_t'][upper].argmax(-1), pre + 'a_1_pred': gout.ndata['a_1_pred'].argmax(-1),
                    pre + 'c_1_pred': gout.ndata['c_1_pred'].argmax(-1), pre + 'e_1_pred': gout.edata['e_1_pred'][upper].argmax(-1)})
        assert torch.equal(gout.edata['e_t'][upper], gout.edata['e_t'][~upper])
        for i, t_ in enumerate(tp.tape):
            out[pre + f'noise{i}'] = t_
    np.savez_compressed(OUT / 'ctmc_step.npz', **_np(out))


# ---------------------------------------------------------------------------------------------------------------
# Long-horizon free-running trajectories at the product's DEFAULT protocol (test.py:25 / flowmol.py:46: n_timesteps = 250;
# BASELINE config C5: 500).  The noise tape is NOT stored (it would be ~100 MB): the fixture stores the two seeds, and the
# test side re-draws the tape from torch's CPU generator in the reference's order (flowmol_amd.engine.StepNoise.draw,
# pinned by tests/test_host_logic.py::test_step_noise_draw_order_matches_reference_rng_stream).  This function records the
# reference's real draws while it runs and asserts that the seeded re-draw reproduces every one of them bit for bit.
LONG_CASES = {       # tag: (preset, sizes, T, weight scale, prior seed, noise seed)
    'flowmol3_47x8_T250': ('flowmol3', [47] * 8, 250, 1.0, 31, 32),
    # weights x2: the endpoint prediction moves every atom by 3-8 % of the coordinate scale per evaluation (x1: < 0.1 %), so the
    # final coordinates really depend on the arithmetic of 250 network evaluations, while a 1-ulp perturbation of x_0 still
    # only moves the result by ~1e-6 (x2.5 and above are chaotic: the reference disagrees with ITSELF under such a perturbation)
    'flowmol3_mixed_T250_w2': ('flowmol3', [5, 33, 60, 90], 250, 2.0, 33, 34),
    'geom_ctmc_mixed_T500': ('geom_ctmc', [5, 17, 8, 30, 44, 60], 500, 1.0, 35, 36),
    # 64 molecules with sizes drawn from the shipped GEOM-drugs histogram (torch.multinomial, seed 64: 19..97 atoms, mean 47.3): 9.3 M tempered
    # categorical decisions per modality set over the 249 steps -- the flip-rate bound of the f32 kernels and of the opt-in split precision
    'flowmol3_geom64_T250': ('flowmol3', [43, 44, 54, 41, 68, 40, 52, 58, 28, 45, 19, 46, 34, 42, 65, 46, 39, 56, 50, 53, 53, 75, 31, 51, 40, 35, 48, 52, 53,
                                          51, 33, 41, 57, 44, 51, 57, 97, 53, 24, 36, 42, 66, 46, 47, 37, 61, 36, 57, 41, 33, 32, 65, 55, 50, 38, 47, 46, 51,
                                          45, 39, 42, 49, 53, 47], 250, 1.0, 37, 38),
    # 16 molecules with sizes from the same histogram (seed 16: 29..66 atoms), unit weights except the position heads (Wu of the last GVP of every
    # NodePositionUpdate x 128): the endpoint prediction moves every atom by ~4.6 % of the coordinate scale per evaluation, so x_rel over 250 steps is a
    # strong statistic for 16 molecules (the unit-weight fixtures move 0.04 %: VERDICT r4 weak #3), on a well-conditioned trajectory (1-ulp test: 2e-7)
    'flowmol3_geom16_T250_pos128': ('flowmol3', [41, 45, 57, 43, 54, 58, 46, 42, 29, 38, 44, 53, 66, 46, 48, 32], 250, 1.0, 39, 40, 128.0),
    # the same 16 sizes with the categorical output heads' last Linear x 256 (weights.scaled_weights: cat_head_scale): head logit gaps of tens instead of
    # < 1, i.e. the regime of a TRAINED model -- most classes of the tempered distribution exactly 0, rows with p == 1.0, exact zeros (log 0 = -inf) and
    # denormals in p, near-one-hot self-conditioning inputs -- over a free-running 250-step trajectory (measured with
    # the oracle on two GEOM-sized molecules: p == 1 in 16-100 % of the rows, exact zeros in up to 50 % of the edge probabilities, denormals in up to 29 %)
    'flowmol3_geom16_T250_heads256': ('flowmol3', [41, 45, 57, 43, 54, 58, 46, 42, 29, 38, 44, 53, 66, 46, 48, 32], 250, 1.0, 43, 44, 1.0, 256.0),
}
LONG_X_STRIDE = 10


def gen_integrate_long(ns, tag):
    from flowmol_amd.engine import StepNoise
    name, sizes, T, scale, seed_prior, seed_noise = LONG_CASES[tag][:6]
    pos_scale = LONG_CASES[tag][6] if len(LONG_CASES[tag]) > 6 else 1.0
    head_scale = LONG_CASES[tag][7] if len(LONG_CASES[tag]) > 7 else 1.0
    cfg = presets.PRESETS[name]()
    sd = weights.scaled_weights(weights.synth_state_dict(cfg, 0), scale, pos_scale, head_scale)
    vf = ref_standin.build_reference_vf(ns, cfg, sd)
    n_atoms = torch.tensor(sizes)
    g, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    N, U = g.num_nodes(), g.num_edges() // 2
    torch.manual_seed(seed_prior)
    x0 = ns.centered_normal_prior_batched_graph(g, nb)
    g.ndata['x_0'] = x0
    g.ndata['a_0'] = ns.ctmc_masked_prior(N, cfg.n_atom_types)
    g.ndata['c_0'] = ns.ctmc_masked_prior(N, cfg.n_charges)
    g.edata['e_0'] = ns.edge_prior(upper, {'type': 'ctmc', 'kwargs': {}}, explicit_aromaticity=False)
    torch.manual_seed(seed_noise)
    with torch.no_grad(), _Tape() as tp:
        gout, frames = vf.integrate(g, nb, upper_edge_mask=upper, n_timesteps=T, visualize=True,
                                    stochasticity=None, high_confidence_threshold=None)
    # the seeded re-draw the tests use == what the reference drew
    torch.manual_seed(seed_noise)
    pos = 0
    for step in range(T - 1):
        nz = StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, step == T - 2, 'cpu')
        for m in 'ace':
            for part in ('q', 'u1', 'u2'):
                t_ = getattr(nz, f'{part}_{m}')
                if t_ is None:
                    continue
                assert torch.equal(t_, tp.tape[pos]), (tag, step, m, part)
                pos += 1
    assert pos == len(tp.tape)
    B = len(sizes)
    u8 = lambda t_: t_.to(torch.uint8)
    cat = lambda key, half=False: torch.cat([(f[key][:, :f[key].shape[1] // 2] if half else f[key]).argmax(-1) for f in frames], dim=1)
    out = {'n_atoms': n_atoms, 'T': T, 'weight_scale': scale, **({'pos_head_scale': pos_scale} if pos_scale != 1 else {}), **({'cat_head_scale': head_scale} if head_scale != 1 else {}), 'seed_prior': seed_prior, 'seed_noise': seed_noise, 'x_0': x0,
           'x_1': gout.ndata['x_1'], 'a_1': gout.ndata['a_1'].argmax(-1), 'c_1': gout.ndata['c_1'].argmax(-1),
           'e_1_upper': gout.edata['e_1'][upper].argmax(-1),
           'e_1_sym': torch.equal(gout.edata['e_1'][upper], gout.edata['e_1'][~upper]),
           # state after every step (frame 0 = prior): tokens in full, coordinates as per-molecule norms + every 10th frame
           'traj.a': u8(cat('a')), 'traj.c': u8(cat('c')), 'traj.e': u8(cat('e', True)),
           'traj.a1': u8(cat('a_1_pred')), 'traj.c1': u8(cat('c_1_pred')), 'traj.e1': u8(cat('e_1_pred', True)),
           'traj.x_norm': torch.stack([f['x'].flatten(1).norm(dim=1) for f in frames], dim=1),
           'traj.x1_norm': torch.stack([f['x_1_pred'].flatten(1).norm(dim=1) for f in frames], dim=1),
           'traj.x_stride': LONG_X_STRIDE,
           'traj.x': torch.cat([f['x'][::LONG_X_STRIDE] for f in frames], dim=1)}
    assert out['traj.a'].shape == (T, N) and out['traj.e'].shape == (T, U) and out['traj.e1'].shape == (T - 1, U)
    assert out['traj.x_norm'].shape == (T, B)
    np.savez_compressed(OUT / f'long_{tag}.npz', **_np(out))


def gen_traj_frames(ns):
    """The reference's `traj_frames` of a visualised run IN ITS OWN FORMAT (ctmc_vector_field.py:188-202,235-255,267-283): per molecule a dict
    whose keys x, a, c, e hold T frames (frame 0 = the prior) and x_1_pred, a_1_pred, c_1_pred, e_1_pred hold T - 1, categorical frames as float
    one-hots incl. the mask column, edge frames over ALL directed edges (upper triangle, then the same pairs swapped).  Pins
    SampledMolecule.traj_frames_reference(), which rebuilds exactly these tensors from the compact token frames (VERDICT r4 missing #2).
    qm9 model, molecules of 5 / 3 / 6 atoms, n_timesteps = 6, default protocol; recorded noise."""
    cfg = presets.qm9(); sd = weights.synth_state_dict(cfg, 0)
    vf = ref_standin.build_reference_vf(ns, cfg, sd)
    n_atoms = torch.tensor([5, 3, 6])
    T = 6
    g, upper, nb, eb = ref_standin.build_reference_graph(ns, n_atoms)
    torch.manual_seed(21)
    x0 = ns.centered_normal_prior_batched_graph(g, nb)
    g.ndata['x_0'] = x0
    g.ndata['a_0'] = ns.ctmc_masked_prior(g.num_nodes(), cfg.n_atom_types)
    g.ndata['c_0'] = ns.ctmc_masked_prior(g.num_nodes(), cfg.n_charges)
    g.edata['e_0'] = ns.edge_prior(upper, {'type': 'ctmc', 'kwargs': {}}, explicit_aromaticity=False)
    torch.manual_seed(22)
    with torch.no_grad(), _Tape() as tp:
        gout, frames = vf.integrate(g, nb, upper_edge_mask=upper, n_timesteps=T, visualize=True, stochasticity=None, high_confidence_threshold=None)
    out = {'n_atoms': n_atoms, 'T': T, 'x_0': x0, 'x_1': gout.ndata['x_1'], 'a_1': gout.ndata['a_1'].argmax(-1)}
    for i, t in enumerate(tp.tape):
        out[f'noise.{i:05d}'] = t
    for m, fr in enumerate(frames):
        assert sorted(fr) == sorted(['x', 'a', 'c', 'e', 'x_1_pred', 'a_1_pred', 'c_1_pred', 'e_1_pred'])
        for k, v in fr.items():
            out[f'mol{m}.{k}'] = v
    np.savez_compressed(OUT / 'traj_frames.npz', **_np(out))


PRIOR_CASES = [          # (kind, n, d, kwargs): every categorical prior FlowMol.sample_prior can dispatch to (priors.py:253-262)
    ('gaussian', 7, 5, {'std': 0.7, 'simplex_center': True}),
    ('uniform-simplex', 6, 4, {}),
    ('barycenter', 5, 6, {}),
    ('barycenter', 9, 6, {'blur': 0.3}),
    ('biased-simplex', 8, 5, {'vertex_prob': 0.6, 'std': 0.25, 'vertex_idx': 2}),
    ('marginal', 11, 5, {'p': [0.5, 0.2, 0.1, 0.15, 0.05]}),
    ('marginal', 11, 5, {'p': [0.5, 0.2, 0.1, 0.15, 0.05], 'blur': 0.2}),
    ('c-given-a', 10, 6, {'blur': 0.1}),
    ('c-given-a', 10, 6, {}),
]


def gen_priors(ns):
    """Outputs of the reference's own prior functions under torch.manual_seed(100 + case) on the CPU generator."""
    out = {}
    p_ca = torch.softmax(torch.randn(4, 6, generator=torch.Generator().manual_seed(5)) * 2, -1)
    a_0 = torch.nn.functional.one_hot(torch.tensor([0, 3, 1, 2, 2, 0, 1, 3, 3, 0]), 4).float()
    out['p_c_given_a'], out['a_0'] = p_ca, a_0
    for i, (kind, n, d, kw) in enumerate(PRIOR_CASES):
        fn = ns.inference_prior_register[kind]
        kw = {k: (torch.tensor(v) if k == 'p' else v) for k, v in kw.items()}
        args = [n, d]
        if kind == 'c-given-a':
            args.append(a_0)
            kw['p_c_given_a'] = p_ca
        torch.manual_seed(100 + i)
        out[f'case{i}'] = fn(*args, **kw)
    import json
    np.savez_compressed(OUT / 'priors.npz', cases_json=np.array(json.dumps(PRIOR_CASES)), **_np(out))


def main():
    """no arguments: every fixture (the three long-horizon runs take ~20 min of the reference on 8 threads);
    ``long`` / ``long:<tag>``: only those; ``--skip-long``: everything else."""
    import argparse
    ap = argparse.ArgumentParser(prog='python -m oracle.make_golden', description='(Re)generate tests/golden/*.npz from the reference under /root/reference. '
                                 'Without arguments EVERY fixture is rewritten (about an hour on 8 threads; they regenerate bit for bit).')
    ap.add_argument('what', nargs='*', metavar='WHAT', help="'long' = every long-horizon fixture, 'long:<tag>' = that one "
                    f"(tags: {', '.join(LONG_CASES)}), 'traj_frames' = only the reference-format frame fixture; nothing = everything")
    ap.add_argument('--skip-long', action='store_true', help='everything except the long-horizon fixtures')
    ns_args = ap.parse_args()
    bad = [a for a in ns_args.what if not (a == 'traj_frames' or a == 'long' or (a.startswith('long:') and a.split(':', 1)[1] in LONG_CASES))]
    if bad:
        ap.error(f'unknown fixture selector(s) {bad}')
    args = list(ns_args.what) + (['--skip-long'] if ns_args.skip_long else [])
    torch.set_num_threads(8)
    OUT.mkdir(parents=True, exist_ok=True)
    ns = ref_standin.import_reference()
    if 'traj_frames' in args:          # only that fixture
        gen_traj_frames(ns)
        return
    only_long = [a for a in args if a.startswith('long')]
    if only_long or '--skip-long' not in args:
        tags = [a.split(':', 1)[1] for a in only_long if ':' in a] or list(LONG_CASES)
        for tag in tags:
            gen_integrate_long(ns, tag)
            print('long', tag, (OUT / f'long_{tag}.npz').stat().st_size // 1024, 'KiB', flush=True)
        if only_long:
            return
    gen_misc(ns)
    gen_ctmc_step(ns)
    gen_stability()
    gen_moldata()
    gen_priors(ns)
    # dev = configs/dev.yml:78-108 (64/64/16 dims, use_dst_feats); geom_arom = geom_full_aromatic.yaml / geom_5_aromatic.yaml (5 bond types)
    for name in ('flowmol3', 'geom_ctmc', 'qm9', 'dev', 'arch_variants', 'geom_arom', 'flowmol3_arom'):
        cfg = presets.PRESETS[name]()
        sd = weights.synth_state_dict(cfg, seed=0)
        gen_forward(ns, name, cfg, sd)
        if name not in ('qm9', 'arch_variants', 'geom_arom', 'flowmol3_arom'):
            gen_modules(ns, name, cfg, sd)
    cfg = presets.flowmol3(); sd = weights.synth_state_dict(cfg, 0)
    gen_integrate(ns, 'flowmol3', cfg, sd, [5, 12, 20, 33], 20, 'F7')
    cfg = presets.qm9(); sd = weights.synth_state_dict(cfg, 0)
    gen_integrate(ns, 'qm9', cfg, sd, [18] * 8, 20, 'C1')          # BASELINE.json configs[0]
    cfg = presets.geom_ctmc(); sd = weights.synth_state_dict(cfg, 0)
    gen_integrate(ns, 'geom_ctmc', cfg, sd, [5, 17, 8, 30], 16, 'C5s')
    cfg = presets.geom_arom(); sd = weights.synth_state_dict(cfg, 0)
    gen_integrate(ns, 'geom_arom', cfg, sd, [5, 17, 8, 30], 16, 'T16')
    cfg = presets.flowmol3_arom(); sd = weights.synth_state_dict(cfg, 0)
    gen_integrate(ns, 'flowmol3_arom', cfg, sd, [6, 14, 9], 12, 'T12')
    cfg = presets.qm9(); sd = weights.synth_state_dict(cfg, 0)
    gen_integrate_variant(ns, 'qm9', cfg, sd, [6, 3, 8], 'gat', 'gat')
    gen_integrate_variant(ns, 'qm9', cfg, sd, [6, 3, 8], 'sched', 'campbell')
    gen_integrate_cosine(ns, cfg, sd, [6, 3, 8])
    gen_integrate_endpoint(ns)
    gen_traj_frames(ns)
    for f in sorted(OUT.glob('*.npz')):
        print(f.name, f.stat().st_size // 1024, 'KiB')


if __name__ == '__main__':
    main()
