"""The CPU oracle (oracle/cpu_ref.py) against golden vectors produced by the REFERENCE's own
modules (oracle/make_golden.py, run in the build container).  Runs anywhere (no /root/reference)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from flowmol_amd import presets, weights
from oracle import cpu_ref

torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
TOL = dict(rtol=1e-6, atol=1e-6)   # same ops on the same CPU; observed bitwise-equal in the build container


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(golden_dir / name).items()}


def _onehots(cfg, batch, a, c, eu):
    m = batch.upper_edge_mask
    e = torch.zeros(batch.E, cfg.n_bond_types + 1)
    e[m] = F.one_hot(eu, cfg.n_bond_types + 1).float()
    e[~m] = F.one_hot(eu, cfg.n_bond_types + 1).float()
    return F.one_hot(a, cfg.n_atom_types + 1).float(), F.one_hot(c, cfg.n_charges + 1).float(), e


@pytest.mark.parametrize('name', ['flowmol3', 'geom_ctmc', 'qm9', 'dev', 'arch_variants', 'geom_arom', 'flowmol3_arom'])      # *_arom: explicit aromaticity, 5 bond types (geom_full_aromatic.yaml / geom_5_aromatic.yaml); arch_variants: n_recycles=2, message_norm='mean', no distance in EdgeUpdate, shared updater
def test_forward_matches_reference(golden_dir, name):
    cfg = presets.PRESETS[name]()
    g = _load(golden_dir, f'forward_{name}.npz')
    orc = cpu_ref.OracleVF(cfg, weights.synth_state_dict(cfg, 0))
    batch = cpu_ref.build_batch(g['n_atoms'])
    with torch.no_grad():
        for tag, tval in (('t0', 0.0), ('th', 0.5)):
            a, c, e = _onehots(cfg, batch, g[f'{tag}.a'], g[f'{tag}.c'], g[f'{tag}.e_upper'])
            prev = None
            if tag == 'th' and cfg.self_conditioning:
                prev = {k: g[f'th.prev.{k}'] for k in 'xace'}
            out = orc.forward(batch, g[f'{tag}.x_t'], a, c, e, torch.full((batch.B,), tval), prev=prev,
                              apply_softmax=True, remove_com=True)
            for k in 'xace':
                torch.testing.assert_close(out[k], g[f'{tag}.out.{k}'], **TOL)


@pytest.mark.parametrize('name', ['flowmol3', 'geom_ctmc', 'dev'])
def test_modules_match_reference(golden_dir, name):
    cfg = presets.PRESETS[name]()
    g = _load(golden_dir, f'modules_{name}.npz')
    orc = cpu_ref.OracleVF(cfg, weights.synth_state_dict(cfg, 0))
    batch = cpu_ref.build_batch(g['n_atoms'])
    with torch.no_grad():
        xd, d = orc.distances(batch, g['x'])
        torch.testing.assert_close(xd, g['x_diff'], **TOL)
        torch.testing.assert_close(d, g['d'], **TOL)
        s2, v2 = orc.conv(0, batch, g['s'], g['v'], g['ef'], xd, d)
        torch.testing.assert_close(s2, g['conv0.s'], **TOL)
        torch.testing.assert_close(v2, g['conv0.v'], **TOL)
        torch.testing.assert_close(orc.position_update(1, g['s'], g['x'], g['v']), g['pos1.x'], **TOL)
        torch.testing.assert_close(orc.edge_update(1, batch, g['s'], g['ef'], d), g['edge1.ef'], **TOL)
        os_, ov = orc.gvp('conv_layers.0.edge_message.0', g['gvp0.in_s'], g['gvp0.in_v'])
        torch.testing.assert_close(os_, g['gvp0.out_s'], **TOL)
        torch.testing.assert_close(ov, g['gvp0.out_v'], **TOL)
        if cfg.self_conditioning:
            prev = {k: g[f'sc.prev.{k}'] for k in 'xace'}
            so, eo = orc.self_conditioning(batch, g['s'], g['x'], g['sc.ef_in'], prev)
            torch.testing.assert_close(so, g['sc.s'], **TOL)
            torch.testing.assert_close(eo, g['sc.ef'], **TOL)


@pytest.mark.parametrize('fname,name', [('integrate_flowmol3_F7.npz', 'flowmol3'),
                                        ('integrate_qm9_C1.npz', 'qm9'),
                                        ('integrate_geom_ctmc_C5s.npz', 'geom_ctmc'),
                                        ('integrate_geom_arom_T16.npz', 'geom_arom'), ('integrate_flowmol3_arom_T12.npz', 'flowmol3_arom')])
def test_integrate_matches_reference(golden_dir, fname, name):
    """Free-running trajectory with the reference's recorded noise: categorical outcomes bit-exact,
    coordinates to 1e-5 (config C1 of BASELINE.json is the qm9 case)."""
    cfg = presets.PRESETS[name]()
    g = _load(golden_dir, fname)
    orc = cpu_ref.OracleVF(cfg, weights.synth_state_dict(cfg, 0))
    batch = cpu_ref.build_batch(g['n_atoms'])
    tape = [g[k] for k in sorted(k for k in g if k.startswith('noise.'))]
    prior = {'x_0': g['x_0'], 'a_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_atom_types),
             'c_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_charges),
             'e_0': cpu_ref.edge_prior(batch.upper_edge_mask, cfg.n_bond_types)}
    noise = cpu_ref.TapeNoise(tape)
    with torch.no_grad():
        out, frames = orc.integrate(batch, prior, int(g['T']), noise=noise, visualize=True)
    assert noise.pos == len(tape)                      # same number/order of draws as the reference
    m = batch.upper_edge_mask
    assert torch.equal(out['a_1'].argmax(-1), g['a_1'])
    assert torch.equal(out['c_1'].argmax(-1), g['c_1'])
    assert torch.equal(out['e_1'][m].argmax(-1), g['e_1_upper'])
    assert torch.equal(out['e_1'][m], out['e_1'][~m])
    torch.testing.assert_close(out['x_1'], g['x_1'], rtol=1e-5, atol=1e-5)
    n0 = int(g['n_atoms'][0])
    x_traj0 = torch.stack([f[:n0] for f in frames['x']])
    torch.testing.assert_close(x_traj0, g['traj0.x'], rtol=1e-5, atol=1e-5)
    # no mask tokens survive the last step
    assert (out['a_1'].argmax(-1) != cfg.n_atom_types).all()
    assert (out['e_1'].argmax(-1) != cfg.n_bond_types).all()


@pytest.mark.parametrize('fname,dfm_type', [('integrate_qm9_gat.npz', 'gat'), ('integrate_qm9_sched.npz', 'campbell')])
def test_integrator_variants_match_reference(golden_dir, fname, dfm_type):
    """non-uniform tspan + 'decay' temperature schedule + inv_temp_func, with dfm_type 'campbell'
    and 'gat' ('beta' forward-weight schedule): oracle vs the reference's own run, recorded noise."""
    from parity_util import variant_cfg, variant_inv_temp
    sys_cfg = variant_cfg(presets.qm9())
    g = _load(golden_dir, fname)
    orc = cpu_ref.OracleVF(sys_cfg, weights.synth_state_dict(sys_cfg, 0))
    batch = cpu_ref.build_batch(g['n_atoms'])
    tape = [g[k] for k in sorted(k for k in g if k.startswith('noise.'))]
    prior = {'x_0': g['x_0'], 'a_0': cpu_ref.ctmc_masked_prior(batch.N, sys_cfg.n_atom_types),
             'c_0': cpu_ref.ctmc_masked_prior(batch.N, sys_cfg.n_charges),
             'e_0': cpu_ref.edge_prior(batch.upper_edge_mask, sys_cfg.n_bond_types)}
    noise = cpu_ref.TapeNoise(tape)
    ctf = lambda t: 0.8 * torch.pow(1 - t, 2)                                  # ctmc_vector_field.py:74
    fwf = lambda t: 1 + 10.0 * torch.pow(t, 0.25) * torch.pow(1 - t, 0.25)     # :87
    with torch.no_grad():
        out, frames = orc.integrate(batch, prior, 0, noise=noise, visualize=True, tspan=g['tspan'], dfm_type=dfm_type,
                                    cat_temp_func=ctf, forward_weight_func=fwf, inv_temp_func=variant_inv_temp)
    assert noise.pos == len(tape)
    m = batch.upper_edge_mask
    assert torch.equal(out['a_1'].argmax(-1), g['a_1'])
    assert torch.equal(out['c_1'].argmax(-1), g['c_1'])
    assert torch.equal(out['e_1'][m].argmax(-1), g['e_1_upper'])
    torch.testing.assert_close(out['x_1'], g['x_1'], rtol=1e-5, atol=1e-5)
    n0 = int(g['n_atoms'][0])
    torch.testing.assert_close(torch.stack([f[:n0] for f in frames['x']]), g['traj0.x'], rtol=1e-5, atol=1e-5)
    assert torch.equal(torch.stack([f[:n0].argmax(-1) for f in frames['a_1_pred']]), g['traj0.a_1_pred'])


def test_misc_matches_reference(golden_dir):
    g = _load(golden_dir, 'misc.npz')
    torch.testing.assert_close(cpu_ref.time_embedding(g['temb.t'], 64), g['temb.out'], **TOL)
    torch.testing.assert_close(cpu_ref.rbf(g['rbf.d'], 10, 32), g['rbf.out10'], **TOL)
    torch.testing.assert_close(cpu_ref.rbf(g['rbf.d'], 12, 32), g['rbf.out12'], **TOL)
    for n in (2, 3, 7):
        assert torch.equal(cpu_ref.build_edge_idxs(n), g[f'edges.{n}'])
    a, ap = cpu_ref.alpha_tables(g['alpha.t'])
    assert torch.equal(a, g['alpha.a']) and torch.equal(ap, g['alpha.ap'])


@pytest.mark.parametrize('case', [0, 1, 2, 3])
def test_campbell_step_matches_reference(golden_dir, case):
    """purity-sampling edge cases: a molecule with h=0, one with m=h, hc=0 branch, last step."""
    g = _load(golden_dir, 'misc.npz')
    cfg = presets.flowmol3()
    orc = cpu_ref.OracleVF(cfg, weights.synth_state_dict(cfg, 0))
    hc, last, eta, alpha, dt = [float(v) for v in g[f'ctmc.{case}.params']]
    sizes = g['ctmc.sizes']
    bidx = torch.arange(sizes.shape[0]).repeat_interleave(sizes)
    tape = [g[f'ctmc.{case}.noise{i}'] for i in range(3 if not last else 2)]
    xt1h, x11h = orc.campbell_step(g['ctmc.p'], g['ctmc.xt'].clone(), eta, hc, torch.tensor(alpha), torch.tensor(1.0),
                                   torch.tensor(dt), sizes.shape[0], 12, 11, bool(last), bidx, cpu_ref.TapeNoise(tape))
    assert torch.equal(xt1h.argmax(-1), g[f'ctmc.{case}.xt_new'])
    assert torch.equal(x11h.argmax(-1), g[f'ctmc.{case}.x1'])


@pytest.mark.parametrize('tag,dataset,arom,fake', [('kek', 'geom_full_kekulized', False, True), ('arom', 'geom_5_aromatic', True, False)])
def test_stability_restatement_matches_reference(golden_dir, tag, dataset, arom, fake):
    """oracle compute_valencies + check_stability vs the verdicts of the reference's own functions
    (molecule_builder.py:138-157, metrics.py:333-363) on the token molecules of tests/golden/stability.npz."""
    import torch.nn.functional as F
    from flowmol_amd import metrics
    g = _load(golden_dir, 'stability.npz')
    table, ar = metrics.load_valency_table(dataset)
    assert ar == arom
    atom_map = ['C', 'H', 'N', 'O', 'F', 'P', 'S', 'Cl', 'Br', 'I']
    nb = 5 if arom else 4
    no = po = 0
    for i, n in enumerate(g[f'{tag}.n_atoms'].tolist()):
        u = n * (n - 1) // 2
        a, c, e = g[f'{tag}.a'][no:no + n], g[f'{tag}.c'][no:no + n], g[f'{tag}.e'][po:po + u]
        no += n; po += u
        pos, sym, chg, bt, bs, bd = cpu_ref.extract_moldata(torch.zeros(n, 3), F.one_hot(a, len(atom_map) + (2 if fake else 1)).float(),
                                                            F.one_hot(c, 6).float(), torch.cat([F.one_hot(e, nb + 1).float()] * 2), n,
                                                            atom_map, fake, nb)
        val = cpu_ref.compute_valencies(len(sym), bt, bs, bd, arom_dependent=arom)
        n_stable, mol_stable = cpu_ref.check_stability(sym, val, chg, table, explicit_aromaticity=arom)
        assert [n_stable, int(mol_stable), len(sym)] == g[f'{tag}.expect'][i].tolist()


def _moldata_cases(golden_dir):
    """(tag, arom, fake, per-molecule inputs and the reference's extract_moldata_from_graph outputs) of tests/golden/moldata.npz."""
    g = _load(golden_dir, 'moldata.npz')
    st = _load(golden_dir, 'stability.npz')
    base = ['C', 'H', 'N', 'O', 'F', 'P', 'S', 'Cl', 'Br', 'I']
    for tag, arom, fake in (('kek', False, True), ('arom', True, False)):
        amap = base + (['Sn'] if fake else []) + ['Se']
        no = po = ao = bo = 0
        for i, n in enumerate(st[f'{tag}.n_atoms'].tolist()):
            u = n * (n - 1) // 2
            na, nbd = g[f'{tag}.counts'][i].tolist()
            yield dict(tag=tag, arom=arom, fake=fake, n=n, base=base, amap=amap,
                       x=g[f'{tag}.x'][no:no + n], a=st[f'{tag}.a'][no:no + n], c=st[f'{tag}.c'][no:no + n], e=st[f'{tag}.e'][po:po + u],
                       pos=g[f'{tag}.pos'][ao:ao + na], sym=[amap[int(k)] for k in g[f'{tag}.sym'][ao:ao + na]], chg=g[f'{tag}.chg'][ao:ao + na],
                       bt=g[f'{tag}.bt'][bo:bo + nbd], bs=g[f'{tag}.bs'][bo:bo + nbd], bd=g[f'{tag}.bd'][bo:bo + nbd])
            no += n; po += u; ao += na; bo += nbd


def test_extract_moldata_matches_reference(golden_dir):
    """the oracle's result extraction vs the reference's own extract_moldata_from_graph
    (molecule_builder.py:217-265, run by oracle/make_golden.py:gen_moldata) -- fake atoms dropped and bonds re-indexed,
    masked bonds = no bond, charge = index - 2, upper-triangle bonds only."""
    import torch.nn.functional as F
    n_cases = 0
    for k in _moldata_cases(golden_dir):
        nb = 5 if k['arom'] else 4
        pos, sym, chg, bt, bs, bd = cpu_ref.extract_moldata(k['x'], F.one_hot(k['a'], len(k['amap'])).float(), F.one_hot(k['c'], 6).float(),
                                                            torch.cat([F.one_hot(k['e'], nb + 1).float()] * 2), k['n'], k['base'], k['fake'], nb)
        assert torch.equal(pos, k['pos']) and sym == k['sym'] and torch.equal(chg, k['chg'])
        assert torch.equal(bt, k['bt']) and torch.equal(bs, k['bs']) and torch.equal(bd, k['bd'])
        n_cases += 1
    assert n_cases == 16


@pytest.mark.parametrize('case', [0, 1, 2, 3, 4])
def test_ctmc_step_matches_reference_step(golden_dir, case):
    """The oracle's step (Euler + tempering + campbell_step + purity sampling) against the reference's own
    CTMCVectorField.step run with a fixed endpoint prediction (tests/golden/ctmc_step.npz): h = 0, m = h, m = 0 molecules,
    a one-pair molecule, an exact-zero probability, hc = 0 branch, last step -- bit-exact tokens and coordinates."""
    g = _load(golden_dir, 'ctmc_step.npz')
    cfg = presets.flowmol3()
    hc, last, eta, s_idx, T = [float(v) for v in g[f'{case}.params']]
    last, s_idx, T = bool(last), int(s_idx), int(T)
    batch = cpu_ref.build_batch(g['n_atoms'])
    dst = {k: g[f'{case}.dst.{k}'] for k in 'xace'}

    class FixedDst(cpu_ref.OracleVF):
        def forward(self, *a_, **k_):
            return dst
    orc = FixedDst(cfg, weights.synth_state_dict(cfg, 0))
    import torch.nn.functional as F
    m = batch.upper_edge_mask
    e1h = torch.zeros(batch.E, cfg.n_bond_types + 1)
    e1h[m] = F.one_hot(g[f'{case}.e_t'], cfg.n_bond_types + 1).float()
    e1h[~m] = F.one_hot(g[f'{case}.e_t'], cfg.n_bond_types + 1).float()
    state = {'x_t': g[f'{case}.x_t'], 'a_t': F.one_hot(g[f'{case}.a_t'], cfg.n_atom_types + 1).float(),
             'c_t': F.one_hot(g[f'{case}.c_t'], cfg.n_charges + 1).float(), 'e_t': e1h}
    t = torch.linspace(0, 1, T)
    al, alp = cpu_ref.alpha_tables(t)
    tape = [g[f'{case}.noise{i}'] for i in range(6 if last else 9)]
    new, _ = orc.step(batch, state, t[s_idx], t[s_idx - 1], al[s_idx - 1], alp[s_idx - 1], prev=None, eta=eta, hc_thresh=hc,
                      last_step=last, noise=cpu_ref.TapeNoise(tape))
    assert torch.equal(new['x_t'], g[f'{case}.x_new'])
    for k in 'ac':
        assert torch.equal(new[f'{k}_t'].argmax(-1), g[f'{case}.{k}_new']) and torch.equal(new[f'{k}_1_pred'].argmax(-1), g[f'{case}.{k}_1_pred'])
    assert torch.equal(new['e_t'][m].argmax(-1), g[f'{case}.e_new']) and torch.equal(new['e_1_pred'][m].argmax(-1), g[f'{case}.e_1_pred'])


def test_cosine_schedule_integrate_matches_reference(golden_dir):
    """Cosine interpolant schedule (interpolant_scheduler.py:131-146) through the oracle: alpha tables incl. the in-place clamp of
    t[0] to 1e-9, and the free-running trajectory (no bootstrap evaluation) against the reference's own integrate()."""
    from parity_util import cosine_cfg
    g = _load(golden_dir, 'integrate_qm9_cosine.npz')
    cfg = cosine_cfg(presets.qm9())
    T = int(g['T'])
    t = torch.linspace(0, 1, T)
    a, ap = cpu_ref.alpha_tables(t, cfg.schedule_type, cfg.cosine_params)
    assert torch.equal(a, g['alpha.a']) and torch.equal(ap, g['alpha.ap']) and torch.equal(t, g['alpha.t_after'])
    orc = cpu_ref.OracleVF(cfg, weights.synth_state_dict(cfg, 0))
    batch = cpu_ref.build_batch(g['n_atoms'])
    tape = [g[k] for k in sorted(k for k in g if k.startswith('noise.'))]
    prior = {'x_0': g['x_0'], 'a_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_atom_types), 'c_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_charges),
             'e_0': cpu_ref.edge_prior(batch.upper_edge_mask, cfg.n_bond_types)}
    with torch.no_grad():
        out = orc.integrate(batch, prior, T, noise=cpu_ref.TapeNoise(tape))
    assert torch.equal(out['a_1'].argmax(-1), g['a_1']) and torch.equal(out['c_1'].argmax(-1), g['c_1'])
    assert torch.equal(out['e_1'][batch.upper_edge_mask].argmax(-1), g['e_1_upper'])
    torch.testing.assert_close(out['x_1'], g['x_1'], **TOL)


def test_endpoint_parameterization_matches_reference(golden_dir):
    """Oracle restatement of EndpointVectorField.forward / step / integrate (vector_field.py:212-293, 388-569) against the reference's
    own module (tests/golden/integrate_endpoint.npz): continuous categorical features, Euler steps of x, a, c, e."""
    from parity_util import endpoint_cfg
    g = _load(golden_dir, 'integrate_endpoint.npz')
    cfg = endpoint_cfg()
    orc = cpu_ref.OracleVF(cfg, weights.synth_state_dict(cfg, 0))
    batch = cpu_ref.build_batch(g['n_atoms'])
    m = batch.upper_edge_mask
    e0 = torch.zeros(batch.E, cfg.n_bond_types)
    e0[m] = g['e_0_upper']; e0[~m] = g['e_0_upper']
    with torch.no_grad():
        d0 = orc.forward(batch, g['x_0'], g['a_0'], g['c_0'], e0, torch.full((batch.B,), 0.25), prev=None, apply_softmax=True, remove_com=True)
        out = orc.integrate_endpoint(batch, {'x_0': g['x_0'], 'a_0': g['a_0'], 'c_0': g['c_0'], 'e_0': e0}, int(g['T']))
    for k in 'xace':
        torch.testing.assert_close(d0[k], g[f'fwd.{k}'], **TOL)
    for k in 'xac':
        torch.testing.assert_close(out[f'{k}_1'], g[f'{k}_1'], **TOL)
    torch.testing.assert_close(out['e_1'][m], g['e_1_upper'], **TOL)


LONG = [('flowmol3_47x8_T250', 'flowmol3'), ('flowmol3_mixed_T250_w2', 'flowmol3'), ('geom_ctmc_mixed_T500', 'geom_ctmc'),
        ('flowmol3_geom64_T250', 'flowmol3'),       # 64 GEOM-sized molecules (r4): 3 steps here (1.3 s of oracle per evaluation)
        ('flowmol3_geom16_T250_pos128', 'flowmol3'),       # 16 GEOM-sized molecules, position heads x128 (r5): the coordinates move 4.6 % per evaluation
        ('flowmol3_geom16_T250_heads256', 'flowmol3')]     # the same sizes, categorical heads x256 (r6): the trained-model regime -- exact zeros, p == 1, denormals, log 0 = -inf


@pytest.mark.parametrize('tag,name', LONG)
def test_long_horizon_reference_trajectory_first_steps(golden_dir, tag, name):
    """The oracle free-running on the reference's default-protocol trajectories (250 / 500 steps; noise re-drawn from the stored seed in the
    reference's order).  Here only the first 12 steps, every state and sampled token of every step against the reference's (the suite has to
    finish in minutes); FM_LONG_ORACLE=1 runs the whole horizon, whose result is committed as profiles/r03_oracle_long_parity.jsonl
    (tools/oracle_long_parity.py).  The full-length comparison that gates the product is the -m gpu test of the same fixtures."""
    import os
    from parity_util import oracle_long_golden
    g = _load(golden_dir, f'long_{tag}.npz')
    cfg = presets.PRESETS[name]()
    sd = weights.long_fixture_weights(cfg, g)
    full = os.environ.get('FM_LONG_ORACLE') == '1'
    res = oracle_long_golden(cpu_ref.OracleVF(cfg, sd), cfg, g, max_steps=None if full else (3 if int(g['n_atoms'].numel()) > 16 else 12))
    assert res['first_divergent_step'] is None and res['state_token_diffs_all_steps'] == 0, res
    assert res['x_norm_rel'] < 1e-5 and res['x1_norm_rel'] < 1e-5 and res['x_frames_rel'] < 1e-5, res
    if full:
        assert res['a_flips'] == res['c_flips'] == res['e_flips'] == 0 and res['x_rel'] < 1e-4, res
    if float(g['weight_scale']) > 1 or 'pos_head_scale' in g:
        assert res['mean_rel_move'] > 0.02, res            # the scaled weights really make the coordinates depend on the network


def test_long_fixture_noise_is_the_seeded_redraw(golden_dir):
    """The long fixtures store seeds instead of the noise tape: the fixture's first sampled tokens must follow from the seeded re-draw
    (StepNoise.draw on torch's CPU generator) fed through the oracle's teacher-forced CTMC step -- i.e. the stored seed really is the
    reference's stream (make_golden asserts the re-draw equals every recorded draw when the fixture is generated)."""
    from flowmol_amd.engine import StepNoise
    g = _load(golden_dir, 'long_geom_ctmc_mixed_T500.npz')
    cfg = presets.geom_ctmc()
    N = int(g['n_atoms'].sum()); U = int((g['n_atoms'] * (g['n_atoms'] - 1) // 2).sum())
    torch.manual_seed(int(g['seed_noise']))
    nz = StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, False, 'cpu')
    torch.manual_seed(int(g['seed_noise']))
    rec = cpu_ref.TorchNoise()
    assert torch.equal(nz.q_a, rec.exp_like(torch.empty(N, cfg.n_atom_types)))
    assert g['traj.a'].shape == (int(g['T']), N) and g['traj.e1'].shape == (int(g['T']) - 1, U)
    assert bool((g['traj.a'][0] == cfg.n_atom_types).all()) and not bool((g['traj.a'][-1] == cfg.n_atom_types).any())   # masked prior -> no mask left
    assert torch.equal(g['traj.a'][-1].long(), g['a_1'].long()) and torch.equal(g['traj.e'][-1].long(), g['e_1_upper'].long())
