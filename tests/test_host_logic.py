"""Host-side logic (no GPU): configs, step scalars, result extraction, checkpoint reader, sharding."""
import math

import numpy as np

import pytest
import torch

from flowmol_amd import presets, weights
from flowmol_amd.config import VFConfig, from_reference_hparams
from flowmol_amd.engine import StepNoise, make_step_plan, time_embedding_host
from flowmol_amd.model import FlowMol, load_pretrained, read_checkpoint
from flowmol_amd.molecule import SampledMolecule, extract_moldata, mol_block
from flowmol_amd.shard import pack_results, partition_lpt, unpack_results
from oracle import cpu_ref


def test_param_counts_match_survey_appendix_a():
    assert weights.n_params(presets.flowmol3()) == 5854185
    assert weights.n_params(presets.geom_ctmc()) == 4288922


def test_update_schedule_quirk():
    # convs_per_update=1: no update after conv 0, updater index = conv index (updater 0 dead) -- SURVEY Appendix C.2
    assert presets.flowmol3().update_schedule() == [-1, 1, 2, 3, 4, 5]
    assert presets.geom_ctmc().update_schedule() == [-1, 1, 2, 3, 4]
    c = VFConfig(n_molecule_updates=2, convs_per_update=2, separate_mol_updaters=False)
    assert c.update_schedule() == [-1, 0, -1, 0]


def test_unsupported_configs_fail_loudly():
    VFConfig(use_dst_feats=True).validate()                 # configs/dev.yml features are implemented since round 2 ...
    VFConfig(n_hidden_scalars=64, n_hidden_edge_feats=64).validate()
    with pytest.raises(NotImplementedError):
        VFConfig(use_dst_feats=True, dst_feat_msg_reduction_factor=1).validate()      # ... except the projection-free variant
    with pytest.raises(NotImplementedError):
        VFConfig(n_hidden_scalars=320).validate()            # wider than the 256-column tiles
    with pytest.raises(NotImplementedError):
        VFConfig(n_hidden_edge_feats=192).validate()
    VFConfig(message_norm='mean', n_recycles=2, update_edge_w_distance=False).validate()       # implemented since ABI 4
    with pytest.raises(ValueError):
        VFConfig(message_norm='max').validate()              # gvp.py:395-396
    with pytest.raises(ValueError):
        VFConfig(n_recycles=0).validate()


@pytest.mark.parametrize('eta,hc,T', [(30.0, 0.9, 250), (10.0, 0.0, 20), (0.0, 0.9, 7)])
def test_step_plan_matches_oracle_arithmetic(eta, hc, T):
    plan = make_step_plan(T, eta, hc, 0.05)
    t = torch.linspace(0, 1, T)
    al, alp = cpu_ref.alpha_tables(t)
    assert len(plan.scalars) == T - 1
    for s_idx in (1, T // 2, T - 1):
        sc = plan.scalars[s_idx - 1]
        dt = t[s_idx] - t[s_idx - 1]
        a_i, ap_i = al[s_idx - 1], alp[s_idx - 1]
        assert sc.t == float(t[s_idx - 1]) and sc.dt == float(dt)
        assert sc.x_coef == float(ap_i[0] / (1 - a_i[0]))
        assert sc.unmask_prob[1] == float(torch.clamp(dt * (ap_i[1] + eta * a_i[1]) / (1 - a_i[1]), min=0, max=1))
        assert sc.mask_prob[2] == float(torch.clamp(dt * eta, min=0, max=1))
        assert bool(sc.last_step) == (s_idx == T - 1)
    assert plan.scalars[-1].unmask_prob[0] == 1.0        # last step resolves every mask token (Appendix C.7)


def test_time_embedding_host_matches_oracle():
    for t in (0.0, 0.004016064, 0.5, 1.0):
        a = time_embedding_host(t, 64)
        b = cpu_ref.time_embedding(torch.tensor([t], dtype=torch.float32), 64)[0]
        assert torch.equal(a, b)
    assert torch.equal(time_embedding_host(0.25, 1), torch.tensor([0.25]))


def test_step_noise_draw_order_matches_reference_rng_stream():
    N, U, na, nc, ne = 7, 21, 11, 6, 4
    torch.manual_seed(3)
    nz = StepNoise.draw(N, U, na, nc, ne, False, 'cpu')
    torch.manual_seed(3)
    rec = cpu_ref.TorchNoise()
    for tag, rows, k in (('a', N, na), ('c', N, nc), ('e', U, ne)):
        assert torch.equal(getattr(nz, f'q_{tag}'), rec.exp_like(torch.empty(rows, k)))
        assert torch.equal(getattr(nz, f'u1_{tag}'), rec.rand(rows, 'cpu'))
        assert torch.equal(getattr(nz, f'u2_{tag}'), rec.rand(rows, 'cpu'))


def test_extract_moldata_matches_oracle_restatement():
    gen = torch.Generator().manual_seed(0)
    n = 9
    amap = presets.GEOM_ATOMS
    a = torch.randint(0, 11, (n,), generator=gen)
    a[2] = 10                      # a fake atom
    c = torch.randint(0, 6, (n,), generator=gen)
    u = n * (n - 1) // 2
    e = torch.randint(0, 5, (u,), generator=gen)
    x = torch.randn(n, 3, generator=gen)
    got = extract_moldata(x, a, c, e, amap, True)
    import torch.nn.functional as F
    e1h = torch.cat([F.one_hot(e, 5), F.one_hot(e, 5)]).float()
    want = cpu_ref.extract_moldata(x, F.one_hot(a, 12).float(), F.one_hot(c, 7).float(), e1h, n, amap, True)
    assert torch.equal(got[0], want[0]) and got[1] == want[1] and torch.equal(got[2], want[2])
    assert torch.equal(got[3], want[3]) and torch.equal(got[4], want[4]) and torch.equal(got[5], want[5])
    m = SampledMolecule(x, a, c, e, amap, fake_atoms=True)
    assert m.num_atoms == n - int((a == 10).sum()) == len(m.atom_types)
    assert m.valencies.shape == (m.num_atoms,)
    blk = mol_block(m.positions, m.atom_types, m.atom_charges, m.bond_src_idxs, m.bond_dst_idxs, m.bond_types)
    assert blk.splitlines()[3].startswith(f'{m.num_atoms:3d}{m.bond_types.shape[0]:3d}') and blk.rstrip().endswith('M  END')
    assert m.rdkit_mol is None or m.rdkit_mol.GetNumAtoms() == m.num_atoms


def test_checkpoint_reader_roundtrip(tmp_path):
    cfg = presets.flowmol3()
    sd = {'vector_field.' + k: v for k, v in weights.synth_state_dict(cfg, 0).items()}
    hp = {'atom_type_map': presets.GEOM_ATOMS, 'n_atoms_hist_file': 'data/geom_full_kekulized/train_data_n_atoms_histogram.pt',
          'marginal_dists_file': 'x', 'n_atom_charges': 6, 'parameterization': 'ctmc', 'fake_atom_p': 0.3, 'default_n_timesteps': 250,
          'prior_config': {**{k: {'type': 'ctmc'} for k in 'ace'}, 'x': {'type': 'centered-normal', 'kwargs': {'std': 1.0}}},
          'interpolant_scheduler_config': {'schedule_type': {k: 'linear' for k in 'xace'}},
          'vector_field_config': dict(self_conditioning=True, stochasticity=30.0, high_confidence_threshold=0.9, n_vec_channels=32,
                                      update_edge_w_distance=True, n_hidden_scalars=256, n_hidden_edge_feats=128, s_message_dim=None,
                                      v_message_dim=None, n_expansion_gvps=3, attention=False, n_heads=32, n_recycles=1,
                                      separate_mol_updaters=True, n_molecule_updates=6, convs_per_update=1, n_cp_feats=4, n_message_gvps=3,
                                      n_update_gvps=3, message_norm='sum', rbf_dmax=10, rbf_dim=32, time_embedding_dim=64, a_token_dim=64,
                                      c_token_dim=64, e_token_dim=64)}
    d = tmp_path / 'flowmol3' / 'checkpoints'
    d.mkdir(parents=True)
    torch.save({'state_dict': sd, 'hyper_parameters': hp}, d / 'last.ckpt')
    hp2, sd2 = read_checkpoint(d / 'last.ckpt')
    cfg2 = from_reference_hparams(hp2)
    assert cfg2.to_dict() == cfg.to_dict()
    model = FlowMol.load_from_checkpoint(d / 'last.ckpt')
    assert model.n_atom_types == 11 and model.default_n_timesteps == 250
    with pytest.raises(RuntimeError, match='MI355X only'):
        model.sample(torch.tensor([3]))            # no silent CPU path
    # configurations the HIP path does not reproduce are rejected at load time, with the REFERENCE's defaults for missing keys
    from flowmol_amd.model import check_reference_hparams
    for bad in ({**hp, 'interpolant_scheduler_config': {'schedule_type': 'tanh'}},                # not a reference schedule
                {**hp, 'prior_config': {**hp['prior_config'], 'x': {'type': 'gaussian'}}},       # would be centred silently
                {**hp, 'exclude_charges': True},
                {**hp, 'parameterization': 'dirichlet'},
                {**hp, 'vector_field_config': {**hp['vector_field_config'], 'attention': True}},
                {**hp, 'vector_field_config': {**hp['vector_field_config'], 'made_up_key': 1}}):
        with pytest.raises((NotImplementedError, ValueError)):
            check_reference_hparams(bad)
            from_reference_hparams(bad)
    with pytest.raises(FileNotFoundError):
        load_pretrained('flowmol3')
    with pytest.raises(ValueError):
        load_pretrained('not-a-model')


def test_n_atoms_distribution():
    m = FlowMol.from_preset('qm9')
    torch.manual_seed(0)
    n = m.sample_n_atoms(1000)
    assert n.min() >= 3 and n.max() <= 29 and abs(float(n.float().mean()) - 18.0) < 0.5
    g = FlowMol.from_preset('flowmol3')
    n = g.sample_n_atoms(2000)
    assert n.min() >= 3 and n.max() <= 181 and abs(float(n.float().mean()) - 46.9) < 1.5


def test_lpt_partition_and_packing():
    torch.manual_seed(0)
    n = torch.randint(5, 90, (37,))
    parts = partition_lpt(n, 8)
    allidx = torch.cat(parts).sort().values
    assert torch.equal(allidx, torch.arange(37))
    cost = (n * (n - 1)).float()
    loads = torch.tensor([float(cost[p].sum()) for p in parts])
    assert loads.max() / loads.mean() < 1.25
    N, U = 11, 20
    x = torch.randn(N, 3); a = torch.randint(0, 12, (N,)); c = torch.randint(0, 7, (N,)); e = torch.randint(0, 5, (U,))
    got = unpack_results(pack_results(x, a, c, e), N, U)
    assert torch.equal(got['x'], x) and torch.equal(got['a'].long(), a) and torch.equal(got['e'].long(), e)


def test_rigid_alignment_recovers_rotation():
    from flowmol_amd.molecule import rigid_alignment
    g = torch.Generator().manual_seed(0)
    x = torch.randn(12, 3, generator=g)
    x = x - x.mean(0, keepdim=True)       # trajectory frames are COM-free; for x_0 with a non-zero mean the reference's
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))   # formula (priors.py:163-167) adds m0 - R m0 on top -- reproduced, not fixed
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    y = x @ q.T + torch.tensor([1.0, -2.0, 0.5])
    assert torch.allclose(rigid_alignment(x, y), y, atol=1e-5)


def test_sampled_molecule_matches_reference_extraction(golden_dir):
    """flowmol_amd.molecule.SampledMolecule (token tensors in, the product's packaging) against the
    reference's own extract_moldata_from_graph outputs (tests/golden/moldata.npz): positions, symbols, charges, bond
    list after fake-atom removal, num_atoms."""
    from test_oracle_golden import _moldata_cases
    from flowmol_amd.molecule import SampledMolecule
    for k in _moldata_cases(golden_dir):
        m = SampledMolecule(k['x'], k['a'].int(), k['c'].int(), k['e'].int(), k['base'], fake_atoms=k['fake'], ctmc_mol=True,
                            explicit_aromaticity=k['arom'])
        assert torch.equal(m.positions, k['pos']) and m.atom_types == k['sym'] and torch.equal(m.atom_charges, k['chg'])
        assert torch.equal(m.bond_types, k['bt']) and torch.equal(m.bond_src_idxs, k['bs']) and torch.equal(m.bond_dst_idxs, k['bd'])
        assert m.num_atoms == len(k['sym']) and m.atom_type_map == k['amap']


def cli_block_token_molecules():
    """Token form of the three hand-derived molecules of tests/golden/cli_expected_blocks.sdf (flowmol3 token tables:
    atoms C,H,N,O,F,P,S,Cl,Br,I + fake 'Sn' (10) + mask (11); charge token = charge + 2; bond token 4 = mask):
    (1) methane with a FAKE atom bonded to C and to an H -- the atom and both bonds disappear, the remaining atoms keep their order;
    (2) ammonium + hydroxide: formal charges +1 / -1 -> charge column 3 / 5 and one 'M  CHG' line;
    (3) formaldehyde with a MASKED H..H pair and explicit no-bond tokens: only the three real bonds are written."""
    def pairs(n, bonds):
        e = torch.zeros(n * (n - 1) // 2, dtype=torch.int32)
        for (i, j), t in bonds.items():
            e[i * (2 * n - i - 1) // 2 + (j - i - 1)] = t
        return e
    m1 = dict(x=torch.tensor([[0, 0, 0], [9.5, 9.5, 9.5], [.63, .63, .63], [-.63, -.63, .63], [-.63, .63, -.63], [.63, -.63, -.63]]),
              a=torch.tensor([0, 10, 1, 1, 1, 1], dtype=torch.int32), c=torch.tensor([2, 2, 2, 2, 2, 2], dtype=torch.int32),
              e=pairs(6, {(0, 1): 1, (1, 2): 2, (0, 2): 1, (0, 3): 1, (0, 4): 1, (0, 5): 1}))
    m2 = dict(x=torch.tensor([[0, 0, 0], [.59, .59, .59], [-.59, -.59, .59], [-.59, .59, -.59], [.59, -.59, -.59], [3., 0, 0], [3.96, 0, .12349]]),
              a=torch.tensor([2, 1, 1, 1, 1, 3, 1], dtype=torch.int32), c=torch.tensor([3, 2, 2, 2, 2, 1, 2], dtype=torch.int32),
              e=pairs(7, {(0, 1): 1, (0, 2): 1, (0, 3): 1, (0, 4): 1, (5, 6): 1}))
    m3 = dict(x=torch.tensor([[0, 0, 0], [1.21, 0, 0], [-.55, .94, 0], [-.55, -.94, -.00006]]),
              a=torch.tensor([0, 3, 1, 1], dtype=torch.int32), c=torch.tensor([2, 2, 2, 2], dtype=torch.int32),
              e=pairs(4, {(0, 1): 2, (0, 2): 1, (0, 3): 1, (2, 3): 4}))
    return [m1, m2, m3]


def test_sdf_writer_reproduces_hand_derived_blocks(golden_dir, tmp_path):
    """the RDKit-free V2000 writer against hand-derived mol blocks (fake atom dropped + bonds re-indexed, masked
    bond = no bond, formal charges in the atom column and in `M  CHG`), through the same write_sdf the CLI uses."""
    from flowmol_amd import cli
    mols = [SampledMolecule(m['x'].float(), m['a'], m['c'], m['e'], presets.GEOM_ATOMS, fake_atoms=True) for m in cli_block_token_molecules()]
    cli.write_sdf(tmp_path / 'o.sdf', [m.to_sdf_block() for m in mols])
    assert (tmp_path / 'o.sdf').read_text() == (golden_dir / 'cli_expected_blocks.sdf').read_text()


def test_cosine_step_plan_matches_reference_tables(golden_dir):
    """engine.alpha_tables / make_step_plan under the cosine schedule against the reference scheduler's own tables
    (tests/golden/integrate_qm9_cosine.npz): alpha, alpha', the clamped first time point, per-modality coefficients."""
    import numpy as np
    from flowmol_amd.engine import alpha_tables
    from parity_util import cosine_cfg
    g = {k: torch.from_numpy(v) for k, v in np.load(golden_dir / 'integrate_qm9_cosine.npz').items()}
    cfg = cosine_cfg(presets.qm9())
    T = int(g['T'])
    t = torch.linspace(0, 1, T)
    a, ap = alpha_tables(t, cfg.schedule_type, cfg.cosine_params)
    assert torch.equal(a, g['alpha.a']) and torch.equal(ap, g['alpha.ap']) and torch.equal(t, g['alpha.t_after'])
    plan = make_step_plan(T, 30.0, 0.9, 0.05, schedule_type=cfg.schedule_type, cosine_params=cfg.cosine_params)
    assert plan.scalars[0].t == float(torch.tensor(1e-9))                 # no bootstrap: t_0 != 0
    i = 4
    dt = t[i + 1] - t[i]
    sc = plan.scalars[i]
    assert sc.x_coef == float(ap[i, 0] / (1 - a[i, 0]))
    assert sc.unmask_prob[0] == float(torch.clamp(dt * (ap[i, 1] + 30.0 * a[i, 1]) / (1 - a[i, 1]), min=0, max=1))
    assert sc.unmask_prob[2] == float(torch.clamp(dt * (ap[i, 3] + 30.0 * a[i, 3]) / (1 - a[i, 3]), min=0, max=1))


def test_endpoint_hparams_are_read_like_the_reference():
    """from_reference_hparams for a non-CTMC checkpoint: parameterization default 'endpoint', prior types / kwargs, token dims 0,
    the vector field's inverse-temperature schedule; the CTMC-only / unimplemented variants stay rejected."""
    from flowmol_amd.model import check_reference_hparams
    hp = {'atom_type_map': ['C', 'H', 'N', 'O', 'F'], 'n_atom_charges': 6, 'fake_atom_p': 0.0,
          'prior_config': {'x': {'type': 'centered-normal', 'kwargs': {'std': 1.0}}, 'a': {'type': 'gaussian', 'kwargs': {'std': 1.0}},
                           'c': {'type': 'uniform-simplex', 'kwargs': {}}, 'e': {'type': 'barycenter', 'kwargs': {}}},
          'interpolant_scheduler_config': {'schedule_type': 'linear'},
          'vector_field_config': dict(n_vec_channels=16, n_hidden_scalars=256, n_hidden_edge_feats=128, n_cp_feats=4, n_molecule_updates=3,
                                      convs_per_update=1, separate_mol_updaters=True, message_norm=100, update_edge_w_distance=True,
                                      rbf_dmax=12, rbf_dim=32, continuous_inv_temp_schedule='linear', continuous_inv_temp_max=4.0)}
    check_reference_hparams(hp)                       # no 'parameterization' key -> the reference's default 'endpoint'
    cfg = from_reference_hparams(hp)
    assert cfg.parameterization == 'endpoint' and not cfg.has_mask and cfg.token_dims == (5, 6, 4)
    assert cfg.prior_types == {'a': 'gaussian', 'c': 'uniform-simplex', 'e': 'barycenter'} and cfg.continuous_inv_temp_max == 4.0
    check_reference_hparams({**hp, 'prior_config': {**hp['prior_config'], 'a': {'type': 'marginal', 'kwargs': {}}}})       # dataset-statistics priors: implemented
    with pytest.raises(NotImplementedError):
        check_reference_hparams({**hp, 'prior_config': {**hp['prior_config'], 'a': {'type': 'no-such-prior', 'kwargs': {}}}})
    with pytest.raises(ValueError):
        check_reference_hparams({**hp, 'prior_config': {**hp['prior_config'], 'a': {'type': 'c-given-a', 'kwargs': {}}}})   # a charge prior only
    with pytest.raises(NotImplementedError):
        from_reference_hparams({**hp, 'vector_field_config': {**hp['vector_field_config'], 'self_conditioning': True}})


def test_categorical_priors_match_reference_functions(golden_dir):
    """Every categorical prior FlowMol.sample_prior can dispatch to (reference priors.py:8-107, register :253-262): the host
    implementations consume the CPU generator exactly like the reference's functions -- fixture = their outputs under
    torch.manual_seed(100 + case), generated by oracle/make_golden.py:gen_priors."""
    import json
    from flowmol_amd.model import FlowMol
    g = np.load(golden_dir / 'priors.npz')
    cases = json.loads(str(g['cases_json']))
    a_0, p_ca = torch.from_numpy(g['a_0']), torch.from_numpy(g['p_c_given_a'])
    assert len(cases) >= 9
    for i, (kind, n, d, kw) in enumerate(cases):
        if kind == 'c-given-a':
            kw = dict(kw, p_c_given_a=p_ca)
        torch.manual_seed(100 + i)
        got = FlowMol._categorical_prior(kind, n, d, kw, a_0=a_0 if kind == 'c-given-a' else None)
        want = torch.from_numpy(g[f'case{i}'])
        assert got.shape == want.shape and got.dtype == want.dtype, (kind, got.shape)
        assert torch.equal(got, want), (kind, kw, (got - want).abs().max())
    with pytest.raises(ValueError):
        FlowMol._categorical_prior('marginal', 4, 5, {})                          # no distribution given
    with pytest.raises(ValueError):
        FlowMol._categorical_prior('marginal', 4, 5, {'p': [0.5, 0.5]})           # wrong number of categories


def test_shipped_marginals_and_simplex_projection():
    from flowmol_amd.model import load_marginal_dists, simplex_projection
    p_a, p_c, p_e, p_ca = load_marginal_dists('geom_full_kekulized')              # reference data/geom_full_kekulized/train_data_marginal_dists.pt
    assert p_a.shape == (10,) and p_c.shape == (6,) and p_e.shape == (4,) and p_ca.shape == (10, 6)
    for p in (p_a, p_c, p_e):
        assert abs(float(p.sum()) - 1.0) < 1e-5
    x = torch.randn(50, 7, generator=torch.Generator().manual_seed(0)) * 2
    y = simplex_projection(x)
    assert (y >= 0).all() and torch.allclose(y.sum(-1), torch.ones(50), atol=1e-6)
    inside = torch.softmax(x, -1)
    assert torch.allclose(simplex_projection(inside), inside, atol=1e-6)          # a point of the simplex is its own projection


def test_trajectory_blocks_batched_path_equals_per_frame_path():
    """traj_mol_blocks() (all frames in numpy, one batched SVD) against frame_moldata() + rigid_alignment() + mol_block() per frame:
    same atoms, charges and bonds; coordinates equal to the last printed digit's rounding."""
    from flowmol_amd.molecule import rigid_alignment
    torch.manual_seed(0)
    n, T = 9, 12
    amap = ['C', 'H', 'N', 'O', 'F']
    fr = {'x': torch.randn(T, n, 3) * 2, 'a': torch.randint(0, 7, (T, n), dtype=torch.int32), 'c': torch.randint(0, 7, (T, n), dtype=torch.int32),
          'e': (torch.randint(0, 5, (T, n * (n - 1) // 2), dtype=torch.int32) * (torch.rand(T, n * (n - 1) // 2) < 0.3)).int()}
    fr.update({k + '_1_pred': v[1:] for k, v in fr.items()})
    m = SampledMolecule(fr['x'][-1], fr['a'][-1].clamp(max=5), fr['c'][-1].clamp(max=5), fr['e'][-1].clamp(max=3), amap, fake_atoms=True, traj_frames=fr)
    for ep in (False, True):
        key = 'x_1_pred' if ep else 'x'
        new = m.traj_mol_blocks(ep_traj=ep)
        assert len(new) == fr[key].shape[0]
        for f, blk in enumerate(new):
            pos, sym, chg, bt, bs, bd = m.frame_moldata(f, ep_traj=ep)
            ref = mol_block(rigid_alignment(pos, fr[key][-1]), sym, chg, bs, bd, bt, name=f'frame {f}').splitlines()
            got = blk.splitlines()
            assert len(ref) == len(got)
            for a, b in zip(ref, got):
                if a != b:          # an atom line whose 4th decimal rounds the other way
                    assert a[30:] == b[30:] and all(abs(float(a[i:i + 10]) - float(b[i:i + 10])) <= 1.01e-4 for i in (0, 10, 20)), (a, b)


def _stub_rdkit(monkeypatch):
    """A minimal stand-in for the three RDKit classes build_rdkit_mol touches, so the RDKit branch of the boundary can run in an
    image without RDKit: records atoms (symbol, formal charge), bonds and conformer positions."""
    import sys
    import types

    class Atom:
        def __init__(self, sym): self.sym, self.chg = sym, 0
        def SetFormalCharge(self, q): self.chg = q

    class Mol:
        def __init__(self): self.atoms, self.bonds, self.conf = [], [], None
        def AddAtom(self, a): self.atoms.append(a)
        def AddBond(self, s, d, t): self.bonds.append((s, d, t))
        def GetMol(self): return self
        def GetNumAtoms(self): return len(self.atoms)
        def AddConformer(self, c): self.conf = c

    class Conformer:
        def __init__(self, n): self.pos = [None] * n
        def SetAtomPosition(self, i, p): self.pos[i] = p

    chem = types.ModuleType('rdkit.Chem')
    chem.Atom, chem.RWMol, chem.Conformer = Atom, Mol, Conformer
    chem.rdchem = types.SimpleNamespace(BondType=types.SimpleNamespace(SINGLE=1, DOUBLE=2, TRIPLE=3, AROMATIC=4))
    geom = types.ModuleType('rdkit.Geometry')
    geom.Point3D = lambda x, y, z: (x, y, z)
    rd = types.ModuleType('rdkit')
    rd.Chem, rd.Geometry = chem, geom
    for k, v in (('rdkit', rd), ('rdkit.Chem', chem), ('rdkit.Geometry', geom)):
        monkeypatch.setitem(sys.modules, k, v)


def test_traj_mols_follow_the_reference_build_switches(monkeypatch):
    """SampledMolecule.traj_mols / .ep_traj_mols (reference molecule_builder.py:76-84,156-214; read by test.py:235,251): present only when the
    molecule carries frames and the matching build switch is on; one RDKit molecule per frame with fake atoms shown and positions aligned
    to the final frame; without RDKit a clear error that names the RDKit-free equivalent."""
    from flowmol_amd.molecule import rigid_alignment
    torch.manual_seed(1)
    n, T = 7, 6
    amap = ['C', 'H', 'N', 'O', 'F']
    fr = {'x': torch.randn(T, n, 3) * 2, 'a': torch.randint(0, 7, (T, n), dtype=torch.int32), 'c': torch.randint(0, 6, (T, n), dtype=torch.int32),
          'e': (torch.randint(0, 5, (T, n * (n - 1) // 2), dtype=torch.int32) * (torch.rand(T, n * (n - 1) // 2) < 0.4)).int()}
    fr.update({k + '_1_pred': v[1:] for k, v in fr.items()})
    last = (fr['x'][-1], fr['a'][-1].clamp(max=5), fr['c'][-1], fr['e'][-1].clamp(max=3))
    plain = SampledMolecule(*last, amap, fake_atoms=True)
    for attr in ('traj_mols', 'ep_traj_mols'):
        assert not hasattr(plain, attr)                                        # no frames -> the reference never sets the attribute
    only_xt = SampledMolecule(*last, amap, fake_atoms=True, traj_frames=fr, build_ep_traj=False)
    assert not hasattr(only_xt, 'ep_traj_mols')
    m = SampledMolecule(*last, amap, fake_atoms=True, traj_frames=fr)
    try:
        import rdkit       # noqa: F401
        have = True
    except Exception:
        have = False
    if not have:
        with pytest.raises(ImportError, match='traj_mol_blocks'):
            m.traj_mols
        _stub_rdkit(monkeypatch)
        assert only_xt.traj_mols is not None
        for ep, attr in ((False, 'traj_mols'), (True, 'ep_traj_mols')):
            mols = getattr(m, attr)
            key = 'x_1_pred' if ep else 'x'
            assert len(mols) == fr[key].shape[0] and getattr(m, attr) is mols             # built once
            for f, mol in enumerate(mols):
                pos, sym, chg, bt, bs, bd = m.frame_moldata(f, ep_traj=ep)
                assert [a.sym for a in mol.atoms] == sym and [a.chg for a in mol.atoms] == [int(q) for q in chg]     # fake atoms shown (Sn), masked atoms Se
                assert [(s, d, t) for s, d, t in mol.bonds] == [(int(s), int(d), int(t)) for s, d, t in zip(bs, bd, bt)]
                want = rigid_alignment(pos, fr[key][-1])
                assert torch.allclose(torch.tensor(mol.conf.pos), want, atol=1e-6)
        unaligned = SampledMolecule(*last, amap, fake_atoms=True, traj_frames=fr, align_traj=False)
        assert torch.allclose(torch.tensor(unaligned.traj_mols[0].conf.pos), fr['x'][0], atol=1e-6)
    else:
        assert len(m.traj_mols) <= T and len(m.ep_traj_mols) <= T - 1


def test_lightning_shaped_checkpoint_is_read_without_lightning(tmp_path, monkeypatch):
    """load_pretrained on a checkpoint shaped like a real Lightning file (hyper_parameters an AttributeDict of a package this image lacks,
    pathlib.PosixPath data files, Lightning's bookkeeping keys, `vector_field.` prefixed weights): the resulting config equals the preset's."""
    import pickle
    from parity_util import write_lightning_shaped_checkpoint
    cfg = presets.flowmol3()
    sd = weights.synth_state_dict(cfg, 0)
    path = write_lightning_shaped_checkpoint(tmp_path, 'flowmol3', sd, 'flowmol3.yml')      # hyper_parameters derived from the reference's YAML, not from `cfg`
    with pytest.raises((ModuleNotFoundError, AttributeError, pickle.UnpicklingError)):
        torch.load(str(path), map_location='cpu', weights_only=False)          # the stock unpickler needs pytorch_lightning: the file really is Lightning-shaped
    hp, sd2 = read_checkpoint(path)
    assert type(hp) is dict and str(hp['n_atoms_hist_file']).endswith('geom_full_kekulized/train_data_n_atoms_histogram.pt')
    monkeypatch.setenv('FLOWMOL_MODELS_DIR', str(tmp_path))
    model = load_pretrained('flowmol3')
    assert model.cfg.to_dict() == cfg.to_dict()
    assert model.cfg.n_atoms_hist == 'geom_full_kekulized' and model.fake_atoms and model.default_n_timesteps == 250
    assert all(torch.equal(model._sd['vector_field.' + k], v) for k, v in sd.items())


YAML_PRESETS = {'flowmol3.yml': 'flowmol3', 'dev.yml': 'dev', 'geom_full_kekulized.yaml': 'geom_ctmc', 'geom_5_kekulized.yaml': 'geom_ctmc',
                'geom_full_aromatic.yaml': 'geom_arom', 'geom_5_aromatic.yaml': 'geom_arom'}


@pytest.mark.parametrize('yaml_name', sorted(YAML_PRESETS))
def test_presets_equal_the_hyper_parameters_derived_from_the_shipped_yamls(yaml_name):
    """Non-circular pin of the presets: tests/golden/hparams_from_yaml.json holds, for each of the reference's six shipped
    YAMLs, the FlowMol(...) keyword arguments exactly as model_from_config builds them (flowmol/model_utils/load.py:13-49) completed with
    FlowMol.__init__'s defaults (flowmol.py:29-55) -- generated in the build container by oracle/make_hparams_fixture.py from
    /root/reference/configs.  Each passes check_reference_hparams, and from_reference_hparams of it equals the preset that claims to be that
    YAML in every field except the name of the size histogram (dataset variant) and cosine_params the linear schedules never read."""
    from parity_util import hparams_from_yaml
    from flowmol_amd.model import check_reference_hparams
    hp = hparams_from_yaml(yaml_name)
    check_reference_hparams(hp)
    got = from_reference_hparams(hp).to_dict()
    want = presets.PRESETS[YAML_PRESETS[yaml_name]]().to_dict()
    diff = {k for k in got if got[k] != want[k]}
    assert diff <= {'n_atoms_hist', 'cosine_params'}, {k: (got[k], want[k]) for k in diff}
    assert all(v == 'linear' for v in got['schedule_type'].values())          # ... which is why cosine_params is dead
    if yaml_name == 'flowmol3.yml':
        assert not diff


def test_bench_flop_model_reproduces_the_survey_counts():
    """bench.py's reference FLOP model, derived from the model dimensions, against SURVEY.md section 8a/8d: flowmol3 2.437 M MAC per directed
    edge and 3.25 M per node (1.0844e10 FLOP at n = 47), geom_full_kekulized model 3.590e6 n(n-1) + 4.39e6 n FLOP (7.968e9 at n = 47),
    conv message 312,251 MAC per edge."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location('bench_mod', Path(__file__).resolve().parent.parent / 'bench.py')
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pe, pn = bench.reference_macs(presets.flowmol3())
    assert round(pe) == 2437218 and pn == 3249692
    assert abs(bench.network_flops(47, presets.flowmol3()) / 1.0844e10 - 1) < 1e-4
    pe, pn = bench.reference_macs(presets.geom_ctmc())
    assert abs(2 * pe / 3.590e6 - 1) < 1e-3 and abs(2 * pn / 4.39e6 - 1) < 2e-3
    assert abs(bench.network_flops(47, presets.geom_ctmc()) / 7.968e9 - 1) < 1e-4
    assert bench.conv_message_flops_per_edge(32) == 2 * 312251
    info = bench.host_cpu_info()
    assert info['logical_cpus'] >= 1


def test_bench_dry_run_plan_and_cost_sample():
    """`bench.py --gpus 8 --dry-run` prints the 8-rank plan of BASELINE configs[3] as one JSON line without touching a GPU (this container has
    none): 1024 molecules and the same cost per rank, payload bytes of the one all-gather; with the GEOM size distribution the LPT deal is
    balanced to ppm.  The CPU-baseline sample of a ragged workload is cost-representative, not its first molecules."""
    import importlib.util
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    line = subprocess.run([sys.executable, str(root / 'bench.py'), '--gpus', '8', '--dry-run'], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
    d = json.loads(line)
    assert d['dry_run'] and d['n_gpus'] == 8 and d['global_molecules'] == 8192 and len(d['ranks']) == 8
    assert all(r['molecules'] == 1024 and r['directed_edges'] == 1024 * 47 * 46 and r['gather_payload_bytes'] == 1024 * (47 * 14 + 1081) for r in d['ranks'])
    assert d['all_gather_total_bytes'] == 8 * d['all_gather_slot_bytes'] and d['all_gather_slot_bytes'] % 16 == 0
    pp = d['multi_gpu_parity_plan']            # the self-check a multi-GPU run performs before its timed region
    assert pp['world_size'] == 8 and pp['molecules'] == 64 == len(pp['sizes']) and sum(pp['molecules_per_rank']) == 64 and pp['n_timesteps'] == 12
    assert pp['all_gather_bytes'] == 8 * pp['all_gather_slot_bytes'] and max(pp['payload_bytes_per_rank']) <= pp['all_gather_slot_bytes']
    spec = importlib.util.spec_from_file_location('bench_mod2', root / 'bench.py')
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    all_sizes, parts, ragged = bench.job_sizes(8, 1024, None, 'geom_full_kekulized')
    cost = (all_sizes * (all_sizes - 1)).double()
    per_rank = torch.tensor([float(cost[p].sum()) for p in parts])
    assert ragged and sorted(torch.cat(parts).tolist()) == list(range(8192)) and float(per_rank.max() / per_rank.mean()) < 1.0001
    c5, _, _ = bench.job_sizes(1, 128, None, None)
    sample = bench._cost_sample(c5, 16)
    ratio = float((sample * (sample - 1)).double().mean() / (c5 * (c5 - 1)).double().mean())
    first = float((c5[:16] * (c5[:16] - 1)).double().mean() / (c5 * (c5 - 1)).double().mean())
    assert abs(ratio - 1) < 0.06 < abs(first - 1)                  # the first 16 molecules of c5 cost 23 % more than the workload's mean
    assert torch.equal(bench._cost_sample(torch.full((64,), 47), 16), torch.full((16,), 47))
