"""CPU-side parity of the HIP kernels through the host emulation (tests/emu): the SAME kernel and host
sources compiled with the host clang++ and executed lane-by-lane on the CPU.  This debugs index math, LDS
layouts, MFMA fragment maps and weight packing without a GPU; the real parity gate is tests/test_gpu_parity.py."""
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from flowmol_amd import _lib, presets, weights
from oracle import cpu_ref
from parity_util import forward_compare



@pytest.mark.parametrize('tile', [16, 32])
@pytest.mark.parametrize('name,sizes,t,prev', [('flowmol3', [4, 7, 2], 0.5, True), ('geom_ctmc', [6, 3], 0.4, False),
                                               ('flowmol3', [3, 1, 2, 1], 0.5, True),        # 1-atom molecules: no edges, no messages
                                               ('dev_narrow', [5, 3, 6], 0.5, True), ('dev_narrow', [4, 7], 0.0, False),    # 64/64-wide model on zero-padded tiles
                                               ('dev', [5, 3, 6], 0.5, True), ('dev', [4, 7, 1], 0.0, False),              # configs/dev.yml incl. use_dst_feats
                                               ('geom_arom', [6, 3], 0.4, False), ('flowmol3_arom', [4, 5], 0.5, True),    # explicit aromaticity: 5 bond types + mask
                                               ('arch_variants', [5, 1, 4], 0.5, True)])      # n_recycles=2, message_norm='mean', EdgeUpdate without distances, shared updater
def test_emulated_forward_matches_oracle(emu_lib, name, sizes, t, prev, tile):
    from flowmol_amd.engine import Engine
    cfg = presets.PRESETS[name]()
    sd = weights.synth_state_dict(cfg, 0)
    eng = Engine(cfg, sd, device='cpu', lib=emu_lib, tuning={'tile_edge': tile, 'tile_node': tile})     # 0 = chosen per batch (16 for these sizes)
    orc = cpu_ref.OracleVF(cfg, sd)
    errs, out, ref = forward_compare(eng, orc, cfg, torch.tensor(sizes), t, prev)
    bad = {k: v for k, v in errs.items() if not v < 2e-5}
    assert not bad, bad


def test_emulated_sample_api_short_trajectory(emu_lib):
    """The public API end to end on the emulation: prior, 3 steps incl. bootstrap, packaging."""
    import flowmol_amd as flowmol
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=emu_lib)
    torch.manual_seed(0)
    mols = model.sample(torch.tensor([4, 3]), n_timesteps=3, device='cpu')
    assert len(mols) == 2 and mols[0].positions.shape[1] == 3
    assert all('Se' not in m.atom_types for m in mols)
    # same seed on the oracle: identical RNG stream -> identical categorical outcome
    cfg = model.cfg
    orc = cpu_ref.OracleVF(cfg, weights.synth_state_dict(cfg, 0))
    batch = cpu_ref.build_batch(torch.tensor([4, 3]))
    torch.manual_seed(0)
    prior = orc.sample_prior(batch)
    with torch.no_grad():
        ref = orc.integrate(batch, prior, 3)
    got_a = torch.cat([m.a_1 for m in mols])
    assert torch.equal(got_a, ref['a_1'].argmax(-1))
    got_x = torch.cat([m.x_1 for m in mols])
    assert torch.allclose(got_x, ref['x_1'], atol=1e-5)


def test_cli_writes_sdf_and_trajectories_on_emulation(emu_lib, tmp_path):
    """test.py-equivalent CLI end to end (reference test.py:99-259) on the emulation."""
    from flowmol_amd import cli
    args = cli.parse_args(['--preset', 'qm9', '--n_mols', '3', '--n_atoms_per_mol', '4', '--n_timesteps', '3', '--max_batch_size', '2',
                           '--seed', '1', '--device', 'cpu', '--output_file', str(tmp_path / 'out.sdf')])
    mols, t = cli.run(args, engine_lib=emu_lib)
    assert len(mols) == 3
    txt = (tmp_path / 'out.sdf').read_text()
    assert txt.count('$$$$') == 3 and txt.count('M  END') == 3
    args = cli.parse_args(['--preset', 'qm9', '--n_mols', '1', '--n_atoms_per_mol', '4', '--n_timesteps', '3', '--xt_traj', '--ep_traj',
                           '--seed', '1', '--device', 'cpu', '--output_file', str(tmp_path / 'tr.sdf')])
    mols, t = cli.run(args, engine_lib=emu_lib)
    xt = (tmp_path / 'tr_0_xt.sdf').read_text()
    ep = (tmp_path / 'tr_0_ep.sdf').read_text()
    assert xt.count('$$$$') == 3 and ep.count('$$$$') == 2          # T frames / T-1 endpoint frames
    assert ' Se ' in xt                                               # masked atoms of early frames show up as Se
    # --metrics: valence stability / connectivity files like test.py:190-199 (qm9 ships no valency table -> choose one)
    import pickle
    args = cli.parse_args(['--preset', 'qm9', '--n_mols', '4', '--n_atoms_per_mol', '4', '--n_timesteps', '3', '--seed', '1',
                           '--device', 'cpu', '--metrics', '--metrics_dataset', 'geom_full_kekulized', '--n_subsets', '2',
                           '--output_file', str(tmp_path / 'm.sdf')])
    cli.run(args, engine_lib=emu_lib)
    met = pickle.loads((tmp_path / 'm_metrics.pkl').read_bytes())
    assert {'frac_atoms_stable', 'frac_mols_stable_valence', 'frac_atoms_stable_ci95', 'frac_connected'} <= set(met)
    assert 0.0 <= met['frac_atoms_stable'] <= 1.0 and 'frac_mols_stable_valence:' in (tmp_path / 'm_metrics.txt').read_text()
    with pytest.raises(FileNotFoundError):
        cli.run(cli.parse_args(['--preset', 'qm9', '--n_mols', '1', '--n_atoms_per_mol', '3', '--n_timesteps', '2', '--device', 'cpu',
                                '--metrics', '--output_file', str(tmp_path / 'n.sdf')]), engine_lib=emu_lib)


def test_sample_kwargs_tspan_cat_temp_and_prior_on_emulation(emu_lib):
    """integrate kwargs of the reference that flow through sample(): tspan, cat_temp_func
    (ctmc_vector_field.py:145-176) and a caller-supplied reference-format prior (flowmol.py:534-545)."""
    import flowmol_amd as flowmol
    import torch.nn.functional as F
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=emu_lib)
    n_atoms = torch.tensor([4, 3])
    torch.manual_seed(0)
    base, _ = model.sample(n_atoms, n_timesteps=4, device='cpu', return_tensors=True)
    torch.manual_seed(0)
    alt, _ = model.sample(n_atoms, n_timesteps=4, device='cpu', return_tensors=True, tspan=torch.linspace(0, 1, 4),
                          cat_temp_func=lambda t: 0.05)
    for k in 'xace':
        assert torch.equal(base[k], alt[k])
    # the same prior passed explicitly in the reference's format (one-hot floats, directed-edge e_0)
    cfg = model.cfg
    batch = cpu_ref.build_batch(n_atoms)
    torch.manual_seed(0)
    x0 = torch.randn(batch.N, 3)
    x0 = x0 - cpu_ref.segment_mean(x0, batch.node_batch_idx, batch.B)[batch.node_batch_idx]
    prior = {'x_0': x0, 'a_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_atom_types), 'c_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_charges),
             'e_0': cpu_ref.edge_prior(batch.upper_edge_mask, cfg.n_bond_types), 'fake_atoms': True}
    torch.manual_seed(0)
    _ = torch.randn(batch.N, 3)            # keep the RNG stream aligned with the run that drew its own prior
    via_prior, _ = model.sample(n_atoms, n_timesteps=4, device='cpu', return_tensors=True, prior=prior)
    assert torch.equal(via_prior['a'], base['a']) and torch.equal(via_prior['e'], base['e'])
    assert torch.allclose(via_prior['x'], base['x'], atol=1e-6)


@pytest.mark.parametrize('fname,dfm_type', [('integrate_qm9_gat.npz', 'gat'), ('integrate_qm9_sched.npz', 'campbell')])
def test_integrator_variants_on_emulation(emu_lib, golden_dir, fname, dfm_type):
    """SURVEY 8f rank 4 through the C ABI: tspan, 'decay' temperature, inv_temp_func and dfm_type 'gat'
    (ctmc_vector_field.py:71-95, 287-340, 463-510) against the reference's own run with its recorded noise."""
    from flowmol_amd.engine import Engine
    from parity_util import integrate_variant_golden
    cfg = presets.qm9()
    g = {k: torch.from_numpy(v) for k, v in np.load(golden_dir / fname).items()}
    eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cpu', lib=emu_lib)
    res = integrate_variant_golden(eng, cfg, g, dfm_type)
    assert res['a_flips'] == 0 and res['c_flips'] == 0 and res['e_flips'] == 0, res
    assert res['traj0_a_flips'] == 0 and res['traj0_a1_flips'] == 0, res
    assert res['x_rel'] < 1e-4 and res['traj0_x_rel'] < 1e-4, res       # BASELINE.json: 1e-4 relative coordinate error


def test_sample_accepts_variant_kwargs_on_emulation(emu_lib):
    """dfm_type / forward_weight_func / inv_temp_func flow through sample() like the reference's **kwargs; a model
    configured with dfm_type='gat' uses it by default; unknown kwargs and dfm types are errors."""
    import dataclasses
    import flowmol_amd as flowmol
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=emu_lib)
    n_atoms = torch.tensor([4, 3])
    torch.manual_seed(0)
    gat, _ = model.sample(n_atoms, n_timesteps=4, device='cpu', return_tensors=True, dfm_type='gat',
                          forward_weight_func=lambda t: 1.5, inv_temp_func=lambda t: 0.9)
    assert torch.isfinite(gat['x']).all()
    assert int(gat['a'].max()) <= model.cfg.n_atom_types and int(gat['e'].max()) <= model.cfg.n_bond_types
    model.cfg = dataclasses.replace(model.cfg, dfm_type='gat', forward_weight_schedule=1.5)
    torch.manual_seed(0)
    gat2, _ = model.sample(n_atoms, n_timesteps=4, device='cpu', return_tensors=True, inv_temp_func=lambda t: 0.9)
    for k in 'xace':
        assert torch.equal(gat[k], gat2[k])
    with pytest.raises(ValueError):
        model.sample(n_atoms, n_timesteps=4, device='cpu', dfm_type='euler')
    with pytest.raises(TypeError):
        model.sample(n_atoms, n_timesteps=4, device='cpu', not_an_argument=1)


@pytest.mark.parametrize('tag,dataset,arom,fake,preset', [('kek', 'geom_full_kekulized', False, True, 'flowmol3'),
                                                          ('arom', 'geom_5_aromatic', True, False, None)])
def test_stability_kernel_on_emulation(emu_lib, golden_dir, tag, dataset, arom, fake, preset):
    """the device valence-stability / connectivity counts vs the reference's check_stability
    verdicts (fixture made by oracle/make_golden.py from the reference's own functions)."""
    import dataclasses
    from flowmol_amd.engine import Engine
    from parity_util import stability_compare
    cfg = presets.flowmol3()
    if preset is None:      # an explicit-aromaticity model without fake atoms (geom_5_aromatic-style)
        cfg = dataclasses.replace(cfg, fake_atoms=False, n_bond_types=5, explicit_aromaticity=True)
    g = {k: torch.from_numpy(v) for k, v in np.load(golden_dir / 'stability.npz').items()}
    eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cpu', lib=emu_lib)
    assert stability_compare(eng, g, tag, dataset, arom, fake) == []


def test_batch_limits_are_reported(emu_lib):
    """A batch beyond the 31-bit gather offsets (2,097,151 nodes) or with an empty molecule is refused with the C ABI's
    error text instead of being mis-addressed (1-atom molecules are legal: no edges, no messages)."""
    from flowmol_amd.engine import Engine
    cfg = presets.qm9()
    eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cpu', lib=emu_lib)
    with pytest.raises(_lib.FlowMolHipError, match='batch too large'):
        eng.bind(torch.full((12000,), 181, dtype=torch.int64))
    with pytest.raises(_lib.FlowMolHipError, match='has 0 atoms'):
        eng.bind(torch.tensor([5, 0, 4]))
    eng.bind(torch.tensor([5, 1, 4]))          # still usable afterwards
    assert eng.N == 10 and eng.E == 32


def test_single_timepoint_returns_the_prior(emu_lib):
    """n_timesteps=1: linspace(0,1,1) has no step (ctmc_vector_field.py:205 loop body never runs); the result is the
    prior: centred Gaussian positions and mask tokens everywhere."""
    import flowmol_amd as flowmol
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=emu_lib).to('cpu')
    torch.manual_seed(0)
    out, _ = model.sample(torch.tensor([3, 4]), n_timesteps=1, return_tensors=True)
    assert (out['a'] == model.cfg.n_atom_types).all() and (out['c'] == model.cfg.n_charges).all() and (out['e'] == model.cfg.n_bond_types).all()
    assert torch.allclose(out['x'][:3].mean(0), torch.zeros(3), atol=1e-6) and torch.allclose(out['x'][3:].mean(0), torch.zeros(3), atol=1e-6)


@pytest.mark.parametrize('case', [0, 1, 2, 3, 4])
def test_emulated_ctmc_step_matches_reference_step(emu_lib, golden_dir, case):
    """fm_ctmc_step (emulated kernels) on the reference's own step() fixture: bit-exact tokens and Euler update."""
    from flowmol_amd.engine import Engine
    from parity_util import ctmc_step_golden
    cfg = presets.flowmol3()
    eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cpu', lib=emu_lib)
    g = {k: torch.from_numpy(v) for k, v in np.load(golden_dir / 'ctmc_step.npz').items()}
    res = ctmc_step_golden(eng, cfg, g, case)
    assert all(v == 0 for v in res.values()), res


def test_philox_noise_is_sharding_independent_and_well_distributed(emu_lib):
    """rng='philox' (SURVEY.md §8e performance mode): prior + CTMC noise from per-molecule counter-based streams inside the
    kernels.  (1) a molecule's result does not depend on which other molecules share its batch or where it sits in it;
    (2) same seed -> same result, other seed -> different; (3) the in-kernel draws have the right distributions
    (Exp(1) race = categorical sample of p; U(0,1) thresholds)."""
    import flowmol_amd as flowmol
    from flowmol_amd.engine import Engine, make_step_plan
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=emu_lib).to('cpu')
    sizes = torch.tensor([4, 6, 3, 5])
    full, _ = model.sample(sizes, n_timesteps=4, return_tensors=True, rng='philox', _philox=1234)
    again, _ = model.sample(sizes, n_timesteps=4, return_tensors=True, rng='philox', _philox=1234)
    other, _ = model.sample(sizes, n_timesteps=4, return_tensors=True, rng='philox', _philox=99)
    for k in 'xace':
        assert torch.equal(full[k], again[k])
    assert not torch.equal(full['x'], other['x'])
    noff = torch.cumsum(sizes, 0) - sizes
    pairs = sizes * (sizes - 1) // 2
    poff = torch.cumsum(pairs, 0) - pairs
    sub_ids = torch.tensor([3, 1])                       # molecules 3 and 1, in another order, as a batch of their own
    sub, _ = model.sample(sizes[sub_ids], n_timesteps=4, return_tensors=True, rng='philox', _philox=1234, _mol_ids=sub_ids)
    o_n = o_p = 0
    for i in sub_ids.tolist():
        n, u = int(sizes[i]), int(pairs[i])
        assert torch.equal(sub['a'][o_n:o_n + n], full['a'][noff[i]:noff[i] + n]) and torch.equal(sub['c'][o_n:o_n + n], full['c'][noff[i]:noff[i] + n])
        assert torch.equal(sub['e'][o_p:o_p + u], full['e'][poff[i]:poff[i] + u])
        torch.testing.assert_close(sub['x'][o_n:o_n + n], full['x'][noff[i]:noff[i] + n], rtol=1e-5, atol=1e-5)
        o_n += n; o_p += u
    # prior: centred, unit variance
    cfg = model.cfg
    eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cpu', lib=emu_lib)
    eng.bind(torch.full((40,), 50))
    x0 = eng.prior_philox(7)
    assert torch.allclose(x0.reshape(40, 50, 3).mean(1), torch.zeros(40, 3), atol=1e-5)
    assert abs(float(x0.std()) - 1.0) < 0.03 and abs(float((x0 ** 3).mean())) < 0.1
    # categorical draws: all rows masked, p fixed, last step (every row unmasks to its sampled endpoint) -> empirical frequencies = p
    eng.bind(torch.tensor([64, 64]))                      # 2 x 2016 pairs
    U, N = eng.U, eng.N
    p_e = torch.tensor([0.1, 0.2, 0.3, 0.4])
    # the kernel tempers with T: feed p^T renormalised so that softmax(log(.)/T) = p
    T = cfg.cat_temperature
    pt = p_e ** T / (p_e ** T).sum()
    plan = make_step_plan(250, cfg.stochasticity, 0.0, T, philox_seed=4242)
    sc = plan.scalars[-1]
    state = eng.prior_state(torch.zeros(N, 3))
    dst = {'x': torch.zeros(N, 3), 'a': torch.full((N, cfg.n_atom_types), 1.0 / cfg.n_atom_types),
           'c': torch.full((N, cfg.n_charges), 1.0 / cfg.n_charges), 'e': pt.repeat(U, 1).contiguous()}
    st_, d_ = eng._state_struct(state), eng._dst_struct(dst)
    import ctypes as C
    with eng._dev():
        eng._check(eng.lib.fm_ctmc_step(eng._ctx, eng._stream(), C.byref(st_), C.byref(d_), None, C.byref(sc), None), 'fm_ctmc_step')
    freq = torch.bincount(state['e_t'].long(), minlength=5).float() / U
    assert freq[4] == 0                                    # last step: nothing stays masked
    assert torch.allclose(freq[:4], p_e, atol=0.03), freq
    fa = torch.bincount(state['a_t'].long(), minlength=cfg.n_atom_types + 1).float() / N
    assert fa[-1] == 0 and float(fa.max()) < 3.0 / cfg.n_atom_types


def test_cosine_schedule_trajectory_on_emulation(emu_lib, golden_dir):
    """Cosine interpolant schedule end to end through the C ABI against the reference's own integrate() (recorded noise)."""
    from flowmol_amd.engine import Engine
    from parity_util import cosine_cfg, integrate_golden
    cfg = cosine_cfg(presets.qm9())
    g = {k: torch.from_numpy(v) for k, v in np.load(golden_dir / 'integrate_qm9_cosine.npz').items()}
    eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cpu', lib=emu_lib)
    res, state = integrate_golden(eng, cfg, g)
    assert res['a_flips'] == 0 and res['c_flips'] == 0 and res['e_flips'] == 0 and res['traj0_a_flips'] == 0, res
    assert res['x_rel'] < 1e-4 and res['traj0_x_rel'] < 1e-4, res


def test_endpoint_parameterization_on_emulation(emu_lib, golden_dir):
    """EndpointVectorField (vector_field.py:212-293, 388-569) through the C ABI against the reference's own module: dense-embedding
    forward and the Euler integration of all four modalities with the 'linear' inverse-temperature schedule."""
    from flowmol_amd.engine import Engine
    from parity_util import endpoint_cfg, endpoint_golden
    cfg = endpoint_cfg()
    g = {k: torch.from_numpy(v) for k, v in np.load(golden_dir / 'integrate_endpoint.npz').items()}
    eng = Engine(cfg, weights.synth_state_dict(cfg, 0), device='cpu', lib=emu_lib)
    res = endpoint_golden(eng, g)
    assert all(v < 1e-5 for v in res.values()), res


def test_endpoint_model_sample_api_on_emulation(emu_lib):
    """model.sample() of an endpoint-parameterised model: reference RNG order for the priors (device randn for x, then the a, c, e
    prior functions on the CPU generator), Euler integration, argmax packaging without a mask symbol; deterministic per seed."""
    import flowmol_amd as flowmol
    m = flowmol.FlowMol.from_preset('endpoint_small', _engine_lib=emu_lib).to('cpu')
    torch.manual_seed(0)
    mols = m.sample(torch.tensor([4, 3]), n_timesteps=4)
    torch.manual_seed(0)
    again = m.sample(torch.tensor([4, 3]), n_timesteps=4)
    assert [x.atom_types for x in mols] == [x.atom_types for x in again] and torch.equal(mols[0].positions, again[0].positions)
    assert mols[0].atom_type_map == ['C', 'H', 'N', 'O', 'F'] and all(s in mols[0].atom_type_map for s in mols[0].atom_types)
    with pytest.raises(_lib.FlowMolHipError, match='fm_forward_dense'):
        m.engine.forward(m.engine.prior_state(torch.zeros(m.engine.N, 3)), 0.5)        # token entry point refuses a dense model
    # the dataset-statistics priors (priors.py:67-98): atom types from the shipped QM9 marginals, charges conditioned on them
    cfg = presets.endpoint_small()
    cfg.prior_types = {'a': 'marginal', 'c': 'c-given-a', 'e': 'biased-simplex'}
    cfg.prior_kwargs = {'a': {'blur': 0.1}, 'c': {}, 'e': {'vertex_prob': 0.7}}
    m2 = flowmol.FlowMol(cfg.validate(), weights.synth_state_dict(cfg, 0), prefix='', _engine_lib=emu_lib).to('cpu')
    torch.manual_seed(1)
    mols2 = m2.sample(torch.tensor([5, 2]), n_timesteps=3)
    assert [x.num_atoms for x in mols2] == [5, 2] and all(torch.isfinite(x.positions).all() for x in mols2)
    # trajectory frames of an endpoint model (EndpointVectorField.integrate visualize=True, vector_field.py:412-466): frame 0 = prior,
    # one frame per step; the last state frame is the returned molecule, the run itself is unchanged by recording
    torch.manual_seed(0)
    tm = m.sample(torch.tensor([4, 3]), n_timesteps=4, xt_traj=True, ep_traj=True)
    assert torch.equal(tm[0].positions, mols[0].positions) and tm[1].atom_types == mols[1].atom_types
    fr = tm[0].traj_frames
    assert fr['x'].shape == (4, 4, 3) and fr['x_1_pred'].shape == (3, 4, 3) and fr['e'].shape[0] == 4 and fr['a_1_pred'].shape == (3, 4)
    assert torch.equal(fr['x'][-1], tm[0].positions)
    assert len(tm[0].traj_mol_blocks()) == 4 and len(tm[0].traj_mol_blocks(ep_traj=True)) == 3


@pytest.mark.parametrize('name,sizes', [('flowmol3', [4, 7, 2]), ('geom_ctmc', [6, 3]), ('dev_narrow', [5, 3])])
def test_split_precision_on_emulation(emu_lib, name, sizes):
    """Opt-in split precision (bf16x3 scalar / gate GEMMs of the edge messages and both EdgeUpdate layers on the emulated
    v_mfma_f32_16x16x32_bf16) against the f32 oracle: every stage within 5e-5, outputs within 2e-5 -- and measurably NOT the f32 path
    (the per-edge scalar messages differ from the exact path by more than its own error), i.e. the flag really selects other arithmetic."""
    from flowmol_amd.engine import Engine
    cfg = presets.PRESETS[name]()
    sd = weights.synth_state_dict(cfg, 0)
    orc = cpu_ref.OracleVF(cfg, sd)
    eng = Engine(cfg, sd, device='cpu', lib=emu_lib, precision='bf16x3')
    errs, out, ref = forward_compare(eng, orc, cfg, torch.tensor(sizes), 0.5, True)
    bad = {k: v for k, v in errs.items() if not (v < (2e-5 if k.startswith('out.') else 5e-5))}
    assert not bad, bad
    assert errs['conv0.msg.s'] > 2e-6          # f32 path: ~1e-6
    with pytest.raises(ValueError):
        Engine(cfg, sd, device='cpu', lib=emu_lib, precision='fp8')


def test_c_abi_error_behaviour_on_emulation(emu_lib, monkeypatch):
    """Error contract of include/flowmol_hip.h: every call returns 0 or a negative fm_status and leaves the text in fm_last_error;
    nothing throws or exits across the ABI; a context stays usable after a refused call.  (Raw ctypes calls, no Engine.)"""
    import ctypes as C
    from flowmol_amd._lib import fm_config
    from flowmol_amd.engine import Engine
    lib = emu_lib
    cfg = presets.qm9()
    sd = weights.synth_state_dict(cfg, 0)
    eng = Engine(cfg, sd, device='cpu', lib=lib)

    def err(ctx=None):
        return lib.fm_last_error(ctx).decode()
    # fm_create: null arguments, wrong ABI version, unsupported dimensions, a missing tensor
    ctx = C.c_void_p()
    assert lib.fm_create(None, None, 0, None, C.byref(ctx)) == -1 and 'null argument' in err()
    bad = fm_config(); bad.abi_version = _lib.FM_ABI_VERSION - 1
    dummy = (C.c_float * 4)()
    from flowmol_amd._lib import fm_tensor_desc
    descs = (fm_tensor_desc * 1)()
    assert lib.fm_create(C.byref(bad), descs, 0, dummy, C.byref(ctx)) == -1 and 'ABI version' in err()
    with pytest.raises(KeyError, match='token_embeddings.a.weight'):                   # the host layer checks the state dict first ...
        Engine(cfg, {k: v for k, v in sd.items() if k != 'token_embeddings.a.weight'}, device='cpu', lib=lib)
    import flowmol_amd.engine as engine_mod
    monkeypatch.setattr(engine_mod, 'state_dict_shapes', lambda c: {k: tuple(v.shape) for k, v in sd.items() if k != 'token_embeddings.a.weight'})
    monkeypatch.setattr(engine_mod, 'check_state_dict', lambda *a, **k: None, raising=False)
    with pytest.raises(_lib.FlowMolHipError, match='token_embeddings.a.weight'):       # ... and the library itself names a missing tensor (FM_ERR_WEIGHTS)
        Engine(cfg, {k: v for k, v in sd.items() if k != 'token_embeddings.a.weight'}, device='cpu', lib=lib)
    monkeypatch.undo()
    # call order: nothing bound yet -> FM_ERR_STATE
    st = eng.lib.fm_remove_com(eng._ctx, None, None)
    assert st < 0 and err(eng._ctx)
    # workspace: too small / misaligned -> FM_ERR_INVALID, context still usable
    n = torch.tensor([4, 3], dtype=torch.int32)
    need = C.c_size_t()
    assert lib.fm_workspace_bytes(eng._ctx, C.c_void_p(n.data_ptr()), 2, C.byref(need)) == 0 and need.value > 0
    ws = torch.empty(need.value + 512, dtype=torch.uint8)
    base = (ws.data_ptr() + 255) // 256 * 256
    assert lib.fm_batch_bind(eng._ctx, None, C.c_void_p(n.data_ptr()), 2, C.c_void_p(base), need.value - 1) == -1 and 'workspace' in err(eng._ctx)
    assert lib.fm_batch_bind(eng._ctx, None, C.c_void_p(n.data_ptr()), 2, C.c_void_p(base + 8), need.value) == -1 and 'aligned' in err(eng._ctx)
    eng.bind(torch.tensor([4, 3]))
    out = eng.forward(eng.prior_state(torch.zeros(eng.N, 3)), 0.0, bootstrap=True)
    assert torch.isfinite(out['x']).all()


def test_error_tracks_the_reference_rounding_sensitivity_on_emulation(emu_lib):
    """Ill-conditioned regime (all weight matrices x3: rounding differences grow ~10x per convolution; the f32 reference itself drifts
    percent-level from its own float64 evaluation by the last conv).  A fixed tolerance is meaningless there -- the claim that holds is
    that the kernels are as accurate as the reference's f32 arithmetic: every stage's error against the f32 oracle stays within a small
    multiple of the oracle's own f32-vs-f64 discrepancy at that stage."""
    from flowmol_amd.engine import Engine
    from parity_util import oracle_rounding_sensitivity, scaled_weights
    cfg = presets.flowmol3()
    sd = scaled_weights(weights.synth_state_dict(cfg, 0), 3.0)
    sizes = torch.tensor([4, 7, 2])
    sens = oracle_rounding_sensitivity(cfg, sd, sizes, 0.5, True)
    assert sens['conv5.s'] > 1e-3                      # the regime really is ill-conditioned (standard weights: 6e-7)
    eng = Engine(cfg, sd, device='cpu', lib=emu_lib)
    errs, out, ref = forward_compare(eng, cpu_ref.OracleVF(cfg, sd), cfg, sizes, 0.5, True)
    bad = {k: (v, sens[k]) for k, v in errs.items() if k in sens and not v <= max(5e-5, 8 * sens[k])}
    assert not bad, bad
    assert all(torch.isfinite(v).all() for v in out.values())


@pytest.mark.parametrize('small', [-1, 1, 2])
def test_mlp_tile_sizes_on_emulation(emu_lib, small):
    """The node- / pair-side MLP kernels (self-conditioning layers, output heads) exist with 64-row tiles (throughput), 16-row tiles
    (small batches) and -- node side, round 5 -- 4-row tiles on v_mfma_f32_4x4x1 (fm_k_mlp4: the smallest batches, chosen automatically while
    such tiles fit one per CU; mlp_small_tiles = 2 forces them): each against the oracle on the same batch, paired and separate launches."""
    from flowmol_amd.engine import Engine
    for pair in (-1, 1):
        cfg = presets.flowmol3()
        sd = weights.synth_state_dict(cfg, 0)
        eng = Engine(cfg, sd, device='cpu', lib=emu_lib, tuning={'mlp_small_tiles': small, 'pair_mlps': pair})
        errs, out, ref = forward_compare(eng, cpu_ref.OracleVF(cfg, sd), cfg, torch.tensor([5, 18, 2, 1]), 0.5, True, taps=False)
        bad = {k: v for k, v in errs.items() if not v < 1e-5}
        assert not bad, (small, pair, bad)


@pytest.mark.parametrize('name,sizes,prev', [('flowmol3', [5, 18, 2, 1], True), ('flowmol3', [5, 18, 2, 1], False), ('geom_ctmc', [6, 3, 9], False),
                                             ('arch_variants', [5, 9, 1, 4], True), ('dev_narrow', [5, 3], True)])
def test_pair_slab_hoist_on_emulation(emu_lib, name, sizes, prev):
    """Self-conditioned evaluations take the [rbf | ef] slab of the first scalar linear of the convolutions before the first molecule update
    from a per-pair table written by the self-conditioning edge kernel (FmMlpArgs::slabQ0 + the PQ instances of fm_k_edge_message;
    fm_config.pair_slab, ABI 6) instead of recomputing it for both directed edges of every pair: forced on (1; automatic only for large
    batches) and off (-1), both against the oracle on every stage; the switch really selects other arithmetic where it applies (another
    summation order) and nothing where it does not (no previous endpoint, models without self-conditioning).  arch_variants:
    n_recycles = 2 -- only the first pass is eligible."""
    from flowmol_amd.engine import Engine
    cfg = presets.PRESETS[name]()
    sd = weights.synth_state_dict(cfg, 0)
    outs = {}
    for flag in (-1, 1, 2, 3, 4):
        # 2: forced on with the launch shape of a large batch (separate node / pair launches, large tiles: the 32-row instance of the fused kernel);
        # 3: the PQ instance of the edge kernel on 64-row tiles; 4: the slab inside the SHARED node + pair launch on 64-row tiles (accepted
        # by fm_create, never executed before)
        tuning = {'pair_slab': flag} if flag < 2 else [{'pair_slab': 1, 'pair_mlps': -1, 'mlp_small_tiles': -1}, {'pair_slab': 1, 'tile_edge': 64},
                                                       {'pair_slab': 1, 'pair_mlps': 1, 'mlp_small_tiles': -1}][flag - 2]
        if flag >= 2 and name != 'flowmol3':
            continue
        eng = Engine(cfg, sd, device='cpu', lib=emu_lib, tuning=tuning)
        errs, out, ref = forward_compare(eng, cpu_ref.OracleVF(cfg, sd), cfg, torch.tensor(sizes), 0.5, prev)
        bad = {k: v for k, v in errs.items() if not v < 2e-5}
        assert not bad, (flag, bad)
        outs[flag] = {k: v.clone() for k, v in out.items()}
    for k in 'xace':
        torch.testing.assert_close(outs[1][k], outs[-1][k], rtol=1e-4, atol=2e-6)
    same = all(torch.equal(outs[1][k], outs[-1][k]) for k in 'xace')
    assert same != bool(prev and cfg.self_conditioning)


@pytest.mark.parametrize('name,sizes,t,prev', [('flowmol3', [5, 9, 2, 1, 11], 0.5, True), ('geom_ctmc', [6, 3], 0.4, False), ('dev_narrow', [5, 3], 0.5, True)])
def test_small_node_tiles_on_emulation(emu_lib, name, sizes, t, prev):
    """fm_config.tile_node = 4 / 8 / 12 / 20 (chosen automatically as the smallest tile that fits one per CU): the node kernel on 4 RG nodes per
    workgroup in a 16- or 32-row frame, its scalar GEMMs and the two 256 x 256 projections on v_mfma_f32_4x4x1_16B_f32 with quad-row packed
    weights (RG instances of fm_k_node_update; the emulation executes the instruction with the operand layout verified on the device,
    tools/ubench/mfma_4x4_layout.cpp).  Every stage against the oracle for every tile; since round 6 the 4-row GEMMs run the regular tiles' fma
    chains (fm_wave_gemm4), so 4 / 8 / 12 / 20-node tiles AND the regular 16-row tile give bit-identical results (on the MI355X too:
    profiles/r06m_*); models the instances do not exist for (dev_narrow) fall back to the frame's regular tile."""
    from flowmol_amd.engine import Engine
    cfg = presets.PRESETS[name]()
    sd = weights.synth_state_dict(cfg, 0)
    outs = {}
    for tile in (4, 8, 12, 16, 20):
        eng = Engine(cfg, sd, device='cpu', lib=emu_lib, tuning={'tile_node': tile, 'tile_edge': 16})
        errs, out, ref = forward_compare(eng, cpu_ref.OracleVF(cfg, sd), cfg, torch.tensor(sizes), t, prev, taps=(tile in (4, 20)))
        bad = {k: v for k, v in errs.items() if not v < 2e-5}
        assert not bad, (tile, bad)
        outs[tile] = {k: v.clone() for k, v in out.items()}
    for tile in (8, 12, 16, 20):
        for k in 'xace':
            assert torch.equal(outs[4][k], outs[tile][k]), (name, tile, k)


def test_traj_frames_reference_format_on_emulation(emu_lib, golden_dir):
    """SampledMolecule.traj_frames_reference() -- the lazy accessor that rebuilds the REFERENCE's `traj_frames` tensors (float one-hots incl. the
    mask column over all directed edges, ctmc_vector_field.py:188-202,267-283) from the compact token frames -- against the frame dicts the
    reference's own integrate(visualize=True) produced (tests/golden/traj_frames.npz; VERDICT r4 missing #2)."""
    import flowmol_amd as flowmol
    from parity_util import traj_frames_reference_compare
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(golden_dir / 'traj_frames.npz').items()}
    model = flowmol.FlowMol.from_preset('qm9', _engine_lib=emu_lib)
    res = traj_frames_reference_compare(model, g, 'cpu')
    assert res['x_rel'] < 1e-4 and res['molecules'] == 3 and res['categorical_cells_bit_equal'] > 0, res
    plain = model.sample(torch.tensor([3, 2]), n_timesteps=2, device='cpu')
    with pytest.raises(AttributeError):
        plain[0].traj_frames_reference()          # no frames were kept


def test_teacher_forced_decision_audit_on_emulation(emu_lib):
    """tests/parity_util.py: integrate_long_teacher_forced + audit_long_decisions (the gate of the 20-M-decision fixture on the GPU) on a tiny
    oracle-made fixture: every decision equals the oracle's (no events); a tampered reference token -- sampled or state -- is reported as an
    UNEXPLAINED difference (its margin is nowhere near a tie), so the audit cannot wave a real disagreement through."""
    from flowmol_amd.engine import Engine
    from parity_util import audit_long_decisions, integrate_long_teacher_forced, oracle_long_fixture
    cfg = presets.qm9()
    sd = weights.synth_state_dict(cfg, 0)
    g = oracle_long_fixture(cfg, sd, torch.tensor([3, 2, 4]), 5, 3, 4)
    eng = Engine(cfg, sd, device='cpu', lib=emu_lib)
    traj, probs = integrate_long_teacher_forced(eng, cfg, g, device='cpu')
    res = audit_long_decisions(cfg, g, traj, probs)
    assert res['sample_diffs'] == 0 and res['state_diffs'] == 0 and not res['events'] and not res['unexplained'], res
    assert res['decisions'] == 4 * (2 * 9 + 10)
    bad = dict(g)
    bad['traj.a1'] = g['traj.a1'].clone()
    bad['traj.a1'][1, 0] = (int(g['traj.a1'][1, 0]) + 1) % cfg.n_atom_types
    bad['traj.e'] = g['traj.e'].clone()
    bad['traj.e'][3, 2] = (int(g['traj.e'][3, 2]) + 1) % cfg.n_bond_types          # new state of step 2
    res = audit_long_decisions(cfg, bad, traj, probs)
    kinds = sorted(e['kind'] for e in res['unexplained'])
    assert kinds == ['sampled token', 'state token'] and res['sample_diffs'] == 1 and res['state_diffs'] == 1, res


@pytest.mark.parametrize('last', [False, True])
def test_ctmc_kernel_1024_thread_instance_on_emulation(emu_lib, last):
    """fm_k_ctmc_fused<1024> -- the instance for batches of a few molecules (one workgroup per molecule and modality is all the parallelism the
    kernel has; chosen when 4 B workgroups do not fill the chip) -- against the oracle's campbell_step on two molecules of 26 / 25 atoms with
    random endpoint probabilities and a half-unmasked state: every new state token and sampled token bit-exact, Euler step bit-exact."""
    import torch.nn.functional as F
    from flowmol_amd.engine import Engine, StepNoise, make_step_plan
    cfg = presets.flowmol3()
    sd = weights.synth_state_dict(cfg, 0)
    eng = Engine(cfg, sd, device='cpu', lib=emu_lib)
    n_atoms = torch.tensor([26, 25])
    eng.bind(n_atoms)
    batch = cpu_ref.build_batch(n_atoms)
    orc = cpu_ref.OracleVF(cfg, sd)
    g = torch.Generator().manual_seed(5)
    N, U = eng.N, eng.U
    T, s_idx = 30, 30 - 1 if last else 12
    plan = make_step_plan(T, cfg.stochasticity, cfg.high_confidence_threshold, cfg.cat_temperature)
    sc = plan.scalars[s_idx - 1]
    dst = {'x': torch.randn(N, 3, generator=g), 'a': torch.softmax(3 * torch.randn(N, cfg.n_atom_types, generator=g), -1),
           'c': torch.softmax(3 * torch.randn(N, cfg.n_charges, generator=g), -1), 'e': torch.softmax(3 * torch.randn(U, cfg.n_bond_types, generator=g), -1)}
    tok = {}
    for k, rows, K in (('a', N, cfg.n_atom_types), ('c', N, cfg.n_charges), ('e', U, cfg.n_bond_types)):
        t_ = torch.randint(0, K, (rows,), generator=g)
        t_[torch.rand(rows, generator=g) < 0.5] = K                      # half of the rows masked
        tok[k] = t_
    x_t = torch.randn(N, 3, generator=g)
    torch.manual_seed(9)
    nz = StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, last, 'cpu')
    state = eng.make_state(x_t, tok['a'], tok['c'], tok['e'])
    i32 = dict(dtype=torch.int32)
    smp = {'a1': torch.zeros(N, **i32), 'c1': torch.zeros(N, **i32), 'e1': torch.zeros(U, **i32)}
    eng.ctmc_step(state, {k: v.contiguous() for k, v in dst.items()}, nz, sc, smp)
    # the oracle's campbell_step with the same draws (tape order per modality: q, u1, u2)
    t = plan.t
    alpha, alpha_p = cpu_ref.alpha_tables(t.clone())
    dt = t[s_idx] - t[s_idx - 1]
    m = batch.upper_edge_mask
    for fi, (k, K, bidx) in enumerate((('a', cfg.n_atom_types, batch.node_batch_idx), ('c', cfg.n_charges, batch.node_batch_idx), ('e', cfg.n_bond_types, batch.edge_batch_idx[m])), start=1):
        tape = [getattr(nz, f'q_{k}'), getattr(nz, f'u1_{k}')] + ([] if last else [getattr(nz, f'u2_{k}')])
        p = F.softmax(torch.log(dst[k]) / cfg.cat_temperature, dim=-1)
        xt1h, x11h = orc.campbell_step(p, tok[k], cfg.stochasticity, cfg.high_confidence_threshold, alpha[s_idx - 1][fi], alpha_p[s_idx - 1][fi], dt,
                                       2, K + 1, K, last, bidx, cpu_ref.TapeNoise(tape))
        assert torch.equal(state[f'{k}_t'].long(), xt1h.argmax(-1)), k
        assert torch.equal(smp[f'{k}1'].long(), x11h.argmax(-1)), k
    vf = alpha_p[s_idx - 1][0] / (1 - alpha[s_idx - 1][0]) * (dst['x'] - x_t)
    assert torch.equal(state['x_t'], x_t + dt * vf * 1.0)


@pytest.mark.parametrize('tile', [16, 32])
@pytest.mark.parametrize('name,sizes', [('flowmol3', [4, 7, 2]), ('geom_ctmc', [6, 3])])
def test_three_term_split_precision_on_emulation(emu_lib, name, sizes, tile):
    """Opt-in THREE-term split (precision='bf16x6', edge-message kernel only: hi + mid + lo bf16 operands, six products per term on the emulated
    v_mfma_f32_16x16x32_bf16): f32-CLASS accuracy -- every stage inside the f32 kernels' own gate (2e-5 per stage, 1e-5 on the outputs; the two-term
    mode needs 5e-5) -- and it really is other arithmetic than the exact f32 path (the per-edge messages differ in their last bits)."""
    from flowmol_amd.engine import Engine
    cfg = presets.PRESETS[name]()
    sd = weights.synth_state_dict(cfg, 0)
    orc = cpu_ref.OracleVF(cfg, sd)
    errs = {}
    outs = {}
    for prec in ('f32', 'bf16x6'):
        eng = Engine(cfg, sd, device='cpu', lib=emu_lib, precision=prec, tuning={'tile_edge': tile})
        errs[prec], out, ref = forward_compare(eng, orc, cfg, torch.tensor(sizes), 0.5, True)
        outs[prec] = {k: v.clone() for k, v in out.items()}
    bad = {k: v for k, v in errs['bf16x6'].items() if not (v < (1e-5 if k.startswith('out.') else 2e-5))}
    assert not bad, bad
    worst = max(errs['bf16x6'][k] / max(errs['f32'][k], 2e-7) for k in errs['f32'])
    assert worst < 3, (worst, errs)                                  # per stage within a small factor of the f32 kernels' own error
    assert not all(torch.equal(outs['f32'][k], outs['bf16x6'][k]) for k in 'xace')


@pytest.mark.parametrize('name,sizes', [('flowmol3', [4, 7, 2]), ('geom_ctmc', [6, 3]), ('dev_narrow', [5, 3])])
def test_half_split_precision_on_emulation(emu_lib, name, sizes):
    """Opt-in precision='f16x3' (round 5): the two-plane split kernels with IEEE-half planes (hi + lo half = 22 mantissa bits, three products on the emulated
    v_mfma_f32_16x16x32_f16; edge messages, node kernel, EdgeUpdate): every stage inside the f32 kernels' own gate (2e-5 per stage, 1e-5 on the outputs), i.e.
    f32-class where the bf16 planes need 5e-5 -- and not the exact f32 path bit for bit."""
    from flowmol_amd.engine import Engine
    cfg = presets.PRESETS[name]()
    sd = weights.synth_state_dict(cfg, 0)
    orc = cpu_ref.OracleVF(cfg, sd)
    errs, outs = {}, {}
    for prec in ('f32', 'bf16x3', 'f16x3'):
        eng = Engine(cfg, sd, device='cpu', lib=emu_lib, precision=prec)
        errs[prec], out, ref = forward_compare(eng, orc, cfg, torch.tensor(sizes), 0.5, True)
        outs[prec] = {k: v.clone() for k, v in out.items()}
    bad = {k: v for k, v in errs['f16x3'].items() if not (v < (1e-5 if k.startswith('out.') else 2e-5))}
    assert not bad, bad
    assert errs['f16x3']['conv0.msg.s'] < 0.5 * errs['bf16x3']['conv0.msg.s'], (errs['f16x3']['conv0.msg.s'], errs['bf16x3']['conv0.msg.s'])
    assert not all(torch.equal(outs['f32'][k], outs['f16x3'][k]) for k in 'xace')


def _one_molecule_inputs(cfg, n, seed):
    g = torch.Generator().manual_seed(seed)
    U = n * (n - 1) // 2
    d = dict(x=torch.randn(n, 3, generator=g) * 1.5, a=torch.randint(0, cfg.n_atom_types + 1, (n,), generator=g),
             c=torch.randint(0, cfg.n_charges + 1, (n,), generator=g), e=torch.randint(0, cfg.n_bond_types + 1, (U,), generator=g))
    d['px'] = d['x'] + 0.3 * torch.randn(n, 3, generator=g)
    d['pa'] = torch.softmax(torch.randn(n, cfg.n_atom_types, generator=g), -1)
    d['pc'] = torch.softmax(torch.randn(n, cfg.n_charges, generator=g), -1)
    d['pe'] = torch.softmax(torch.randn(U, cfg.n_bond_types, generator=g), -1)
    return d


def _forward_per_molecule(eng, cfg, mols, t=0.4):
    n_atoms = torch.tensor([m['x'].shape[0] for m in mols])
    eng.bind(n_atoms)
    cat = lambda k: torch.cat([m[k] for m in mols])
    st = eng.make_state(cat('x'), cat('a'), cat('c'), cat('e'))
    pv = {'x': cat('px'), 'a': cat('pa'), 'c': cat('pc'), 'e': cat('pe')} if cfg.self_conditioning else None
    out = eng.forward(st, t, prev=pv, bootstrap=False, remove_com=True)
    pr = n_atoms * (n_atoms - 1) // 2
    no, po = torch.cumsum(n_atoms, 0) - n_atoms, torch.cumsum(pr, 0) - pr
    return [{k: out[k][(po[i] if k == 'e' else no[i]):(po[i] + pr[i] if k == 'e' else no[i] + n_atoms[i])].clone() for k in 'xace'} for i in range(len(mols))]


@pytest.mark.parametrize('name,n', [('flowmol3', 34), ('geom_ctmc', 21), ('dev', 9)])
def test_canonical_arithmetic_a_molecules_bits_do_not_depend_on_its_batch(emu_lib, name, n):
    """fm_config.canonical (default): the f32 summation order of everything computed for a molecule is a function of the molecule alone, as every reduction of
    the reference is per molecule (gvp.py:491-492, ctmc_utils.py:11-20, vector_field.py:347-350) -- edge-message tiles start at the molecule's first edge
    row, in-edges are summed in 16-row chunks counted from it, LayerNorm / gate sums have one order for every tile height.  One network evaluation of a
    molecule ALONE, first / in the middle / last in other batches, under 16- vs 32-row edge and node tiles, 4 .. 20-node tiles and 4-row MLPs must give identical bits (n = 34: 33
    in-edges per destination span 3 chunks and every alignment of the molecule's first row).  The emulation executes the kernels' own index and
    reduction code lane by lane; the GPU suite repeats this at 1024 x 47 atoms over 12 integration steps."""
    from flowmol_amd.engine import Engine
    cfg = presets.PRESETS[name]()
    sd = weights.synth_state_dict(cfg, 0)
    A = _one_molecule_inputs(cfg, n, 1)
    others = [_one_molecule_inputs(cfg, k, 10 + k) for k in (5, 12, 3)]
    ref = None
    batches = (([A], 0), ([A, others[0], others[1]], 0), ([others[0], A, others[2]], 1), ([others[1], others[2], others[0], A], 3))
    for tuning, sel in (({}, batches[:2]), ({'tile_edge': 32, 'tile_node': 32}, batches), ({'tile_edge': 16, 'tile_node': 32}, batches), ({'tile_edge': 32, 'tile_node': 16}, batches),
                        # round 6: the small-batch kernels run the regular tiles' fma chains -- 4 / 8 / 12 / 20-node tiles (fm_wave_gemm4), 4-row node MLPs
                        # (fm_rows4_linear; mlp_small_tiles = 2) and 16-row MLP tiles (= 1): the molecule alone and in the middle of a batch
                        ({'tile_node': 4}, batches[::2]), ({'tile_node': 8}, batches[:1]), ({'tile_node': 12, 'mlp_small_tiles': 2}, batches[2:3]), ({'tile_node': 20}, batches[:1]),
                        ({'mlp_small_tiles': 2}, batches[::2]), ({'mlp_small_tiles': 1, 'tile_node': 4}, batches[:1])):
        eng = Engine(cfg, sd, device='cpu', lib=emu_lib, tuning=tuning)
        for batch, idx in sel:
            got = _forward_per_molecule(eng, cfg, batch)[idx]
            ref = ref or got
            for k in 'xace':
                assert torch.equal(got[k], ref[k]), (tuning, len(batch), idx, k, float((got[k] - ref[k]).abs().max()))
        eng.close()
