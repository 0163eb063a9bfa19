"""Shared helpers of the parity tests: run the engine (HIP on the GPU box, or the host emulation in
tests/test_emu_parity.py) and the CPU oracle on the same seeded inputs and compare stage by stage."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle import cpu_ref


def rand_tokens(n, k, frac_masked, gen):
    t = torch.randint(0, k, (n,), generator=gen)
    t[torch.rand(n, generator=gen) < frac_masked] = k
    return t


def onehots(cfg, batch, a, c, eu):
    m = batch.upper_edge_mask
    e = torch.zeros(batch.E, cfg.n_bond_types + 1)
    e[m] = F.one_hot(eu, cfg.n_bond_types + 1).float()
    e[~m] = F.one_hot(eu, cfg.n_bond_types + 1).float()
    return F.one_hot(a, cfg.n_atom_types + 1).float(), F.one_hot(c, cfg.n_charges + 1).float(), e


def edge_perm(eng, batch):
    """internal (destination-major) edge index -> reference edge index"""
    e_src, e_dst = eng.query('e_src').cpu().long(), eng.query('e_dst').cpu().long()
    N = batch.N
    ref = torch.full((N * N,), -1, dtype=torch.int64)
    ref[batch.src * N + batch.dst] = torch.arange(batch.E)
    perm = ref[e_src * N + e_dst]
    assert (perm >= 0).all()
    return perm


def forward_compare(eng, orc, cfg, n_atoms, t_val, with_prev, seed=3, frac_masked=0.4, taps=True, dtype=torch.float32):
    """Returns {stage: relative error (max abs diff / max abs ref)} for every tap and the outputs.  dtype = torch.float64 with an oracle whose
    parameters are float64 (oracle_f64): the kernels' error against the EXACT result instead of against the f32 reference arithmetic."""
    dev = eng.device
    batch = cpu_ref.build_batch(n_atoms)
    eng.bind(n_atoms)
    gen = torch.Generator().manual_seed(seed)
    N, U, E = eng.N, eng.U, eng.E
    a = rand_tokens(N, cfg.n_atom_types, frac_masked, gen)
    c = rand_tokens(N, cfg.n_charges, frac_masked, gen)
    eu = rand_tokens(U, cfg.n_bond_types, frac_masked, gen)
    x = torch.randn(N, 3, generator=gen) * 1.5
    prev = None
    if with_prev and cfg.self_conditioning:
        prev = {'x': x + 0.3 * torch.randn(N, 3, generator=gen),
                'a': torch.softmax(torch.randn(N, cfg.n_atom_types, generator=gen), -1),
                'c': torch.softmax(torch.randn(N, cfg.n_charges, generator=gen), -1),
                'e': torch.softmax(torch.randn(U, cfg.n_bond_types, generator=gen), -1)}
    a1h, c1h, e1h = onehots(cfg, batch, a, c, eu)
    orc.taps = {}
    try:
        torch.set_default_dtype(dtype)
        with torch.no_grad():
            ref = orc.forward(batch, x.to(dtype), a1h.to(dtype), c1h.to(dtype), e1h.to(dtype), torch.full((batch.B,), float(t_val), dtype=dtype),
                              prev=None if prev is None else {k: v.to(dtype) for k, v in prev.items()}, apply_softmax=True, remove_com=True)
    finally:
        torch.set_default_dtype(torch.float32)
    taps_o, orc.taps = orc.taps, None
    state = eng.make_state(x, a, c, eu)
    V = cfg.n_vec_channels
    bufs = {}
    prev_d = {k: v.to(dev).contiguous() for k, v in prev.items()} if prev is not None else None
    bootstrap = (t_val == 0) and prev is None
    # The evaluation's LAST EdgeUpdate may run the edge head as its epilogue and then does not store its rows (fm_k_edge_update<32, false, true>); asking for
    # that tap selects the separate kernels instead.  Whether the fused kernel runs is the ENGINE's decision (fm_engine.cpp:evaluate) and is asked of the engine
    # (no copy of its predicate here): a tap-free pass under the profiler -- did `edge_update_head` launch?  Where it did, the tap is not requested,
    # `out.e` is what checks the fused kernel, and the instrumented pass below must launch it again.
    eng.profile(True)
    eng.forward(state, t_val, prev=prev_d, bootstrap=bootstrap, remove_com=True)
    eng.synchronize()
    fused_head = eng.profile_get('edge_update_head')[1] > 0
    eng.profile(False)
    if taps:
        for i in range(cfg.n_convs):
            bufs[f'conv{i}.s'] = torch.zeros(N, 256, device=dev)
            bufs[f'conv{i}.v'] = torch.zeros(N, 3, V, device=dev)
            bufs[f'conv{i}.agg.s'] = torch.zeros(N, 256, device=dev)
            bufs[f'conv{i}.agg.v'] = torch.zeros(N, 3, V, device=dev)
            if cfg.update_schedule()[i] >= 0:
                bufs[f'upd{i}.x'] = torch.zeros(N, 3, device=dev)
                last = i == cfg.n_convs - 1 and getattr(cfg, 'n_recycles', 1) <= 1
                if not (last and fused_head):
                    bufs[f'upd{i}.ef'] = torch.zeros(E, 128, device=dev)
        bootstrap_ = (t_val == 0) and prev is None and cfg.self_conditioning
        first = 'sc' if (prev is not None or bootstrap_) else 'embed'   # the bootstrap result feeds the SC layer
        bufs[f'{first}.s'] = torch.zeros(N, 256, device=dev)
        bufs[f'{first}.ef'] = torch.zeros(E, 128, device=dev)
        bufs['conv0.msg.s'] = torch.zeros(E, 256, device=dev)
        bufs['conv0.msg.v'] = torch.zeros(E, 3, V, device=dev)
    eng.profile(True)
    out = eng.forward(state, t_val, prev=prev_d, bootstrap=bootstrap, remove_com=True, taps=bufs)
    eng.synchronize()
    last_ef_tapped = f'upd{cfg.n_convs - 1}.ef' in bufs          # (recycled stacks tap every pass's rows: the tap un-fuses the last pass on purpose)
    assert (eng.profile_get('edge_update_head')[1] > 0) == (fused_head and not last_ef_tapped), 'the instrumented pass must run the kernels the plain pass runs'
    eng.profile(False)
    perm = edge_perm(eng, batch) if taps else None
    errs = {}

    def rel(got, want):
        got = got.detach().cpu()
        if torch.isnan(got).any():
            return float('nan')
        if got.shape[-1] > want.shape[-1] and got.dim() == 2:      # narrow model in a 256 / 128-column tile: the padding must be exactly 0
            if float(got[:, want.shape[-1]:].abs().max()) != 0.0:
                return float('inf')
            got = got[:, :want.shape[-1]]
        return float((got.to(want.dtype) - want).abs().max() / want.abs().max().clamp(min=1e-20))
    for k, v in bufs.items():
        if k.endswith('.msg.s'):
            want = taps_o['conv0.msg2.s'][perm]
        elif k.endswith('.msg.v'):
            want = taps_o['conv0.msg2.v'][perm].transpose(1, 2)
        elif k.endswith('.ef'):
            want = taps_o[k][perm]
        elif k.endswith('.v'):
            want = taps_o[k].transpose(1, 2)
        else:
            want = taps_o[k]
        errs[k] = rel(v, want)
    for k in 'xace':
        errs['out.' + k] = rel(out[k], ref[k])
    return errs, out, ref


def integrate_golden(eng, cfg, g, chunk=8, device=None):
    """Free-running trajectory on the engine with the reference's recorded noise (golden fixture g)."""
    from flowmol_amd.engine import StepNoise, make_step_plan
    device = device or eng.device
    eng.bind(g['n_atoms'])
    T = int(g['T'])
    tape = [g[k] for k in sorted(k for k in g if k.startswith('noise.'))]
    plan = make_step_plan(T, cfg.stochasticity, cfg.high_confidence_threshold, cfg.cat_temperature,
                          schedule_type=cfg.schedule_type, cosine_params=cfg.cosine_params)
    state = eng.prior_state(g['x_0'])
    pos = [0]

    def noise_for_step(i, last):
        nz, pos[0] = StepNoise.from_tape(tape, pos[0], last, device)
        return nz
    traj = {'x': torch.zeros(T - 1, eng.N, 3, device=device), 'a': torch.zeros(T - 1, eng.N, dtype=torch.int32, device=device)}
    eng.integrate(state, plan, noise_for_step, chunk=chunk, traj=traj)
    assert pos[0] == len(tape)
    res = {
        'a_flips': int((state['a_t'].cpu().long() != g['a_1']).sum()),
        'c_flips': int((state['c_t'].cpu().long() != g['c_1']).sum()),
        'e_flips': int((state['e_t'].cpu().long() != g['e_1_upper']).sum()),
        'x_rel': float((state['x_t'].cpu() - g['x_1']).abs().max() / g['x_1'].abs().max()),
    }
    n0 = int(g['n_atoms'][0])
    res['traj0_x_rel'] = float((traj['x'][:, :n0].cpu() - g['traj0.x'][1:]).abs().max() / g['traj0.x'].abs().max())
    res['traj0_a_flips'] = int((traj['a'][:, :n0].cpu().long() != g['traj0.a'][1:]).sum())
    return res, state


# the schedule used by oracle/make_golden.py:gen_integrate_variant (fixtures integrate_qm9_gat / integrate_qm9_sched)
def variant_inv_temp(t):
    return 1 - 0.25 * t


def variant_cfg(cfg):
    """The preset with the variant fixtures' sampling schedules: 'decay' temperature, 'beta' forward weight."""
    import dataclasses
    return dataclasses.replace(cfg, cat_temperature_schedule='decay', cat_temp_decay_max=0.8, cat_temp_decay_a=2,
                               forward_weight_schedule='beta')


def integrate_variant_golden(eng, cfg, g, dfm_type, chunk=4, device=None):
    """Engine run of the variant fixtures (non-uniform tspan, decay temperature, inverse-temperature function,
    dfm_type campbell|gat) with the reference's recorded noise."""
    from flowmol_amd.engine import StepNoise, cat_temp_schedule, forward_weight_schedule, make_step_plan
    device = device or eng.device
    vcfg = variant_cfg(cfg)
    eng.bind(g['n_atoms'])
    tape = [g[k] for k in sorted(k for k in g if k.startswith('noise.'))]
    plan = make_step_plan(0, vcfg.stochasticity, vcfg.high_confidence_threshold, cat_temp_schedule(vcfg), tspan=g['tspan'],
                          dfm_type=dfm_type, forward_weight_func=forward_weight_schedule(vcfg), inv_temp_func=variant_inv_temp)
    n_steps = len(plan.scalars)
    state = eng.prior_state(g['x_0'])
    pos = [0]

    def noise_for_step(i, last):
        nz, pos[0] = StepNoise.from_tape(tape, pos[0], last, device, dfm_type=dfm_type)
        return nz
    i32 = dict(dtype=torch.int32, device=device)
    traj = {'x': torch.zeros(n_steps, eng.N, 3, device=device), 'a': torch.zeros(n_steps, eng.N, **i32),
            'a1': torch.zeros(n_steps, eng.N, **i32)}
    eng.integrate(state, plan, noise_for_step, chunk=chunk, traj=traj)
    assert pos[0] == len(tape)
    n0 = int(g['n_atoms'][0])
    return {
        'a_flips': int((state['a_t'].cpu().long() != g['a_1']).sum()),
        'c_flips': int((state['c_t'].cpu().long() != g['c_1']).sum()),
        'e_flips': int((state['e_t'].cpu().long() != g['e_1_upper']).sum()),
        'x_rel': float((state['x_t'].cpu() - g['x_1']).abs().max() / g['x_1'].abs().max()),
        'traj0_x_rel': float((traj['x'][:, :n0].cpu() - g['traj0.x'][1:]).abs().max() / g['traj0.x'].abs().max()),
        'traj0_a_flips': int((traj['a'][:, :n0].cpu().long() != g['traj0.a'][1:]).sum()),
        'traj0_a1_flips': int((traj['a1'][:, :n0].cpu().long() != g['traj0.a_1_pred']).sum()),
    }


def stability_compare(eng, g, tag, dataset, arom, fake):
    """fm_stability on the fixture's token molecules vs (a) the reference's own check_stability verdicts stored in
    tests/golden/stability.npz and (b) the oracle's bond-graph components.  Returns the list of mismatches."""
    import torch.nn.functional as F
    from flowmol_amd import metrics
    from oracle import cpu_ref
    atom_map = ['C', 'H', 'N', 'O', 'F', 'P', 'S', 'Cl', 'Br', 'I']
    nb = 5 if arom else 4
    table, ar = metrics.load_valency_table(dataset)
    enc = metrics.encode_valency_table(table, atom_map, 6, arom)
    n_atoms = g[f'{tag}.n_atoms']
    eng.bind(n_atoms)
    state = eng.make_state(torch.zeros(eng.N, 3), g[f'{tag}.a'], g[f'{tag}.c'], g[f'{tag}.e'])
    got = eng.stability(state, enc, len(atom_map) if fake else -1, arom).cpu()
    bad = []
    no = po = 0
    for i, n in enumerate(n_atoms.tolist()):
        u = n * (n - 1) // 2
        a, c, e = g[f'{tag}.a'][no:no + n], g[f'{tag}.c'][no:no + n], g[f'{tag}.e'][po:po + u]
        no += n; po += u
        pos, sym, chg, bt, bs, bd = cpu_ref.extract_moldata(torch.zeros(n, 3), F.one_hot(a, len(atom_map) + (2 if fake else 1)).float(),
                                                            F.one_hot(c, 6).float(), torch.cat([F.one_hot(e, nb + 1).float()] * 2), n,
                                                            atom_map, fake, nb)
        ncomp, largest = cpu_ref.bond_graph_components(len(sym), bs, bd)
        n_stable, _, n_real = g[f'{tag}.expect'][i].tolist()
        want = [n_stable, n_real, ncomp, largest]
        if got[i].tolist() != want:
            bad.append((i, got[i].tolist(), want))
    return bad


def ctmc_step_golden(eng, cfg, g, case, device=None):
    """fm_ctmc_step on the inputs and recorded RNG draws of tests/golden/ctmc_step.npz (the reference's own
    CTMCVectorField.step with a fixed endpoint prediction; purity-sampling edge cases by construction) ->
    token flips against the reference's outputs and the max abs error of the Euler step."""
    from flowmol_amd.engine import StepNoise, make_step_plan
    device = device or eng.device
    hc, last, eta, s_idx, T = [float(v) for v in g[f'{case}.params']]
    last, s_idx, T = bool(last), int(s_idx), int(T)
    eng.bind(g['n_atoms'])
    plan = make_step_plan(T, eta, hc, cfg.cat_temperature)
    sc = plan.scalars[s_idx - 1]
    assert bool(sc.last_step) == last
    tape = [g[f'{case}.noise{i}'] for i in range(6 if last else 9)]
    nz, used = StepNoise.from_tape(tape, 0, last, device)
    assert used == len(tape)
    state = eng.make_state(g[f'{case}.x_t'], g[f'{case}.a_t'], g[f'{case}.c_t'], g[f'{case}.e_t'])
    dst = {k: g[f'{case}.dst.{k}'].to(device).contiguous() for k in 'xace'}
    i32 = dict(dtype=torch.int32, device=device)
    smp = {'a1': torch.zeros(eng.N, **i32), 'c1': torch.zeros(eng.N, **i32), 'e1': torch.zeros(eng.U, **i32)}
    eng.ctmc_step(state, dst, nz, sc, smp)
    eng.synchronize()
    res = {}
    for k in 'ace':
        res[f'{k}_flips'] = int((state[f'{k}_t'].cpu().long() != g[f'{case}.{k}_new']).sum())
        res[f'{k}1_flips'] = int((smp[f'{k}1'].cpu().long() != g[f'{case}.{k}_1_pred']).sum())
    res['x_abs'] = float((state['x_t'].cpu() - g[f'{case}.x_new']).abs().max())
    return res


def cosine_cfg(cfg):
    """The preset under the schedule of tests/golden/integrate_qm9_cosine.npz (oracle/make_golden.py:gen_integrate_cosine)."""
    import dataclasses
    return dataclasses.replace(cfg, schedule_type={'x': 'cosine', 'a': 'cosine', 'c': 'cosine', 'e': 'linear'},
                               cosine_params={'x': 1, 'a': 2, 'c': 2})


def endpoint_cfg():
    """Config of tests/golden/integrate_endpoint.npz (oracle/make_golden.py:gen_integrate_endpoint)."""
    import dataclasses
    from flowmol_amd import presets
    return dataclasses.replace(presets.endpoint_small(), continuous_inv_temp_schedule='linear', continuous_inv_temp_max=1.5)


def endpoint_golden(eng, g, device=None):
    """Engine vs the reference's EndpointVectorField on the fixture: one network evaluation of the prior state at t = 0.25
    (dense-embedding path) and the free-running Euler integration of x, a, c, e.  Returns relative errors (max abs / max abs)."""
    device = device or eng.device
    eng.bind(g['n_atoms'])
    res = {}

    def rel(got, want):
        return float((got.detach().cpu() - want).abs().max() / want.abs().max())
    st = eng.make_dense_state(g['x_0'], g['a_0'], g['c_0'], g['e_0_upper'])
    out = eng.forward_dense(st, 0.25, remove_com=True)
    eng.synchronize()
    for k in 'xace':
        res[f'fwd.{k}'] = rel(out[k], g[f'fwd.{k}'])
    st = eng.make_dense_state(g['x_0'], g['a_0'], g['c_0'], g['e_0_upper'])
    eng.integrate_endpoint(st, int(g['T']))
    for k, ref in (('x', 'x_1'), ('a', 'a_1'), ('c', 'c_1'), ('e', 'e_1_upper')):
        res[f'int.{k}'] = rel(st[f'{k}_t'], g[ref])
    return res


def oracle_rounding_sensitivity(cfg, sd, n_atoms, t_val, with_prev, seed=3, frac_masked=0.4):
    """How far the reference arithmetic itself moves when only its rounding changes: the oracle in float32 against the oracle in float64 on
    the inputs forward_compare() uses (same generator order).  {stage: max |f32 - f64| / max |f64|} for every tap and output -- the yardstick for
    kernel errors in ill-conditioned regimes (large weights), where a fixed tolerance says nothing."""
    batch = cpu_ref.build_batch(n_atoms)
    gen = torch.Generator().manual_seed(seed)
    N = int(n_atoms.sum())
    U = int((n_atoms * (n_atoms - 1) // 2).sum())
    a = rand_tokens(N, cfg.n_atom_types, frac_masked, gen)
    c = rand_tokens(N, cfg.n_charges, frac_masked, gen)
    eu = rand_tokens(U, cfg.n_bond_types, frac_masked, gen)
    x = torch.randn(N, 3, generator=gen) * 1.5
    prev = None
    if with_prev and cfg.self_conditioning:
        prev = {'x': x + 0.3 * torch.randn(N, 3, generator=gen),
                'a': torch.softmax(torch.randn(N, cfg.n_atom_types, generator=gen), -1),
                'c': torch.softmax(torch.randn(N, cfg.n_charges, generator=gen), -1),
                'e': torch.softmax(torch.randn(U, cfg.n_bond_types, generator=gen), -1)}
    a1h, c1h, e1h = onehots(cfg, batch, a, c, eu)
    res = {}
    try:
        for dt in (torch.float32, torch.float64):
            torch.set_default_dtype(dt)
            orc = cpu_ref.OracleVF(cfg, sd)
            orc.p = {k: v.to(dt) for k, v in orc.p.items()}
            orc.taps = {}
            with torch.no_grad():
                out = orc.forward(batch, x.to(dt), a1h.to(dt), c1h.to(dt), e1h.to(dt), torch.full((batch.B,), float(t_val), dtype=dt),
                                  prev=None if prev is None else {k: v.to(dt) for k, v in prev.items()}, apply_softmax=True, remove_com=True)
            res[dt] = dict(orc.taps)
            res[dt].update({f'out.{k}': v for k, v in out.items()})
    finally:
        torch.set_default_dtype(torch.float32)
    lo, hi = res[torch.float32], res[torch.float64]
    return {k: float((lo[k].double() - hi[k]).abs().max() / hi[k].abs().max().clamp(min=1e-30)) for k in hi if k in lo}


def scaled_weights(sd, scale):
    """flowmol_amd.weights.scaled_weights (shared with oracle/make_golden.py's long-horizon fixtures)."""
    from flowmol_amd.weights import scaled_weights as f
    return f(sd, scale)


# ------------------------------------------------------------------------------------------------------------------
# long-horizon reference trajectories (tests/golden/long_*.npz, oracle/make_golden.py:gen_integrate_long): the product's
# default protocol (250 / 500 steps), noise re-drawn from the stored seed in the reference's order
def long_case(g):
    """(preset name is in the file name) -> (weight scale, T, seeds) of a long fixture."""
    return float(g['weight_scale']), int(g['T']), int(g['seed_prior']), int(g['seed_noise'])


def long_report(g, traj, final):
    """Compare a free-running trajectory -- traj['a'|'c'|'e'|'a1'|'c1'|'e1'] (T-1, rows) tokens and traj['x'|'x1'] (T-1, N, 3) after every
    step, final = {'x','a','c','e'} -- with the reference's stored one.  Returns flips of the final state, the first step at which any
    token of the state differs (None = never), the number of differing state tokens summed over all steps, and coordinate errors
    (final, per-step per-molecule norms, the stored every-10th frames), all as max |diff| / max |ref|."""
    T = int(g['T'])
    sizes = g['n_atoms'].tolist()
    res = {}
    for k, key in (('a', 'a_1'), ('c', 'c_1'), ('e', 'e_1_upper')):
        res[f'{k}_flips'] = int((final[k].cpu().long() != g[key].long()).sum())
    res['x_rel'] = float((final['x'].cpu() - g['x_1']).abs().max() / g['x_1'].abs().max())
    first, total = None, 0
    for k in 'ace':
        d = traj[k].cpu().long() != g[f'traj.{k}'][1:].long()                   # (T-1, rows)
        d1 = traj[f'{k}1'].cpu().long() != g[f'traj.{k}1'].long()
        total += int(d.sum())
        res[f'{k}_state_diffs'] = int(d.sum())
        res[f'{k}1_sample_diffs'] = int(d1.sum())
        bad = torch.nonzero(d.any(dim=1) | d1.any(dim=1)).flatten()
        if bad.numel():
            first = int(bad[0]) if first is None else min(first, int(bad[0]))
    res['first_divergent_step'] = first
    res['state_token_diffs_all_steps'] = total
    # molecules whose state tokens differ from the reference's at ANY step: molecules never interact, so a decision that falls the other way on a
    # near-tie (two p~/q values, or a purity against the high-confidence threshold, equal to f32 summation order) is confined to its molecule
    n_t = torch.tensor(sizes)
    mol_of = {'a': torch.repeat_interleave(torch.arange(len(sizes)), n_t), 'e': torch.repeat_interleave(torch.arange(len(sizes)), n_t * (n_t - 1) // 2)}
    mol_of['c'] = mol_of['a']
    div = set()
    for k in 'ace':
        d = (traj[k].cpu().long() != g[f'traj.{k}'][1:].long()).any(dim=0)
        div |= set(mol_of[k][d].tolist())
    res['molecules_with_state_diffs'] = sorted(div)
    x = traj['x'].cpu()
    nrm = torch.stack([c.flatten(1).norm(dim=1) for c in torch.split(x, sizes, dim=1)], dim=1)          # (T-1, B)
    res['x_norm_rel'] = float(((nrm - g['traj.x_norm'][1:]).abs() / g['traj.x_norm'][1:]).max())
    x1 = traj['x1'].cpu()
    nrm1 = torch.stack([c.flatten(1).norm(dim=1) for c in torch.split(x1, sizes, dim=1)], dim=1)
    res['x1_norm_rel'] = float(((nrm1 - g['traj.x1_norm']).abs() / g['traj.x1_norm']).max())
    st = int(g['traj.x_stride'])
    ref_fr = g['traj.x'][1:]                                                     # frames st, 2 st, ... of the state trajectory (frame 0 = prior)
    got_fr = x[st - 1::st][:ref_fr.shape[0]]
    res['x_frames_rel'] = float((got_fr - ref_fr).abs().max() / ref_fr.abs().max()) if got_fr.numel() else 0.0     # fewer steps than the frame stride
    # how far the endpoint prediction is from the state: the coordinates' sensitivity to the network's arithmetic
    res['mean_rel_move'] = float(((x1 - x).flatten(1).norm(dim=1) / x.flatten(1).norm(dim=1))[: T - 2].mean())
    return res


def integrate_long_golden(eng, cfg, g, chunk=16, device=None, max_steps=None):
    """Engine run of a long fixture.  ``max_steps``: only the first steps (CPU emulation / quick checks) -- then only the per-step
    comparisons are meaningful and the final-state entries are dropped."""
    from flowmol_amd.engine import IntegrationRun, StepNoise, make_step_plan
    device = device or eng.device
    eng.bind(g['n_atoms'])
    T = int(g['T'])
    plan = make_step_plan(T, cfg.stochasticity, cfg.high_confidence_threshold, cfg.cat_temperature,
                          schedule_type=cfg.schedule_type, cosine_params=cfg.cosine_params)
    state = eng.prior_state(g['x_0'])
    N, U = eng.N, eng.U
    torch.manual_seed(int(g['seed_noise']))

    def noise_for_step(i, last):            # the reference's draws, re-drawn from the seed on torch's CPU generator, in its order
        nz = StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, last, 'cpu')
        return StepNoise(**{k: (None if getattr(nz, k) is None else getattr(nz, k).to(device)) for k in nz.__slots__})
    n_steps = T - 1 if max_steps is None else min(max_steps, T - 1)
    i32 = dict(dtype=torch.int32, device=device)
    traj = {'x': torch.zeros(n_steps, N, 3, device=device), 'x1': torch.zeros(n_steps, N, 3, device=device)}
    for k, rows in (('a', N), ('c', N), ('e', U)):
        traj[k] = torch.zeros(n_steps, rows, **i32)
        traj[f'{k}1'] = torch.zeros(n_steps, rows, **i32)
    run = IntegrationRun(eng, state, plan, noise_for_step, traj=traj)
    run.run(0, n_steps, chunk=chunk)
    eng.synchronize()
    final = {k: state[f'{k}_t'] for k in 'xace'}
    if n_steps < T - 1:
        gg = dict(g)
        for k in ('traj.a', 'traj.c', 'traj.e', 'traj.x_norm'):
            gg[k] = g[k][: n_steps + 1]
        for k in ('traj.a1', 'traj.c1', 'traj.e1', 'traj.x1_norm'):
            gg[k] = g[k][:n_steps]
        st = int(g['traj.x_stride'])
        gg['traj.x'] = g['traj.x'][: n_steps // st + 1]
        res = long_report(gg, traj, final)
        for k in ('a_flips', 'c_flips', 'e_flips', 'x_rel'):
            res.pop(k)
        res['steps'] = n_steps
        return res
    return long_report(g, traj, final)


def oracle_long_golden(orc, cfg, g, max_steps=None):
    """The CPU oracle on a long fixture (same seeds, torch's global generator like the reference)."""
    batch = cpu_ref.build_batch(g['n_atoms'])
    T = int(g['T'])
    prior = {'x_0': g['x_0'], 'a_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_atom_types), 'c_0': cpu_ref.ctmc_masked_prior(batch.N, cfg.n_charges),
             'e_0': cpu_ref.edge_prior(batch.upper_edge_mask, cfg.n_bond_types)}
    m = batch.upper_edge_mask
    rec = {k: [] for k in ('x', 'a', 'c', 'e', 'x1', 'a1', 'c1', 'e1')}

    class Stop(Exception):
        pass

    def hook(s_idx, new, dst):
        rec['x'].append(new['x_t'].clone()); rec['x1'].append(new['x_1_pred'].clone())
        for k in 'ac':
            rec[k].append(new[f'{k}_t'].argmax(-1)); rec[f'{k}1'].append(new[f'{k}_1_pred'].argmax(-1))
        rec['e'].append(new['e_t'][m].argmax(-1)); rec['e1'].append(new['e_1_pred'][m].argmax(-1))
        if max_steps is not None and s_idx >= max_steps:
            raise Stop
    torch.manual_seed(int(g['seed_noise']))
    out = None
    try:
        with torch.no_grad():
            out = orc.integrate(batch, prior, T, step_hook=hook)
    except Stop:
        pass
    traj = {k: torch.stack(v) for k, v in rec.items()}
    n_steps = traj['x'].shape[0]
    if out is None:
        gg = dict(g)
        for k in ('traj.a', 'traj.c', 'traj.e', 'traj.x_norm'):
            gg[k] = g[k][: n_steps + 1]
        for k in ('traj.a1', 'traj.c1', 'traj.e1', 'traj.x1_norm'):
            gg[k] = g[k][:n_steps]
        gg['traj.x'] = g['traj.x'][: n_steps // int(g['traj.x_stride']) + 1]
        final = {'x': traj['x'][-1], 'a': traj['a'][-1], 'c': traj['c'][-1], 'e': traj['e'][-1]}
        res = long_report(gg, traj, final)
        for k in ('a_flips', 'c_flips', 'e_flips', 'x_rel'):
            res.pop(k)
        res['steps'] = n_steps
        return res
    final = {'x': out['x_1'], 'a': out['a_1'].argmax(-1), 'c': out['c_1'].argmax(-1), 'e': out['e_1'][m].argmax(-1)}
    return long_report(g, traj, final)


# ------------------------------------------------------------------------------------------------------------------
def hparams_from_yaml(yaml_name='flowmol3.yml'):
    """``hyper_parameters`` of a checkpoint trained from one of the reference's shipped YAMLs, as derived mechanically by
    oracle/make_hparams_fixture.py (model_from_config's mapping, load.py:13-49, plus FlowMol.__init__'s defaults) -- NOT from a preset."""
    import json
    import pathlib
    fx = json.loads((pathlib.Path(__file__).resolve().parent / 'golden' / 'hparams_from_yaml.json').read_text())
    return json.loads(json.dumps(fx[yaml_name]['hyper_parameters']))        # deep copy


def write_lightning_shaped_checkpoint(root, model_name, sd, yaml_name='flowmol3.yml'):
    """<root>/<model_name>/checkpoints/last.ckpt + config.yaml shaped like the files the reference's load_pretrained reads
    (flowmol/__init__.py:30-56, trained_models/readme.md): a Lightning checkpoint whose ``hyper_parameters`` is an instance of
    pytorch_lightning's AttributeDict (a class this image does not have), holding the kwargs of the reference's FlowMol.__init__
    (flowmol.py:29-55) as model_from_config passes them for the shipped YAML ``yaml_name`` (tests/golden/hparams_from_yaml.json, derived
    from /root/reference/configs by oracle/make_hparams_fixture.py -- not from the preset the loaded model is then compared with) -- the
    two data files as pathlib.PosixPath objects -- next to Lightning's bookkeeping keys, and a state dict with the ``vector_field.``
    prefix.  The AttributeDict class exists only while the file is written, so reading it exercises the Lightning-free unpickler."""
    import pathlib
    import sys
    import types
    mod = types.ModuleType('pytorch_lightning.utilities.parsing')

    class AttributeDict(dict):
        pass
    AttributeDict.__module__, AttributeDict.__qualname__ = mod.__name__, 'AttributeDict'
    mod.AttributeDict = AttributeDict
    names = ['pytorch_lightning', 'pytorch_lightning.utilities', 'pytorch_lightning.utilities.parsing']
    saved = {n: sys.modules.get(n) for n in names}
    sys.modules['pytorch_lightning'] = types.ModuleType('pytorch_lightning')
    sys.modules['pytorch_lightning.utilities'] = types.ModuleType('pytorch_lightning.utilities')
    sys.modules[mod.__name__] = mod
    hp = AttributeDict(hparams_from_yaml(yaml_name))
    for k in ('n_atoms_hist_file', 'marginal_dists_file'):           # model_from_config passes pathlib paths (load.py:23-25)
        hp[k] = pathlib.PosixPath(hp[k])
    ck = {'epoch': 19, 'global_step': 1234567, 'pytorch-lightning_version': '2.1.3',
          'state_dict': {'vector_field.' + k: v for k, v in sd.items()}, 'loops': {'fit_loop': {'epoch_progress': {'total': {'ready': 20}}}},
          'callbacks': {"ModelCheckpoint{'monitor': 'val_total_loss'}": {'best_model_score': torch.tensor(1.5), 'dirpath': '/net/runs/flowmol3/checkpoints'}},
          'optimizer_states': [], 'lr_schedulers': [], 'hparams_name': 'kwargs', 'hyper_parameters': hp}
    d = pathlib.Path(root) / model_name / 'checkpoints'
    d.mkdir(parents=True, exist_ok=True)
    try:
        torch.save(ck, d / 'last.ckpt')
    finally:
        for n in names:
            if saved[n] is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = saved[n]
    (d.parent / 'config.yaml').write_text('# the resolved training config ships next to the checkpoint (trained_models/readme.md); load_pretrained does not read it\n')
    assert 'pytorch_lightning' not in sys.modules or saved['pytorch_lightning'] is not None
    return d / 'last.ckpt'


def traj_frames_reference_compare(model, g, device):
    """FlowMol.sample(xt_traj, ep_traj) driven with the reference's prior and recorded noise -> every molecule's traj_frames_reference() against
    the frame dicts the reference's own CTMCVectorField.integrate(visualize=True) returned (tests/golden/traj_frames.npz): same keys, shapes and
    dtypes; categorical frames (float one-hots incl. the mask column, all directed edges) bit for bit; coordinates within 1e-4 relative."""
    import torch.nn.functional as F
    from flowmol_amd.engine import StepNoise
    cfg = model.cfg
    n_atoms = g['n_atoms']
    N = int(n_atoms.sum())
    E = int((n_atoms * (n_atoms - 1)).sum())
    tape = [g[k] for k in sorted(k for k in g if k.startswith('noise.'))]
    pos = [0]

    def noise_for_step(i, last):
        nz, pos[0] = StepNoise.from_tape(tape, pos[0], last, device)
        return nz
    prior = {'x_0': g['x_0'], 'a_0': F.one_hot(torch.full((N,), cfg.n_atom_types), cfg.n_atom_types + 1).float(),
             'c_0': F.one_hot(torch.full((N,), cfg.n_charges), cfg.n_charges + 1).float(),
             'e_0': F.one_hot(torch.full((E,), cfg.n_bond_types), cfg.n_bond_types + 1).float(), 'fake_atoms': cfg.fake_atoms}
    mols = model.sample(n_atoms, n_timesteps=int(g['T']), xt_traj=True, ep_traj=True, prior=prior, _noise_for_step=noise_for_step, device=device)
    assert pos[0] == len(tape) and len(mols) == len(n_atoms)
    worst, cells = 0.0, 0
    for m, mol in enumerate(mols):
        got = mol.traj_frames_reference()
        ref = {k.split('.', 1)[1]: g[k] for k in g if k.startswith(f'mol{m}.')}
        assert sorted(got) == sorted(ref), (sorted(got), sorted(ref))
        for k, r in ref.items():
            assert got[k].shape == r.shape and got[k].dtype == r.dtype, (m, k, got[k].shape, r.shape, got[k].dtype)
            if k.startswith('x'):
                worst = max(worst, float((got[k] - r).abs().max() / r.abs().max()))
            else:
                assert torch.equal(got[k], r), (m, k, int((got[k] != r).sum()))
                cells += r.numel()
    return {'molecules': len(mols), 'x_rel': worst, 'categorical_cells_bit_equal': cells}


# ------------------------------------------------------------------------------------------------------------------
# teacher-forced run of a long fixture with an audit of every decision that differs from the reference's
SAMPLE_TIE_TOL = 1e-4       # two candidates' (p~ / sum) / q within this relative distance: their order is decided by f32 summation order (measured flips: <= 2e-5)
PURITY_TIE_TOL = 5e-5       # |max p~ - hc_thresh| / hc_thresh


def integrate_long_teacher_forced(eng, cfg, g, device=None, max_steps=None):
    """Every step of a long fixture started from the REFERENCE's token state of that step (our own coordinates and self-conditioning input run
    free), so that every one of the fixture's categorical decisions -- sampled endpoint token and new state token of every row at every step -- is
    compared under the reference's own preconditions instead of only up to the first divergence.  Each step's endpoint probabilities are kept;
    audit_long_decisions() then explains every differing decision.  Returns (traj, probs): traj[k] / traj[k+'1'] (steps, rows) int32 new state /
    sampled tokens, traj['x'|'x1']; probs[k] (steps, rows, K)."""
    from flowmol_amd.engine import IntegrationRun, StepNoise, make_step_plan
    device = device or eng.device
    eng.bind(g['n_atoms'])
    T = int(g['T'])
    plan = make_step_plan(T, cfg.stochasticity, cfg.high_confidence_threshold, cfg.cat_temperature,
                          schedule_type=cfg.schedule_type, cosine_params=cfg.cosine_params)
    state = eng.prior_state(g['x_0'])
    N, U = eng.N, eng.U
    torch.manual_seed(int(g['seed_noise']))

    def noise_for_step(i, last):
        nz = StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, last, 'cpu')
        return StepNoise(**{k: (None if getattr(nz, k) is None else getattr(nz, k).to(device)) for k in nz.__slots__})
    n_steps = T - 1 if max_steps is None else min(max_steps, T - 1)
    i32 = dict(dtype=torch.int32, device=device)
    ref = {k: g[f'traj.{k}'][:n_steps].to(device, torch.int32) for k in 'ace'}            # frame i = the state step i starts from
    traj = {'x': torch.zeros(n_steps, N, 3, device=device), 'x1': torch.zeros(n_steps, N, 3, device=device)}
    probs = {}
    for k, rows, K in (('a', N, cfg.n_atom_types), ('c', N, cfg.n_charges), ('e', U, cfg.n_bond_types)):
        traj[k] = torch.zeros(n_steps, rows, **i32)
        traj[f'{k}1'] = torch.zeros(n_steps, rows, **i32)
        probs[k] = torch.zeros(n_steps, rows, K, device=device)
    run = IntegrationRun(eng, state, plan, noise_for_step, traj=traj)
    for i in range(n_steps):
        for k in 'ace':
            state[f'{k}_t'].copy_(ref[k][i])
        run.run(i, i + 1, chunk=1)
        d = run.last_dst()
        for k in 'ace':
            probs[k][i].copy_(d[k])
    eng.synchronize()
    return traj, probs


def audit_long_decisions(cfg, g, traj, probs):
    """Compare the teacher-forced decisions with the reference's and explain every difference.  A sampled endpoint token may differ only where the
    two candidates' (p~ / sum p~) / q -- OUR tempered probabilities, the shared noise -- lie within SAMPLE_TIE_TOL of each other; a new state token may
    differ only at a row whose sampled token differs that way, or in a molecule with a masked row whose purity max p~ lies within PURITY_TIE_TOL of
    the high-confidence threshold (ctmc_utils.py:10-18: the count h of high-confidence rows then differs by one and with it the molecule's
    unmasking probabilities).  Returns {'decisions', 'sample_diffs', 'state_diffs', 'events': [...], 'unexplained': [...]}."""
    from flowmol_amd.engine import StepNoise
    sizes = g['n_atoms']
    n_steps = int(traj['a'].shape[0])
    N, U = int(sizes.sum()), int((sizes * (sizes - 1) // 2).sum())
    B = int(sizes.numel())
    mol_n = torch.repeat_interleave(torch.arange(B), sizes)
    mol_p = torch.repeat_interleave(torch.arange(B), sizes * (sizes - 1) // 2)
    mods = (('a', cfg.n_atom_types, mol_n), ('c', cfg.n_charges, mol_n), ('e', cfg.n_bond_types, mol_p))
    ours = {k: v.cpu().long() for k, v in traj.items() if k not in ('x', 'x1')}
    flagged = set()
    for k, K, _ in mods:
        d = (ours[k] != g[f'traj.{k}'][1:n_steps + 1].long()) | (ours[f'{k}1'] != g[f'traj.{k}1'][:n_steps].long())
        flagged |= set(torch.nonzero(d.any(dim=1)).flatten().tolist())
    out = {'steps': n_steps, 'decisions': n_steps * (2 * N + U), 'sample_diffs': 0, 'state_diffs': 0, 'events': [], 'unexplained': []}
    if not flagged:
        return out
    temp, hc = float(cfg.cat_temperature), float(cfg.high_confidence_threshold)
    torch.manual_seed(int(g['seed_noise']))
    for i in range(max(flagged) + 1):              # the reference's draws again, in its order; only flagged steps are looked at
        nz = StepNoise.draw(N, U, cfg.n_atom_types, cfg.n_charges, cfg.n_bond_types, i == int(g['T']) - 2, 'cpu')
        if i not in flagged:
            continue
        for k, K, mol in mods:
            p = probs[k][i].cpu()
            pt = torch.softmax(torch.log(p) / temp, dim=-1)
            v = (pt / pt.sum(-1, keepdim=True)) / getattr(nz, f'q_{k}')
            s_our, s_ref = ours[f'{k}1'][i], g[f'traj.{k}1'][i].long()
            tie_rows = {}
            for r in torch.nonzero(s_our != s_ref).flatten().tolist():
                a_, b_ = float(v[r, s_our[r]]), float(v[r, s_ref[r]])
                margin = abs(a_ - b_) / max(a_, b_)
                tie_rows[r] = margin
                out['sample_diffs'] += 1
                ev = {'step': i, 'modality': k, 'kind': 'sampled token', 'row': r, 'molecule': int(mol[r]), 'ours': int(s_our[r]), 'reference': int(s_ref[r]), 'margin': margin}
                out['events'].append(ev)
                if not margin < SAMPLE_TIE_TOL:
                    out['unexplained'].append(ev)
            masked = g[f'traj.{k}'][i].long() == K                      # the state the step started from (= the reference's)
            purity = pt.max(-1).values
            near = masked & ((purity - hc).abs() <= PURITY_TIE_TOL * hc)
            tie_mols = set(mol[near].tolist()) if hc > 0 else set()
            t_our, t_ref = ours[k][i], g[f'traj.{k}'][i + 1].long()
            bad_rows = torch.nonzero(t_our != t_ref).flatten().tolist()
            out['state_diffs'] += len(bad_rows)
            seen = set()
            for r in bad_rows:
                m = int(mol[r])
                if r in tie_rows and tie_rows[r] < SAMPLE_TIE_TOL:
                    continue                                       # the row took its (differently) sampled token: explained by the tie above
                if m in tie_mols:
                    if (k, m) not in seen:
                        seen.add((k, m))
                        rr = [x_ for x_ in torch.nonzero(near & (mol == m)).flatten().tolist()]
                        out['events'].append({'step': i, 'modality': k, 'kind': 'purity at the high-confidence threshold', 'molecule': m, 'rows': rr,
                                              'purity_minus_threshold': [float(purity[x_] - hc) for x_ in rr], 'state_rows_differing': sum(1 for q_ in bad_rows if int(mol[q_]) == m)})
                    continue
                out['unexplained'].append({'step': i, 'modality': k, 'kind': 'state token', 'row': r, 'molecule': m, 'ours': int(t_our[r]), 'reference': int(t_ref[r])})
    return out


def oracle_long_fixture(cfg, sd, n_atoms, T, seed_prior, seed_noise):
    """A fixture with the keys of tests/golden/long_*.npz, produced by the CPU ORACLE instead of the reference (tiny cases for the CPU tests of the
    teacher-forced audit; the committed fixtures come from the reference itself, oracle/make_golden.py:gen_integrate_long)."""
    batch = cpu_ref.build_batch(n_atoms)
    orc = cpu_ref.OracleVF(cfg, sd)
    torch.manual_seed(seed_prior)
    prior = orc.sample_prior(batch)
    m = batch.upper_edge_mask
    rec = {k: [] for k in ('a', 'c', 'e', 'a1', 'c1', 'e1')}
    rec['a'].append(prior['a_0'].argmax(-1)); rec['c'].append(prior['c_0'].argmax(-1)); rec['e'].append(prior['e_0'][m].argmax(-1))

    def hook(s_idx, new, dst):
        for k in 'ac':
            rec[k].append(new[f'{k}_t'].argmax(-1)); rec[f'{k}1'].append(new[f'{k}_1_pred'].argmax(-1))
        rec['e'].append(new['e_t'][m].argmax(-1)); rec['e1'].append(new['e_1_pred'][m].argmax(-1))
    torch.manual_seed(seed_noise)
    with torch.no_grad():
        out = orc.integrate(batch, prior, T, step_hook=hook)
    g = {'n_atoms': n_atoms, 'T': torch.tensor(T), 'weight_scale': torch.tensor(1.0), 'seed_prior': torch.tensor(seed_prior), 'seed_noise': torch.tensor(seed_noise),
         'x_0': prior['x_0'], 'x_1': out['x_1']}
    for k, v in rec.items():
        g[f'traj.{k}'] = torch.stack(v).to(torch.uint8)
    return g


def oracle_f64(cfg, sd):
    """The CPU oracle evaluated in float64 (parameters converted): the exact-arithmetic yardstick for kernel errors (forward_compare(..., dtype=torch.float64))."""
    try:
        torch.set_default_dtype(torch.float64)
        orc = cpu_ref.OracleVF(cfg, sd)
        orc.p = {k: v.to(torch.float64) for k, v in orc.p.items()}
    finally:
        torch.set_default_dtype(torch.float32)
    return orc
