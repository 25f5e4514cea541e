#!/usr/bin/env python3
"""Differential fuzzing of the CPU oracle against the REFERENCE itself, on the wild inputs of tools/fuzz_parity.py.

Runs only in the development container (needs /root/reference; same loader as make_golden.py, nothing is copied).
The device path is fuzzed against the oracle on the GPU box (tests/test_fuzz_parity.py); this closes the chain in the
regimes that fuzzing found -- exp() flushed to zero in the tail of a narrow beam, denormal or infinite amplitudes -- by
holding the oracle to the reference there: NaN / inf patterns and the invalid samples (profile == 1e-20) identical,
values within the same conditioning-aware tolerances.  It then writes tests/golden/plume_fuzz.npz: a sample of those
inputs with the reference's outputs, so that the pin travels with the repository.

Usage:  python tests/golden/fuzz_reference.py [--seeds 60] [--n 20000]
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tools'))
sys.path.insert(0, str(ROOT / 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=60)
    ap.add_argument('--seed-list', type=int, nargs='*', default=[], help='further seeds; with --no-fixture a quick look at given seeds')
    ap.add_argument('--n', type=int, default=20_000)
    ap.add_argument('--no-fixture', action='store_true', help='do not rewrite tests/golden/plume_fuzz.npz')
    args = ap.parse_args()
    import make_golden
    import parity_rules as pr
    from fuzz_parity import wild
    from oracle import oracle_ctypes as oc
    cathode, plume, _thruster, _const = make_golden._load_reference()
    k = make_golden.TORR_2_PA
    keep_in, keep_out = [], []
    worst = {'V_cc': 0.0, 'j_ion': 0.0, 'div_angle': 0.0, 'T_c': 0.0}
    seen = {'j_ion': {'cond': 0.0, 'tau_seen': 0.0, 'n_cancelling': 0}, 'div_angle': {'cond': 0.0, 'tau_seen': 0.0, 'n_cancelling': 0}}
    n_invalid = n_nan = 0
    seeds = list(range(args.seeds)) + [s_ for s_ in args.seed_list if s_ >= args.seeds]
    for seed in seeds:
        x = wild(np.random.default_rng(1000 + seed), args.n)
        with np.errstate(all='ignore'):
            v_ref = np.asarray(cathode.cathode_coupling({q: x[q] for q in ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')})['V_cc'])
            v_or = oc.cathode(x['P_b'], x['V_a'], x['T_e'], x['V_vac'], x['Pstar'], x['P_T'], k)
            th = oc.thruster(x['V_a'], v_or, x['mdot_a'], x['a_1'])            # the test double's formulas (sim_hallthruster.jl)
            p = {q: x[q] for q in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')}
            p['I_B0'], p['T'] = th['I_B0'], th['T']
            ref = plume.current_density(dict(p), sweep_radius=1.0)
            orc = oc.plume(p['P_b'], p['c0'], p['c1'], p['c2'], p['c3'], p['c4'], p['c5'], p['sigma_cex'], p['I_B0'], k, T=p['T'])
            terms = oc.plume_terms(p['P_b'], p['c0'], p['c1'], p['c2'], p['c3'], p['c4'], p['c5'], p['sigma_cex'], p['I_B0'], k)
            bounds = pr.plume_bounds(terms, p['I_B0'])
        assert np.array_equal(np.isnan(v_ref), np.isnan(v_or)), f'V_cc NaN pattern (seed {seed})'
        with np.errstate(all='ignore'):          # same operations in the same order; numpy's log and libm's differ by an ulp
            lg = np.log(1.0 + x['P_b'] * k / (x['P_T'] * k))
            v_scale = np.abs(x['V_vac']) + np.abs(x['T_e'] * lg) + np.abs(x['T_e'] / ((x['P_T'] + x['Pstar']) * k) * (x['P_b'] * k))
        fin = np.isfinite(v_ref) & np.isfinite(v_scale)
        assert np.array_equal(np.isinf(v_ref), np.isinf(v_or))
        worst['V_cc'] = max(worst['V_cc'], float(np.max(np.abs(v_ref[fin] - v_or[fin]) / (np.abs(v_ref[fin]) + (pr.TAU / pr.TOL) * v_scale[fin] + 1e-300), initial=0.0)))
        assert worst['V_cc'] == worst['V_cc']
        rj, oj = np.asarray(ref['j_ion'], dtype=np.float64), orc['j_ion'].reshape(args.n, 91)
        inv_ref = np.all(rj == 1e-20, axis=1)
        assert np.array_equal(inv_ref, orc['invalid']), f'invalid samples differ (seed {seed})'
        n_invalid += int(inv_ref.sum())
        n_nan += int(np.isnan(rj).any(axis=1).sum())
        # the oracle against the reference under the very rules the device path is held to against the oracle
        r = pr.j_ion_error(oj, rj, bounds, f'j_ion seed {seed}')
        d = pr.divergence_error(orc['div_angle'], ref['div_angle'], orc['T_c'], ref['T_c'], bounds, f'seed {seed}')
        worst['j_ion'] = max(worst['j_ion'], r['err'])
        worst['div_angle'] = max(worst['div_angle'], d['err_div'])
        worst['T_c'] = max(worst['T_c'], d['err_tc'])
        for key, rec in (('j_ion', r), ('div_angle', d)):
            seen[key]['cond'] = max(seen[key]['cond'], rec['cond'])
            seen[key]['tau_seen'] = max(seen[key]['tau_seen'], rec['tau_seen'])
            seen[key]['n_cancelling'] += rec['n_cancelling']
        # fixture: a slice of every seed plus every sample that sits in one of the special regimes
        with np.errstate(all='ignore'):
            base = p['I_B0'] * np.exp(-(x['c4'] * (x['P_b'] * k) + x['c5']) * x['sigma_cex'])
            special = inv_ref | np.isnan(rj).any(axis=1) | np.isinf(rj).any(axis=1) | ((np.abs(base) < 1e-280) & (base != 0))
        pick = np.zeros(args.n, dtype=bool)
        pick[:12] = True
        pick[np.flatnonzero(special)[:14]] = True
        keep_in.append({q: v[pick] for q, v in p.items()})
        keep_out.append({'j_ion': rj[pick], 'div_angle': np.asarray(ref['div_angle'], dtype=np.float64)[pick], 'T_c': np.asarray(ref['T_c'], dtype=np.float64)[pick]})
    print(f'{len(seeds)} seeds x {args.n} wild samples, oracle vs reference: {n_invalid} invalid and {n_nan} NaN '
          f'samples, patterns identical; worst errors (1e-10 = at the bound of tests/parity_rules.py) {worst}; cancellation met: {seen}')
    assert worst['j_ion'] <= 1e-10 and worst['div_angle'] <= 1e-10 and worst['T_c'] <= 1e-10 and worst['V_cc'] <= 1e-10
    if args.no_fixture or args.seed_list:
        return
    arrays = {f'in_{q}': np.concatenate([d[q] for d in keep_in]) for q in keep_in[0]}
    arrays.update({f'out_{q}': np.concatenate([d[q] for d in keep_out]) for q in keep_out[0]})
    make_golden._save('plume_fuzz', **arrays, radii=np.array([1.0]))


def fuzz_filter_outputs(cases: int = 300):
    """drivers.filter_outputs (numpy and torch paths) against the reference's own `_filter_outputs` (gen_data.py:125-174,
    pulled out of the script's AST as in make_golden.py) on random outputs with NaNs, heavy tails and zero-IQR fields."""
    import ast
    import torch
    from hallthrusterpem_amd import drivers
    src = (Path('/root/reference/scripts/gen_data.py')).read_text()
    wanted = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in ('_object_to_numeric', '_filter_outputs')]
    ns = {'np': np, 'COORDS_STR_ID': '_coords'}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), 'gen_data.py[extract]', 'exec'), ns)
    for seed in range(cases):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(5, 400))
        fo = {'a': rng.normal(0, 1, n), 'b': rng.standard_cauchy((n, int(rng.integers(1, 40)))), 'c': rng.normal(0, 1, (n, 3, 4)),
              'a_coords': np.zeros((n, 3)), 'errors': np.zeros(n)}
        for key in ('a', 'b', 'c'):
            m = rng.random(fo[key].shape) < 0.02
            fo[key][m] = np.nan if seed % 3 == 0 else fo[key][m] * 100
        if seed % 5 == 0:
            fo['b'][:] = 1.0
        q = float(rng.choice([1.5, 3.0, 0.5]))
        with np.errstate(all='ignore'):
            rn, ro = ns['_filter_outputs'](dict(fo), iqr_factor=q)
            gn, go = drivers.filter_outputs(dict(fo), iqr_factor=q)
            tn, to = drivers.filter_outputs({key: torch.from_numpy(np.asarray(v)) for key, v in fo.items()}, iqr_factor=q)
        assert set(rn) == set(gn) == set(tn) == {'a', 'b', 'c'}
        for key in rn:
            assert np.array_equal(rn[key], gn[key]) and np.array_equal(ro[key], go[key]), (seed, key)
            assert np.array_equal(rn[key], tn[key].numpy()) and np.array_equal(ro[key], to[key].numpy()), (seed, key)
    print(f'filter_outputs: {cases} random cases, NaN and outlier masks identical to the reference function (numpy and torch paths)')


def fuzz_thruster_host(cases: int = 2000):
    """The host-side thruster helpers against the reference's (thruster.py:93-181): `_default_model_fidelity` on random
    fidelity tuples / configs / CFL numbers, `_convert_to_julia` / `_convert_to_pem` on random path maps -- results and
    exception types identical."""
    import copy
    import warnings
    import make_golden
    from hallthrusterpem_amd.models import thruster as mine
    _c, _p, ref, _k = make_golden._load_reference()
    rng = np.random.default_rng(0)
    for i in range(cases):
        mf = () if i % 17 == 0 else (int(rng.integers(0, 5)), int(rng.integers(0, 4)))
        cfg = {}
        if i % 3:
            cfg['config'] = {}
        if i % 3 == 1:
            cfg['config'].update({'domain': [0.0, float(rng.uniform(0.02, 0.2))], 'discharge_voltage': float(rng.uniform(100, 800)),
                                  'cathode_coupling_voltage': float(rng.uniform(0, 60))})
        if i % 5 == 0 and 'config' in cfg:
            cfg['config']['propellant'] = str(rng.choice(['Xenon', 'Krypton', 'Argon']))
        cfl = float(rng.uniform(0.05, 0.5))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            assert ref._default_model_fidelity(mf, copy.deepcopy(cfg), cfl) == mine._default_model_fidelity(mf, copy.deepcopy(cfg), cfl), (mf, cfg)

    def outcome(fn, *a):
        try:
            return fn(*a), None
        except Exception as e:                                             # noqa: BLE001 -- the type is what is compared
            return None, type(e).__name__

    for i in range(cases):
        keys = [f'k{j}' for j in range(int(rng.integers(1, 6)))]
        p2j = {q: [str(rng.choice(['a', 'b', 'c', 'output', 'average', 'config', 'x'])) for _ in range(int(rng.integers(1, 4)))] for q in keys}
        for q in keys:
            if rng.random() < 0.3:
                p2j[q][0] = 'output'
        pem = {q: float(rng.normal()) for q in keys if rng.random() < 0.8}
        if rng.random() < 0.2:
            pem['unmapped'] = 1.0
        ja, jb = {}, {}
        _, ea = outcome(ref._convert_to_julia, copy.deepcopy(pem), ja, copy.deepcopy(p2j))
        _, eb = outcome(mine._convert_to_julia, copy.deepcopy(pem), jb, copy.deepcopy(p2j))
        assert ea == eb and (ea is not None or ja == jb), (pem, p2j, ea, eb)
        if ea is None:
            assert outcome(ref._convert_to_pem, copy.deepcopy(ja), copy.deepcopy(p2j)) == outcome(mine._convert_to_pem, copy.deepcopy(jb), copy.deepcopy(p2j))
    # _format_hallthruster_jl_input (thruster.py:184-330) on random input subsets, anomalous-transport models, optional
    # blocks and fidelities; the four random characters of the output file name are the only thing allowed to differ
    import json
    import re
    assert ref.PEM_TO_JULIA == mine.PEM_TO_JULIA
    in_keys = [q for q, v in ref.PEM_TO_JULIA.items() if v and v[0] != 'output']
    norm = lambda o: re.sub(r'_[A-Z0-9]{4}\.json', '_XXXX.json', json.dumps(o, sort_keys=True, default=str))   # noqa: E731

    def quiet(fn, *a, **kw):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            return outcome(lambda: fn(*a, **kw))

    for i in range(cases):
        ti = {q: float(rng.uniform(0.001, 100)) for q in in_keys if rng.random() < 0.5}
        typ = str(rng.choice(['TwoZoneBohm', 'GaussianBohm', 'NoAnom']))
        cfg = {'anom_model': {'type': typ}}
        if typ == 'TwoZoneBohm' and rng.random() < 0.7:
            cfg['anom_model'].update({'c1': 0.00625, 'c2': 0.0625})
        if typ == 'GaussianBohm' and rng.random() < 0.7:
            cfg['anom_model'].update({'hall_min': 0.00625, 'hall_max': 0.0625})
        if rng.random() < 0.5:
            cfg['domain'] = [0, 0.08]
        if rng.random() < 0.3:
            cfg = None
        kw = dict(thruster=None, config=cfg, simulation=None if rng.random() < 0.5 else {'dt': 1e-9, 'duration': 1e-3},
                  postprocess=None if rng.random() < 0.5 else {'average_start_time': 5e-4},
                  model_fidelity=None if rng.random() < 0.3 else (() if rng.random() < 0.2 else (int(rng.integers(0, 3)), int(rng.integers(0, 3)))),
                  output_path=None if rng.random() < 0.5 else 'out.json')
        a = quiet(ref._format_hallthruster_jl_input, copy.deepcopy(ti), ref.PEM_TO_JULIA, **copy.deepcopy(kw))
        b = quiet(mine._format_hallthruster_jl_input, copy.deepcopy(ti), mine.PEM_TO_JULIA, **copy.deepcopy(kw))
        assert norm(a) == norm(b), (ti, kw)
    print(f'thruster host helpers: 3 x {cases} random cases identical to the reference functions')


if __name__ == '__main__':
    main()
    fuzz_filter_outputs()
    fuzz_thruster_host()
