#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the development container (needs /root/reference); the GPU box never sees the
reference, only the .npz files this script writes.  Nothing from the reference is copied:
its two model files are imported from where they lie, called on seeded inputs, and the
inputs + outputs are stored.

The reference imports `pem_core` (not installed, not in-tree: uv.lock:1655-1657).  The three
names the model files read from it are supplied by an in-memory module: the constant
TORR_2_PA (value recorded in every fixture), the `Dataset`/`ArrayLike` typing aliases, and
`get_logger`.  See SURVEY.md §8c.

Usage:  python tests/golden/make_golden.py           (rewrites tests/golden/*.npz)
"""
import importlib.util
import json
import logging
import sys
import types
from pathlib import Path

sys.dont_write_bytecode = True   # never leave __pycache__ inside the read-only reference tree

import numpy as np

REF = Path('/root/reference/src/hallmd')
OUT = Path(__file__).resolve().parent
TORR_2_PA = 133.322          # assumed value of pem_core.constants.TORR_2_PA


def _load_reference():
    pc = types.ModuleType('pem_core')
    pc.get_logger = lambda name: logging.getLogger(name)
    const = types.ModuleType('pem_core.constants')
    const.TORR_2_PA = TORR_2_PA
    const.AVOGADRO_CONSTANT = 6.02214076e23
    const.FUNDAMENTAL_CHARGE = 1.602176634e-19
    const.MOLECULAR_WEIGHTS = {'Xenon': 131.293, 'Krypton': 83.798}
    typ = types.ModuleType('pem_core.types')
    typ.Dataset = dict
    typ.ArrayLike = object
    sys.modules.update({'pem_core': pc, 'pem_core.constants': const, 'pem_core.types': typ})

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    cathode = load('_ref_cathode', REF / 'models' / 'cathode.py')
    plume = load('_ref_plume', REF / 'models' / 'plume.py')
    sys.path.insert(0, str(REF.parent))
    import hallmd.models.thruster as thruster  # noqa: E402  (needs only yaml/json + the stand-in)
    return cathode, plume, thruster, const


def _save(name, **arrays):
    arrays['TORR_2_PA'] = np.float64(TORR_2_PA)
    np.savez(OUT / f'{name}.npz', **arrays)
    print(f'wrote {name}.npz  ({(OUT / (name + ".npz")).stat().st_size / 1024:.0f} KiB)')


def _plume_out(plume, inputs, radius):
    with np.errstate(all='ignore'):
        out = plume.current_density(dict(inputs), sweep_radius=radius)
    res = {'out_j_ion': np.asarray(out['j_ion'], dtype=np.float64),
           'out_div_angle': np.asarray(out['div_angle'], dtype=np.float64)}
    if 'T_c' in out:
        res['out_T_c'] = np.asarray(out['T_c'], dtype=np.float64)
    coords = out['j_ion_coords']
    res['out_coords0'] = np.asarray(coords.flat[0], dtype=np.float64)
    res['out_coords_shape'] = np.asarray(coords.shape, dtype=np.int64)
    return res


def main():
    cathode, plume, thruster, const = _load_reference()

    # ---- cathode: tests/test_cathode.py:19-21 ranges, seeded --------------------------------------
    rng = np.random.default_rng(20260101)
    N = 1024
    cin = {'P_b': 10 ** (rng.random(N) * 4 - 8), 'V_a': rng.random(N) * 200 + 200,
           'T_e': rng.random(N) * 4 + 1, 'V_vac': rng.random(N) * 60,
           'Pstar': rng.random(N) * 90e-6 + 10e-6, 'P_T': rng.random(N) * 90e-6 + 10e-6}
    vcc = cathode.cathode_coupling(dict(cin))['V_cc']
    _save('cathode_random', **{f'in_{k}': v for k, v in cin.items()}, out_V_cc=vcc)

    # cathode edge cases: both clips, NaN propagation, huge pressure ratio, scalar call, sweep
    ce = {'P_b':   np.array([1e-5, 1e-4, 1e-8, 1e-4, np.nan, 1e-5, 1e-3, 1e-6]),
          'V_a':   np.array([300., 20., 300., 300., 300., 300., 300., 5.]),
          'T_e':   np.array([3., 5., 1., 5., 3., np.nan, 5., 3.]),
          'V_vac': np.array([30., 59., 0., 0., 30., 30., 0., 30.]),
          'Pstar': np.array([2e-5, 1e-5, 1e-4, 1e-5, 2e-5, 2e-5, 1e-6, 2e-5]),
          'P_T':   np.array([5e-5, 1e-5, 1e-4, 1e-5, 5e-5, 5e-5, 1e-6, 5e-5])}
    with np.errstate(all='ignore'):
        vcc_e = cathode.cathode_coupling(dict(ce))['V_cc']
    scalar_in = {'P_b': 10e-6, 'V_a': 300, 'T_e': 3, 'V_vac': 30, 'Pstar': 20e-6, 'P_T': 50e-6}  # test_cathode.py:14
    vcc_s = cathode.cathode_coupling(dict(scalar_in))['V_cc']
    sweep_pb = 10 ** np.linspace(-6, -4, 100)                                                      # test_cathode.py:27
    sweep_in = {'P_b': sweep_pb, 'V_a': np.full(100, 300.), 'T_e': np.full(100, 1.33), 'V_vac': np.full(100, 31.6),
                'Pstar': np.full(100, 24.6e-6), 'P_T': np.full(100, 10.2e-6)}
    vcc_sw = cathode.cathode_coupling(dict(sweep_in))['V_cc']
    _save('cathode_edges', **{f'in_{k}': v for k, v in ce.items()}, out_V_cc=vcc_e,
          scalar_out_V_cc=vcc_s, sweep_in_P_b=sweep_pb, sweep_out_V_cc=vcc_sw)

    # wild cathode inputs: zero / negative pressures (log of a non-positive number), huge ratios, negative voltages
    rng = np.random.default_rng(20260108)
    N = 400
    pick = lambda vals, size: rng.choice(np.asarray(vals, dtype=np.float64), size=size)                 # noqa: E731
    cw = {'P_b': 10 ** rng.uniform(-10, -1, N) * pick([1, 1, 1, 0, -1], N), 'V_a': rng.uniform(-50, 500, N),
          'T_e': rng.uniform(-1, 8, N), 'V_vac': rng.uniform(-20, 100, N),
          'Pstar': 10 ** rng.uniform(-8, -3, N) * pick([1, 1, 1, 0, -1], N),
          'P_T': 10 ** rng.uniform(-8, -3, N) * pick([1, 1, 1, 0, -1], N)}
    with np.errstate(all='ignore'):
        vw = cathode.cathode_coupling(dict(cw))['V_cc']
    _save('cathode_wild', **{f'in_{k}': v for k, v in cw.items()}, out_V_cc=vw)

    # ---- plume: tests/test_plume.py:19-31 ranges (reaches the invalid alpha1<=0 branch), 5 radii ----
    rng = np.random.default_rng(20260102)
    N = 96
    pin = {'P_b': 10 ** (rng.random(N) * 4 - 8), 'c0': rng.random(N) * 0.8 + 0.1, 'c1': rng.random(N) * 0.8 + 0.1,
           'c2': rng.random(N) * 30 - 15, 'c3': rng.random(N) + 0.1, 'c4': 10 ** (rng.random(N) * 4 + 18),
           'c5': 10 ** (rng.random(N) * 4 + 14), 'sigma_cex': rng.random(N) * 7e-20 + 51e-20,
           'I_B0': rng.random(N) * 6 + 2}
    radii = rng.random(5) * 0.2 + 1
    _save('plume_random_r5', **{f'in_{k}': v for k, v in pin.items()}, radii=radii, **_plume_out(plume, pin, radii))

    # same ranges, R = 1 (squeezed), with thrust, more samples
    rng = np.random.default_rng(20260103)
    N = 512
    pin = {'P_b': 10 ** (rng.random(N) * 4 - 8), 'c0': rng.random(N) * 0.8 + 0.1, 'c1': rng.random(N) * 0.8 + 0.1,
           'c2': rng.random(N) * 30 - 15, 'c3': rng.random(N) + 0.1, 'c4': 10 ** (rng.random(N) * 4 + 18),
           'c5': 10 ** (rng.random(N) * 4 + 14), 'sigma_cex': rng.random(N) * 7e-20 + 51e-20,
           'I_B0': rng.random(N) * 6 + 2, 'T': rng.random(N) * 0.1 + 0.02}
    _save('plume_random_r1', **{f'in_{k}': v for k, v in pin.items()}, radii=np.array([1.0]),
          **_plume_out(plume, pin, 1.0))

    # Appendix-A priors (pem_v0_SPT-100.yml:221-270), config-2 style inputs, R = 1 at 1.0 m
    rng = np.random.default_rng(20260104)
    N = 512
    pin = {'P_b': 10 ** (rng.random(N) * 4 - 8), 'c0': rng.random(N), 'c1': rng.random(N) * 0.8 + 0.1,
           'c2': rng.random(N) * 30 - 15, 'c3': rng.random(N) * (1.570796 - 0.2) + 0.2,
           'c4': 10 ** (rng.random(N) * 4 + 18), 'c5': 10 ** (rng.random(N) * 4 + 14),
           'sigma_cex': rng.random(N) * 7e-20 + 51e-20, 'I_B0': rng.random(N) * 6 + 2,
           'T': rng.random(N) * 0.1 + 0.02}
    _save('plume_priors_r1', **{f'in_{k}': v for k, v in pin.items()}, radii=np.array([1.0]),
          **_plume_out(plume, pin, 1.0))

    # normaliser sweep: alpha1 = c3 over [1e-3, pi/2] (c2 = 0), alpha2 = alpha1/c1 up to ~52 (c1 down to 0.03)
    a1 = np.concatenate([np.logspace(-3, np.log10(np.pi / 2), 120), np.full(40, np.pi / 2)])
    c1 = np.concatenate([np.tile([0.9, 0.5, 0.25, 0.1], 30), np.linspace(1.0, 0.03, 40)])
    N = a1.size
    pin = {'P_b': np.full(N, 1e-5), 'c0': np.full(N, 0.4), 'c1': c1, 'c2': np.zeros(N), 'c3': a1,
           'c4': np.full(N, 1e20), 'c5': np.full(N, 1e16), 'sigma_cex': np.full(N, 55e-20),
           'I_B0': np.full(N, 3.0), 'T': np.full(N, 0.08)}
    _save('plume_alpha_sweep', **{f'in_{k}': v for k, v in pin.items()}, radii=np.array([1.0]),
          **_plume_out(plume, pin, 1.0))

    # edge cases: alpha1 = 0, < 0, clipped > pi/2, c0 = 0 / 1 / > 1, I_B0 = 0,
    # P_b at both ends, NaN input, alpha2 past the reference's erfi overflow (c1 = 0.02), c1 = 0, c4 = c5 = 0
    base = {'P_b': 1e-5, 'c0': 0.5, 'c1': 0.5, 'c2': -8.0, 'c3': 0.3, 'c4': 1e20, 'c5': 1e16,
            'sigma_cex': 55e-20, 'I_B0': 3.0, 'T': 0.08}
    edits = [{}, {'c2': 0., 'c3': 0.}, {'c2': 0., 'c3': -0.3}, {'c2': 0., 'c3': 0.3}, {'c2': 15., 'c3': 1.5, 'P_b': 1e-4},
             {'c0': 0.}, {'c0': 1.}, {'c0': 1.5}, {'c0': -0.2}, {'I_B0': 0.}, {'I_B0': -1.}, {'P_b': 1e-8}, {'P_b': 1e-4},
             {'c3': np.nan}, {'P_b': np.nan}, {'I_B0': np.nan}, {'c2': 0., 'c3': 1.5, 'c1': 0.02},
             {'c2': 0., 'c3': 1.5, 'c1': 0.03}, {'c2': 0., 'c3': 1.5, 'c1': 0.}, {'c4': 0., 'c5': 0.},
             {'c2': 0., 'c3': 0.24}, {'c2': 0., 'c3': 0.26}, {'c2': 0., 'c3': 1e-4}, {'c1': 1.0}, {'c1': -0.5},
             {'sigma_cex': 0.}, {'c2': 0., 'c3': 1.5, 'c1': 0.0295}, {'c2': -15., 'c3': 0.2, 'P_b': 1e-4}]
    pin = {k: np.array([{**base, **e}[k] for e in edits], dtype=np.float64) for k in base}
    _save('plume_edges', **{f'in_{k}': v for k, v in pin.items()}, radii=np.array([1.0]),
          **_plume_out(plume, pin, 1.0))
    rr = np.array([0.5, 1.0, 2.5])
    _save('plume_edges_r3', **{f'in_{k}': v for k, v in pin.items()}, radii=rr, **_plume_out(plume, pin, rr))

    # exact cancellation in the two affine set-up expressions (plume.py:56,59): c3 = -fl(c2 P_B) makes alpha1 exactly 0 in
    # the reference (invalid sample, NaN normaliser) where a fused multiply-add would leave the rounding error of the
    # product, of either sign; its two neighbours in c3 give alpha1 = +-1 ulp(c3).  Likewise c5 = -fl(c4 P_B): n = 0,
    # exp(0) = 1, j_cex = 0 exactly.
    rng = np.random.default_rng(20260211)
    N = 96
    Pb = 10 ** rng.uniform(-7, -4, N)
    c2 = rng.uniform(-15, 15, N)
    c2[c2 == 0] = 1.0
    prod = c2 * (Pb * TORR_2_PA)
    c3 = -prod
    c3[1::3] = np.nextafter(c3[1::3], np.inf)
    c3[2::3] = np.nextafter(c3[2::3], -np.inf)
    c4 = 10 ** rng.uniform(18, 22, N)
    c5 = 10 ** rng.uniform(14, 18, N)
    half = np.arange(N) >= N // 2                 # second half: the density cancels instead, alpha1 stays ordinary
    c3[half] = rng.uniform(0.2, 1.5, half.sum())
    c5[half] = -(c4[half] * (Pb[half] * TORR_2_PA))
    nudge = np.flatnonzero(half)[1::3]
    c5[nudge] = np.nextafter(c5[nudge], np.inf)
    pin = {'P_b': Pb, 'c0': rng.uniform(0, 1, N), 'c1': rng.uniform(0.1, 0.9, N), 'c2': c2, 'c3': c3, 'c4': c4, 'c5': c5,
           'sigma_cex': rng.uniform(51e-20, 58e-20, N), 'I_B0': rng.uniform(2, 8, N), 'T': rng.uniform(0.02, 0.2, N)}
    _save('plume_cancel', **{f'in_{k}': v for k, v in pin.items()}, radii=np.array([1.0]), **_plume_out(plume, pin, 1.0))
    _save('plume_cancel_r3', **{f'in_{k}': v for k, v in pin.items()}, radii=rr, **_plume_out(plume, pin, rr))

    # wild inputs far outside the priors (signs, zeros, huge/tiny magnitudes): NaN / inf / invalid patterns must agree
    rng = np.random.default_rng(20260107)
    N = 600
    pick = lambda vals, size: rng.choice(np.asarray(vals, dtype=np.float64), size=size)                 # noqa: E731
    pin = {'P_b': 10 ** rng.uniform(-10, -2, N) * pick([1, 1, 1, 0], N),
           'c0': rng.uniform(-0.5, 1.5, N), 'c1': rng.uniform(-1.0, 1.2, N) * pick([1, 1, 1, 1, 0.01], N),
           'c2': rng.uniform(-40, 40, N), 'c3': rng.uniform(-0.5, 2.5, N) * pick([1, 1, 1, 0], N),
           'c4': 10 ** rng.uniform(15, 24, N) * pick([1, 1, 0], N), 'c5': 10 ** rng.uniform(10, 20, N) * pick([1, 1, 0], N),
           'sigma_cex': rng.uniform(0, 1e-18, N), 'I_B0': rng.uniform(-2, 10, N) * pick([1, 1, 1, 0], N),
           'T': rng.uniform(-0.05, 0.2, N)}
    _save('plume_wild', **{f'in_{k}': v for k, v in pin.items()}, radii=np.array([1.0]), **_plume_out(plume, pin, 1.0))

    # shape semantics: all-scalar inputs (leading axis of 1), loop shape (3, 4) with R = 2, no thrust
    sc = {k: v for k, v in base.items() if k != 'T'}
    o_sc = _plume_out(plume, sc, 1.0)
    rng = np.random.default_rng(20260105)
    shp = (3, 4)
    pin = {'P_b': 10 ** (rng.random(shp) * 4 - 8), 'c0': rng.random(shp), 'c1': rng.random(shp) * 0.8 + 0.1,
           'c2': rng.random(shp) * 30 - 15, 'c3': rng.random(shp) * 1.3 + 0.2, 'c4': 10 ** (rng.random(shp) * 4 + 18),
           'c5': 10 ** (rng.random(shp) * 4 + 14), 'sigma_cex': rng.random(shp) * 7e-20 + 51e-20,
           'I_B0': rng.random(shp) * 6 + 2}
    o_nd = _plume_out(plume, pin, np.array([0.8, 1.3]))
    _save('plume_shapes', **{f'scalar_in_{k}': np.float64(v) for k, v in sc.items()},
          **{f'scalar_{k}': v for k, v in o_sc.items()},
          **{f'nd_in_{k}': v for k, v in pin.items()}, nd_radii=np.array([0.8, 1.3]),
          **{f'nd_{k}': v for k, v in o_nd.items()})

    # tests/test_plume.py:64-98 pressure sweep (total current invariant = I_B0 = 3 A)
    ps = 10 ** np.linspace(-6, -4, 100)
    pin = {'P_b': ps, 'c0': 0.1, 'c1': 0.7, 'c2': -8.0, 'c3': 0.2, 'c4': 1e20, 'c5': 1e16, 'sigma_cex': 55e-20, 'I_B0': 3}
    _save('plume_pressure_sweep', in_P_b=ps, **{f'in_{k}': np.float64(v) for k, v in pin.items() if k != 'P_b'},
          radii=np.array([1.0]), **_plume_out(plume, pin, 1))

    # ---- thruster host-side pre/post-processing (thruster.py:93-181, 266-276) ---------------------------
    fid = []
    for mf in [(0, 0), (1, 0), (2, 2), (), (3, 1)]:
        for cfg in [{}, {'config': {'domain': [0, 0.08], 'discharge_voltage': 300., 'cathode_coupling_voltage': 30.,
                                    'propellant': 'Xenon'}},
                    {'config': {'domain': [0, 0.1], 'discharge_voltage': 250., 'cathode_coupling_voltage': 12.5,
                                'propellant': 'Krypton'}}]:
            r = thruster._default_model_fidelity(mf, cfg)
            fid.append({'model_fidelity': list(mf), 'json_config': cfg, 'result': r})
    # path-blazing conversion (tests/test_thruster.py:43-67 style, our own inputs)
    p2j = {'a': ['config', 'x'], 'b': ['config', 'lst', 2], 'c': ['config', 'deep', 'er', 1, 'leaf'],
           'o1': ['output', 'average', 'thrust'], 'o2': ['output', 'average', 'ui', 0], 'o3': ['output', 'missing', 'q']}
    jd = {'config': {'keep': 1, 'lst': [0, 1, 2]}}
    thruster._convert_to_julia({'a': 1.5, 'b': 'two', 'c': [3, 4]}, jd, p2j)
    try:   # a list-index LEAF is not blazed: the reference raises IndexError (thruster.py:118)
        thruster._convert_to_julia({'b': 1}, {'config': {}}, p2j)
        leaf_error = None
    except Exception as e:
        leaf_error = type(e).__name__
    try:
        thruster._convert_to_julia({'zz': 1}, {}, p2j)
        key_error = None
    except Exception as e:
        key_error = type(e).__name__
    back = thruster._convert_to_pem({'output': {'average': {'thrust': 0.08, 'ui': [[1., 2., 3.]]}}}, p2j)
    fmt = thruster._format_hallthruster_jl_input(
        {'V_a': 310.0, 'a_1': 0.01, 'a_2': 20.0, 'V_cc': 25.0, 'mdot_a': 5e-6}, thruster.PEM_TO_JULIA,
        thruster={'name': 'X'}, config={'anom_model': {'type': 'LogisticPressureShift',
                                                        'model': {'type': 'TwoZoneBohm', 'c1': 0.00625, 'c2': 0.0625}},
                                         'domain': [0, 0.08], 'propellant': 'Xenon'},
        simulation={'duration': 0.002}, postprocess={}, model_fidelity=(1, 0))
    fmt_g = thruster._format_hallthruster_jl_input(
        {'anom_max': 50.0, 'anom_min': 0.005}, thruster.PEM_TO_JULIA, thruster=None,
        config={'anom_model': {'type': 'GaussianBohm', 'hall_min': 0.00625, 'hall_max': 0.0625}},
        model_fidelity=None)
    # ---- scripts/gen_data.py:104-174 `_filter_outputs` (NaN + IQR masks).  The script imports amisc at module
    # level (absent), so only the two function definitions are pulled out of its AST and executed, with the
    # one amisc name they read (COORDS_STR_ID, the '_coords' suffix of plume.py:157) supplied here.
    import ast
    src = (REF.parents[1] / 'scripts' / 'gen_data.py').read_text()
    tree = ast.parse(src)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('_object_to_numeric', '_filter_outputs')]
    ns = {'np': np, 'COORDS_STR_ID': '_coords'}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), 'gen_data.py[extract]', 'exec'), ns)
    rng = np.random.default_rng(20260106)
    N = 400
    fo = {'V_cc': rng.normal(30, 2, N), 'T': rng.normal(0.08, 0.005, N), 'div_angle': rng.uniform(0.2, 0.6, N),
          'j_ion': np.abs(rng.normal(0, 1, (N, 91))) + np.linspace(5, 0.1, 91), 'u_ion': rng.normal(10, 1, (N, 7, 3))}
    fo['V_cc'][[3, 77]] = [55.0, -10.0]                  # scalar outliers
    fo['T'][5] = np.nan                                  # NaN sample
    fo['j_ion'][10] += 50.0                              # whole profile off -> field outlier
    fo['j_ion'][11, :60] += 50.0                         # 60/91 = 66 % of entries off -> NOT an outlier (threshold 75 %)
    fo['j_ion'][12, :70] += 50.0                         # 70/91 = 77 % -> outlier
    fo['j_ion'][13, 4] = np.nan
    fo['u_ion'][20] -= 30.0
    fo['j_ion_coords'] = np.tile(np.linspace(0, 1, 91), (N, 1))   # must be skipped
    fo['errors'] = np.zeros(N)                                   # must be skipped
    with np.errstate(all='ignore'):
        nan_q15, out_q15 = ns['_filter_outputs'](dict(fo), iqr_factor=1.5)
        nan_q3, out_q3 = ns['_filter_outputs'](dict(fo), iqr_factor=3.0)
    _save('filter_outputs', **{f'in_{k}': v for k, v in fo.items()},
          **{f'nan15_{k}': v for k, v in nan_q15.items()}, **{f'out15_{k}': v for k, v in out_q15.items()},
          **{f'nan30_{k}': v for k, v in nan_q3.items()}, **{f'out30_{k}': v for k, v in out_q3.items()})

    _hallthruster_jl_golden(thruster)
    _pem_v0_variable_table()

    with open(OUT / 'thruster_host.json', 'w') as fd:
        json.dump({'fidelity': fid, 'convert_map': p2j, 'convert_to_julia': jd, 'convert_to_pem': back,
                   'format_twozone': fmt, 'format_gaussian': fmt_g,
                   'leaf_index_error': leaf_error, 'unknown_key_error': key_error, 'PEM_TO_JULIA_keys': sorted(thruster.PEM_TO_JULIA),
                   'constants': {'AVOGADRO_CONSTANT': const.AVOGADRO_CONSTANT,
                                 'FUNDAMENTAL_CHARGE': const.FUNDAMENTAL_CHARGE,
                                 'MOLECULAR_WEIGHTS': const.MOLECULAR_WEIGHTS}}, fd, indent=1)
    print('wrote thruster_host.json')


def fake_run_simulation(json_input, jl_env=None, jl_script=None, **kwargs):
    """What tests/sim_hallthruster.jl does to its input file, in numpy (the formulas of its lines 35-47; Julia is absent):
    the stand-in for `run_hallthruster_jl` when the reference's `hallthruster_jl` is run here.  The output goes through
    a JSON file and back, as it does in the reference, so that the golden values are what json.load returns."""
    doc = json_input
    cfg = doc['config']
    V, Vc, md, c1 = cfg['discharge_voltage'], cfg['cathode_coupling_voltage'], cfg['anode_mass_flow_rate'], cfg['anom_model']['model']['c1']
    nc, dom = doc['simulation']['grid']['num_cells'], cfg['domain']
    q, mi = 1.6e-19, 2.18e-25
    beam = (q / mi) * md
    ceff = 1 - c1 * 2
    Id = beam / ceff
    v = float(np.sqrt(2 * q * (V - Vc) / mi))
    thrust = md * v
    z = [dom[0] + (dom[1] - dom[0]) * (i / (nc - 1)) for i in range(nc)]
    ui = [v / (1 + float(np.exp(-100 * (zz - 0.04)))) for zz in z]
    out = {'output': {'average': {'thrust': thrust, 'ion_current': beam, 'current_eff': ceff, 'discharge_current': Id, 'v_exh': v,
                                  'mass_eff': 1 - c1 * 5, 'voltage_eff': 1 - c1 * 2,
                                  'anode_eff': 0.5 * thrust ** 2 / (md * V * Id), 'ui': [ui], 'z': z}},
           'config': cfg, 'simulation': doc['simulation'], 'postprocess': doc['postprocess']}
    if target := doc['postprocess'].get('output_file'):
        with open(target, 'w') as fd:
            json.dump(out, fd)
    return json.loads(json.dumps(out))


def _pem_v0_variable_table():
    """The PEM-v0 variable table as DATA: scripts/pem_v0/pem_v0_SPT-100.yml read with
    constructors that ignore amisc's tags (`!System`, `!Component`, `!Variable`, `!!python/name:`), every variable of every
    component with the fields the sampling loops use -- name, category, nominal, domain, distribution, norm, units,
    compression -- exactly as the file spells them (strings stay strings: `U(1, 5)`, `(1.0e-8, 1.0e-4)`, `linear(1e6)`).
    tests/test_variable_table.py parses them and holds sampling.PEM_V0_PRIORS, system.PemV0System and the synthetic inputs of
    bench.py / tests/_inputs.py to the result."""
    import yaml

    class Loader(yaml.SafeLoader):
        pass

    def plain(loader, suffix, node):
        if isinstance(node, yaml.MappingNode):
            return loader.construct_mapping(node, deep=True)
        if isinstance(node, yaml.SequenceNode):
            return loader.construct_sequence(node, deep=True)
        return suffix if node.value == '' else loader.construct_scalar(node)     # !!python/name:a.b.c -> 'a.b.c'
    Loader.add_multi_constructor('!', plain)
    Loader.add_multi_constructor('tag:yaml.org,2002:python/name:', plain)
    yml = REF.parents[1] / 'scripts' / 'pem_v0' / 'pem_v0_SPT-100.yml'
    with open(yml, 'r', encoding='utf-8') as fd:
        system = yaml.load(fd, Loader=Loader)
    keep = ('name', 'category', 'nominal', 'domain', 'distribution', 'norm', 'units', 'compression')
    table = {'source': 'scripts/pem_v0/pem_v0_SPT-100.yml', 'system': system['name'], 'hallmd_version': system['hallmd_version'],
             'components': []}
    for comp in system['components']:
        rec = {'name': comp['name'], 'model': comp['model'], 'vectorized': bool(comp.get('vectorized', False))}
        for k in ('sweep_radius', 'model_fidelity', 'thruster'):
            if k in comp:
                rec[k] = comp[k]
        for side in ('inputs', 'outputs'):
            rec[side] = [{k: v[k] for k in keep if k in v} for v in comp[side]]
        table['components'].append(rec)
    with open(OUT / 'pem_v0_variables.json', 'w') as fd:
        json.dump(table, fd, indent=1, sort_keys=True)
    print('wrote pem_v0_variables.json')


def _hallthruster_jl_golden(thruster):
    """Run the reference's own `hallthruster_jl` (thruster.py:378-512) with its Julia launcher replaced by
    `fake_run_simulation`, and store what it returns / raises: keys, QoIs, output_path shape, the two filters."""
    import re
    import tempfile
    thruster.run_hallthruster_jl = fake_run_simulation
    cfg = {'anom_model': {'type': 'LogisticPressureShift', 'model': {'type': 'TwoZoneBohm', 'c1': 0.008, 'c2': 0.08}}, 'domain': [0, 0.08]}
    sim = {'grid': {'type': 'EvenGrid', 'num_cells': 100}, 'duration': 1e-3, 'dt': 1e-9}
    cases = [
        # tests/test_thruster.py:70-94, with a dict for the device instead of the downloaded file
        dict(name='reference_test', inputs={'V_a': 250, 'V_cc': 25, 'mdot_a': 3.5e-6}, kwargs=dict(config=cfg, simulation=sim, thruster={'name': 'SPT-100'}, postprocess={'average_start_time': 0.5e-3}), with_path=True),
        dict(name='pem_inputs', inputs={'V_a': 300.0, 'V_cc': 32.5, 'mdot_a': 5.16e-6, 'a_1': 0.0068, 'a_2': 14.6, 'P_b': 3.5e-5, 'T_e': 1.33},
             kwargs=dict(config={'anom_model': {'type': 'LogisticPressureShift', 'model': {'type': 'TwoZoneBohm', 'c1': 0.00625, 'c2': 0.0625}}, 'domain': [0, 0.08]},
                         thruster=None, model_fidelity=(1, 1)), with_path=False),
        dict(name='default_fidelity', inputs={'V_a': 280.0, 'V_cc': 20.0, 'mdot_a': 4e-6, 'a_1': 0.01}, kwargs=dict(config={'domain': [0, 0.1]}, thruster=None), with_path=True),
        dict(name='negative_flow', inputs={'V_a': 300.0, 'V_cc': 30.0, 'mdot_a': -5e-6, 'a_1': 0.01}, kwargs=dict(config={'domain': [0, 0.08]}, thruster=None), with_path=False),
        dict(name='shock', inputs={'V_a': 300.0, 'V_cc': 30.0, 'mdot_a': 5e-6, 'a_1': 0.01}, kwargs=dict(config={'domain': [0, 0.08]}, thruster=None, shock_threshold=0.09), with_path=False),
        dict(name='no_shock', inputs={'V_a': 300.0, 'V_cc': 30.0, 'mdot_a': 5e-6, 'a_1': 0.01}, kwargs=dict(config={'domain': [0, 0.08]}, thruster=None, shock_threshold=0.04), with_path=False),
        dict(name='pem_to_julia_extra', inputs={'V_a': 300.0, 'V_cc': 30.0, 'mdot_a': 5e-6, 'a_1': 0.01, 'my_in': 7.0},
             kwargs=dict(config={'domain': [0, 0.08]}, thruster=None, pem_to_julia={'my_in': ['config', 'my', 'input'], 'v_exh': ['output', 'average', 'v_exh']}), with_path=False),
    ]
    out = []
    for case in cases:
        rec = {'name': case['name'], 'inputs': case['inputs'], 'kwargs': {k: v for k, v in case['kwargs'].items()}, 'with_path': case['with_path']}
        with tempfile.TemporaryDirectory() as tmp:
            kw = dict(case['kwargs'])
            if case['with_path']:
                kw['output_path'] = tmp
            try:
                res = dict(thruster.hallthruster_jl(dict(case['inputs']), **kw))
                cost = res.pop('model_cost')
                assert isinstance(cost, float) and cost >= 0
                rec['keys'] = sorted(list(res) + ['model_cost'])
                if case['with_path']:
                    path = res.pop('output_path')
                    rec['output_path_pattern'] = re.sub(r'_[A-Z0-9]{4}\.json$', '_XXXX.json', path)
                    with open(Path(tmp) / path) as fd:
                        rec['file_average_keys'] = sorted(json.load(fd)['output']['average'])
                sim_out = res.pop('thruster_output')
                sim_out['postprocess'].pop('output_file', None)
                rec['thruster_output'] = sim_out
                rec['result'] = res
            except Exception as e:                     # noqa: BLE001 -- type and message are what is stored
                rec['raises'] = type(e).__name__
                rec['message'] = str(e)
        out.append(rec)
    # utils.load_thruster on a small device directory (the reference's own loader; hallmd.devices is not in the tree)
    sys.path.insert(0, str(REF.parent))
    from hallmd.utils import load_thruster
    with tempfile.TemporaryDirectory() as tmp:
        dev = Path(tmp) / 'MyDevice'
        (dev / 'fields').mkdir(parents=True)
        (dev / 'bfield.csv').write_text('z,B\n0,0.01\n')
        (dev / 'fields' / 'extra.csv').write_text('x\n')
        spec = {'name': 'MyDevice', 'geometry': {'channel_length': 0.025, 'inner_radius': 0.0345, 'outer_radius': 0.05},
                'magnetic_field': {'file': 'bfield.csv'}, 'other': {'table': 'fields/extra.csv'}, 'shielded': False}
        import yaml
        (dev / 'thruster.yml').write_text(yaml.safe_dump(spec))
        loaded = load_thruster(dev)
        loaded_txt = json.dumps(loaded).replace(str(dev.resolve()), '<DEVICE>')
    # ... and where the reference's walk is particular (src/hallmd/utils.py:67-85): a file name referenced twice (the first
    # occurrence in depth-first key order is the only one replaced), a reference two levels deep (replaced), one three levels
    # deep (the walk restarts from the top-level dict at every key: KeyError), names inside lists (never looked at)
    more = []
    layouts = {
        'twice': {'name': 'D', 'magnetic_field': {'file': 'bfield.csv'}, 'backup': {'file': 'bfield.csv'}, 'plain': 'bfield.csv'},
        'two_levels': {'name': 'D', 'a': {'b': 'fields/extra.csv'}, 'shielded': True},
        'three_levels': {'name': 'D', 'a': {'b': {'c': 'bfield.csv'}}},
        'three_levels_by_path': {'name': 'D', 'a': {'b': {'c': 'fields/extra.csv'}}},
        'in_a_list': {'name': 'D', 'files': ['bfield.csv', 'fields/extra.csv'], 'magnetic_field': {'file': 'bfield.csv'}},
        'bare_name_of_nested_file': {'name': 'D', 'other': {'table': 'extra.csv'}},
        'top_level': {'name': 'D', 'file': 'bfield.csv'},
        'json_spec': {'name': 'D', 'magnetic_field': {'file': 'bfield.csv'}},
    }
    for label, lay in layouts.items():
        with tempfile.TemporaryDirectory() as tmp:
            dev = Path(tmp) / 'Dev'
            (dev / 'fields').mkdir(parents=True)
            (dev / 'bfield.csv').write_text('z,B\n0,0.01\n')
            (dev / 'fields' / 'extra.csv').write_text('x\n')
            fname = 'thruster.json' if label == 'json_spec' else 'thruster.yml'
            text = json.dumps(lay) if label == 'json_spec' else yaml.safe_dump(lay, sort_keys=False)
            (dev / fname).write_text(text)
            rec = {'label': label, 'spec_text': text, 'filename': fname}      # the text: key ORDER decides which mention is first
            try:
                got = load_thruster(dev, fname)
                rec['loaded'] = json.loads(json.dumps(got).replace(str(dev.resolve()), '<DEVICE>'))
            except Exception as e:                     # noqa: BLE001
                rec['raises'] = type(e).__name__
            more.append(rec)
    with open(OUT / 'hallthruster_jl.json', 'w') as fd:
        json.dump({'cases': out, 'device_spec': spec, 'device_loaded': json.loads(loaded_txt), 'device_cases': more}, fd, indent=1, sort_keys=True)
    print('wrote hallthruster_jl.json')


if __name__ == '__main__':
    main()
