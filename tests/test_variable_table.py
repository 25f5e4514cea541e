"""The PEM-v0 variable table (scripts/pem_v0/pem_v0_SPT-100.yml:9-284) is the one input of every BASELINE
configuration that the reference holds as data.  tests/golden/pem_v0_variables.json is that file read by
tests/golden/make_golden.py (tag-ignoring YAML constructors, strings kept as the file spells them); everything in this
repository that restates the table by hand is held to it here: `sampling.PEM_V0_PRIORS`, `system.PemV0System`'s variables
(category, norm), the synthetic inputs of `bench.synth_inputs` and of `tests/_inputs.py`."""
import ast
import json
import math
import re
from pathlib import Path

import numpy as np
import pytest

from _inputs import cathode_inputs, coupled_inputs, plume_inputs

ROOT = Path(__file__).resolve().parents[1]
TABLE = json.loads((ROOT / 'tests' / 'golden' / 'pem_v0_variables.json').read_text())


def _variables():
    """name -> merged record over the components (a variable is spelled out once and referred to by name afterwards)"""
    merged = {}
    for comp in TABLE['components']:
        for side in ('inputs', 'outputs'):
            for v in comp[side]:
                rec = merged.setdefault(v['name'], {})
                for k, val in v.items():
                    assert rec.get(k, val) == val, (v['name'], k)        # no component contradicts another
                    rec[k] = val
    return merged


def _pair(text):
    a, b = ast.literal_eval(text)
    return float(a), float(b)


def _sampling_range(rec):
    """(kind, lo, hi) the sampling loop draws a variable from: `gen_data.py:238` samples calibration / nuisance variables from
    their distribution and the others uniformly over their domain, in normalised space (log10 where `norm: log10`)."""
    dist = rec.get('distribution', '')
    m = re.fullmatch(r'(U|Uniform|LogUniform)\((.*)\)', dist)
    if rec.get('category') in ('calibration', 'nuisance') and m:
        lo, hi = _pair('(' + m.group(2) + ')')
        return ('loguniform' if m.group(1) == 'LogUniform' else 'uniform'), lo, hi
    lo, hi = _pair(rec['domain'])
    return ('loguniform' if rec.get('norm') == 'log10' else 'uniform'), lo, hi


def test_the_fixture_is_the_whole_file():
    assert TABLE['source'] == 'scripts/pem_v0/pem_v0_SPT-100.yml' and TABLE['hallmd_version'] == '0.3.0'
    comps = {c['name']: c for c in TABLE['components']}
    assert list(comps) == ['Cathode', 'Thruster', 'Plume']
    assert comps['Cathode']['model'] == 'hallmd.models.cathode.cathode_coupling' and comps['Cathode']['vectorized']
    assert comps['Plume']['model'] == 'hallmd.models.plume.current_density' and comps['Plume']['sweep_radius'] == 1.0
    assert comps['Thruster']['model'] == 'hallmd.models.thruster.hallthruster_jl' and comps['Thruster']['model_fidelity'] == '(2, 2)'
    v = _variables()
    inputs = {x['name'] for c in TABLE['components'] for x in c['inputs']}
    assert len(inputs - {'V_cc', 'I_B0'}) == 21 and len(inputs) == 23                     # 21 free inputs + 2 coupling variables
    assert [x['name'] for x in comps['Cathode']['inputs']] == ['P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T']
    assert [x['name'] for x in comps['Plume']['inputs']] == ['P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex', 'I_B0']
    assert v['j_ion']['compression'] == {'method': 'svd', 'reconstruction_tol': 0.01} and v['j_ion']['norm'] == 'log10'
    assert v['u_ion']['norm'] == 'linear(1.0e-3)'


def test_prior_table_is_the_yaml():
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    from hallthrusterpem_amd.sampling import LOGUNIFORM, PEM_V0_PRIORS, UNIFORM
    v = _variables()
    assert set(COUPLED_INPUTS) <= set(v)
    for name in COUPLED_INPUTS:
        kind, lo, hi = _sampling_range(v[name])
        p = PEM_V0_PRIORS[name]
        if kind == 'loguniform':
            assert p.kind == LOGUNIFORM and p.a == math.log10(lo) and p.b == math.log10(hi), name
        else:
            assert p.kind == UNIFORM and (p.a, p.b) == (lo, hi), name
        # the line numbers each entry cites exist and the yml really names the variable there
    # the variables of the real thruster solver that the analytic test double does not take are not in the table of 15
    assert {'u_n', 'l_t', 'a_2', 'dz', 'z0', 'p0'} == {x['name'] for c in TABLE['components'] for x in c['inputs']} - set(COUPLED_INPUTS) - {'V_cc', 'I_B0'}


def test_system_variables_are_the_yaml():
    from hallthrusterpem_amd import system
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    v = _variables()
    for name in COUPLED_INPUTS:
        assert system.CATEGORY.get(name, 'calibration') == v[name]['category'], name
        norm = v[name].get('norm')
        if norm is None:
            assert name not in system.NORM, name
        elif norm == 'log10':
            assert system.NORM[name] == 'log10', name
        else:
            scale = float(re.fullmatch(r'linear\((.*)\)', norm).group(1))
            assert system.NORM[name] == ('linear', scale), name
    # outputs the coupled graph produces, with the yml's norm for the compressed field
    assert v['j_ion']['norm'] == 'log10' and {'V_cc', 'I_B0', 'T', 'j_ion', 'div_angle'} <= set(system.OUTPUTS)


def _check_ranges(x: dict, v: dict, overrides=None):
    """every array of x stays inside the yml's sampling range, reaches both ends of it, and is (log-)uniform there"""
    for name, arr in x.items():
        kind, lo, hi = (overrides or {}).get(name) or _sampling_range(v[name])
        u = (np.log10(arr) - math.log10(lo)) / (math.log10(hi) - math.log10(lo)) if kind == 'loguniform' else (arr - lo) / (hi - lo)
        assert u.min() >= -1e-12 and u.max() <= 1 + 1e-12, (name, arr.min(), arr.max(), lo, hi)
        assert u.min() < 0.01 and u.max() > 0.99 and abs(u.mean() - 0.5) < 0.02 and abs(u.std() - 12 ** -0.5) < 0.02, name


def test_test_inputs_are_the_yaml():
    v = _variables()
    _check_ranges(coupled_inputs(20_000, seed=3), v)
    # cathode / plume stand-alone inputs: the yml where it has an entry; the coupling variable I_B0 and the extra T are the
    # ranges of the reference's own tests (tests/test_plume.py:26; T is not wired in PEM v0)
    _check_ranges(cathode_inputs(20_000, seed=4), v)
    _check_ranges(plume_inputs(20_000, seed=5), v, {'I_B0': ('uniform', 2.0, 8.0), 'T': ('uniform', 0.02, 0.12)})


def test_bench_inputs_are_the_yaml():
    """bench.synth_inputs fills a batch on the device; its arithmetic is exercised here on a CPU stand-in batch."""
    import torch
    import bench
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS

    class Batch:
        n, device = 20_000, torch.device('cpu')

        def __init__(self):
            self.rows = torch.empty((15, self.n), dtype=torch.float64)

        def load_soa(self, x, first):
            self.rows[:, first:first + x.shape[1]] = x
    b = Batch()
    bench.synth_inputs(b, seed=2, rank=0)
    _check_ranges({k: b.rows[i].numpy() for i, k in enumerate(COUPLED_INPUTS)}, _variables())
