"""The reference's own unit tests for this path, restated against the drop-in package: same inputs (seeded instead
of unseeded `np.random.rand`), same assertions, same call signatures.
  tests/test_cathode.py:8-31     -> test_cathode_coupling
  tests/test_plume.py:17-44      -> test_random_samples
  tests/test_plume.py:64-98      -> test_pressure_sweep
  tests/test_thruster.py:70-114  -> test_sim_hallthruster_jl (the wrapper, run on the GPU test double) and
                                    test_sim_hallthruster_stage (the stage alone, batched)
"""
from typing import cast

import numpy as np
import pytest
from scipy.integrate import simpson

pytestmark = pytest.mark.gpu

J_MIN, J_MAX, N = 0, 5e3, 100


def test_cathode_coupling():
    from hallthrusterpem_amd.models.cathode import cathode_coupling
    VCC_LB, VCC_UB = 0, 100
    inputs = {'P_b': 10e-6, 'V_a': 300, 'T_e': 3, 'V_vac': 30, 'Pstar': 20e-6, 'P_T': 50e-6}
    outputs = cathode_coupling(inputs)
    assert outputs['V_cc'].shape == (1,)
    rng = np.random.default_rng(0)
    inputs_rand = {'P_b': 10 ** (rng.random(N) * 4 - 8), 'V_a': rng.random(N) * 200 + 200, 'T_e': rng.random(N) * 4 + 1,
                   'V_vac': rng.random(N) * 60, 'Pstar': rng.random(N) * 90e-6 + 10e-6, 'P_T': rng.random(N) * 90e-6 + 10e-6}
    outputs_rand = cathode_coupling(inputs_rand)
    assert np.all(outputs_rand['V_cc'] >= VCC_LB) and np.all(outputs_rand['V_cc'] <= VCC_UB)
    inputs_sweep = {'P_b': 10 ** (np.linspace(-6, -4, N)), 'V_a': 300, 'T_e': 1.33, 'V_vac': 31.6, 'Pstar': 24.6e-6,
                    'P_T': 10.2e-6}
    outputs_sweep = cathode_coupling(inputs_sweep)
    assert np.all(outputs_sweep['V_cc'] >= VCC_LB) and np.all(outputs_sweep['V_cc'] <= VCC_UB)


def test_random_samples():
    from hallthrusterpem_amd.models.plume import current_density
    rng = np.random.default_rng(1)
    inputs_rand = {'P_b': 10 ** (rng.random(N) * 4 - 8), 'c0': rng.random(N) * 0.8 + 0.1, 'c1': rng.random(N) * 0.8 + 0.1,
                   'c2': rng.random(N) * 30 - 15, 'c3': rng.random(N) + 0.1, 'c4': 10 ** (rng.random(N) * 4 + 18),
                   'c5': 10 ** (rng.random(N) * 4 + 14), 'sigma_cex': rng.random(N) * 7e-20 + 51e-20,
                   'I_B0': rng.random(N) * 6 + 2}
    r_p = rng.random(25) * 0.2 + 1
    outputs_rand = cast(dict, current_density(inputs_rand, sweep_radius=r_p))
    assert outputs_rand['j_ion'].shape == (N, 91, 25)
    assert np.min(outputs_rand['j_ion']) >= J_MIN and np.max(outputs_rand['j_ion']) <= J_MAX


def test_pressure_sweep():
    from hallthrusterpem_amd.models.plume import current_density
    pressure_sweep = 10 ** (np.linspace(-6, -4, N))
    inputs_sweep = {'P_b': pressure_sweep, 'c0': 0.1, 'c1': 0.7, 'c2': -8.0, 'c3': 0.2, 'c4': 1e20, 'c5': 1e16,
                    'sigma_cex': 55e-20, 'I_B0': 3}
    outputs_sweep = cast(dict, current_density(inputs_sweep, sweep_radius=1))
    assert np.min(outputs_sweep['j_ion']) >= J_MIN and np.max(outputs_sweep['j_ion']) <= J_MAX
    R = 1
    theta = np.linspace(0, np.pi / 2, outputs_sweep['j_ion'].shape[-1])
    current = np.empty(outputs_sweep['j_ion'].shape[0])
    for i in range(outputs_sweep['j_ion'].shape[0]):
        current[i] = 2 * np.pi * R ** 2 * simpson(outputs_sweep['j_ion'][i, :] * np.sin(theta), x=theta)
    err = np.sqrt(np.sum((current - np.mean(current)) ** 2) / np.sum(current ** 2))
    assert err < 1e-4


def test_sim_hallthruster_stage():
    """tests/test_thruster.py:70-114 runs `hallthruster_jl(..., julia_script=tests/sim_hallthruster.jl)` and checks
    that the PEM outputs are present; the same QoIs from the batched test double, with the ranges the real-solver test
    asserts (tests/test_thruster.py:185-189)."""
    from hallthrusterpem_amd.models.thruster import check_thruster_outputs, thruster_analytic
    num_cells = 100
    out = thruster_analytic({'V_a': 300.0, 'V_cc': 30.0, 'mdot_a': 5e-6, 'a_1': 0.00625}, num_cells=num_cells + 2)
    for key in ('I_B0', 'I_d', 'T', 'eta_c', 'eta_m', 'eta_v', 'eta_a', 'u_ion', 'u_ion_coords'):
        assert key in out
    assert out['u_ion'].shape[-1] == num_cells + 2 == len(out['u_ion_coords'])
    assert 0 < out['T'][0] < 0.2 and 0 < out['I_B0'][0] < 10 and 0 < out['I_d'][0] < 10
    assert not check_thruster_outputs({k: v[0] if k not in ('u_ion_coords',) else v for k, v in out.items()},
                                      shock_threshold=0.02)


def test_sim_hallthruster_jl(tmp_path):
    """tests/test_thruster.py:70-114 line for line: the reference runs `hallthruster_jl(..., julia_script=tests/
    sim_hallthruster.jl)`; here the same call (the script's arithmetic is the default backend, evaluated on the GPU) with a
    dict for the device instead of the downloaded SPT-100 file.  tests/test_hallthruster_jl.py holds the values to what
    the reference's own function returned."""
    import json
    from hallthrusterpem_amd.models import hallthruster_jl
    thruster_inputs = {'V_a': 250, 'V_cc': 25, 'mdot_a': 3.5e-6}
    config = {'anom_model': {'type': 'LogisticPressureShift', 'model': {'type': 'TwoZoneBohm', 'c1': 0.008, 'c2': 0.08}},
              'domain': [0, 0.08]}
    simulation = {'grid': {'type': 'EvenGrid', 'num_cells': 100}, 'duration': 1e-3, 'dt': 1e-9}
    postprocess = {'average_start_time': 0.5e-3}
    outputs = hallthruster_jl(thruster_inputs, config=config, simulation=simulation, thruster={'name': 'SPT-100'},
                              postprocess=postprocess, julia_script='sim_hallthruster.jl', output_path=tmp_path)
    outputs = cast(dict, outputs)
    for key in ['T', 'I_B0', 'I_d', 'u_ion', 'u_ion_coords']:
        assert key in outputs
    with open(tmp_path / outputs['output_path'], 'r') as fd:
        data = json.load(fd)
        for key in ['thrust', 'ion_current', 'discharge_current', 'mass_eff', 'voltage_eff', 'current_eff']:
            assert key in data['output']['average']
    assert data['output']['average']['thrust'] == outputs['T'] and len(data['output']['average']['z']) == len(outputs['u_ion'])
