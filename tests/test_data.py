"""`hallthrusterpem_amd.data.pem_to_xarray` against the indexing rules of src/hallmd/data.py:239-279 (unpinned: the reference has
no test of it and needs xarray + pem_core, both absent here): corrected thrust of the LAST radius, `j_ion` as (r, theta),
`u_ion` on its z grid, one entry per operating condition.  The fallback records (plain dicts) are what this image produces."""
import numpy as np
import pytest

from _inputs import coupled_inputs


def _leaf(field):
    return field['val'] if isinstance(field, dict) else field.val


def _check(entries, ops, out, radii, corrected=True):
    assert len(entries) == len(ops)
    R = np.atleast_1d(radii).size
    for i, e in enumerate(entries):
        assert e['operating_condition'] is ops[i]
        d = e['data']
        assert list(d) == ['discharge current', 'cathode coupling voltage', 'thrust', 'ion velocity', 'ion current density']
        assert [d[k]['unit'] for k in d] == ['A', 'V', 'N', 'm/s', 'A/m^2']
        for k, src in (('discharge current', 'I_d'), ('cathode coupling voltage', 'V_cc')):
            a = _leaf(d[k])
            assert a['dims'] == () and a['val'].shape == () and a['val'] == np.asarray(out[src])[i]
        want_T = np.atleast_1d(np.asarray(out['T_c'])[i])[-1] if corrected else np.asarray(out['T'])[i]
        assert _leaf(d['thrust'])['val'] == want_T and _leaf(d['thrust'])['val'].shape == ()
        u = _leaf(d['ion velocity'])
        assert u['dims'] == ('z',) and np.array_equal(u['val'], np.asarray(out['u_ion'])[i]) and u['coords']['z'].shape == u['val'].shape
        j = _leaf(d['ion current density'])
        assert j['dims'] == ('r', 'theta') and j['val'].shape == (R, 91)
        assert np.array_equal(j['coords']['r'], np.atleast_1d(radii)) and j['coords']['theta'].shape == (91,)
        full = np.asarray(out['j_ion']).reshape(len(ops), 91, R)
        assert np.array_equal(j['val'], full[i].T)


def test_shape_rules_on_synthetic_outputs():
    from hallthrusterpem_amd.data import pem_to_xarray
    rng = np.random.default_rng(3)
    n, nz = 4, 102
    theta = np.linspace(0, np.pi / 2, 91)
    coords = np.empty(n, dtype=object)
    for i in range(n):
        coords[i] = theta
    zc = np.empty(n, dtype=object)
    for i in range(n):
        zc[i] = np.linspace(0, 0.08, nz) + i        # per-sample coordinates, as amisc hands them over
    ops = [{'P_b': 1e-5 * (i + 1), 'V_a': 300.0, 'mdot_a': 5e-6} for i in range(n)]
    for R in (1, 3):
        radii = np.array([1.0]) if R == 1 else np.array([0.5, 1.0, 1.5])
        out = {'T_c': rng.random(n) if R == 1 else rng.random((n, R)), 'T': rng.random(n), 'I_d': rng.random(n), 'V_cc': rng.random(n),
               'u_ion': rng.random((n, nz)), 'u_ion_coords': zc, 'j_ion': rng.random((n, 91)) if R == 1 else rng.random((n, 91, R)),
               'j_ion_coords': coords}
        _check(pem_to_xarray(ops, out, radii), ops, out, radii)
        _check(pem_to_xarray(ops, out, radii, use_corrected_thrust=False), ops, out, radii, corrected=False)
        got = pem_to_xarray(ops, out, radii)
        assert np.array_equal(_leaf(got[2]['data']['ion velocity'])['coords']['z'], zc[2])      # sample 2's own grid
    with pytest.raises(ValueError):
        pem_to_xarray(ops, out, np.array([1.0, 2.0]))                                            # 3 radii in j_ion, 2 given
    with pytest.raises(KeyError):
        pem_to_xarray(ops, {k: v for k, v in out.items() if k != 'I_d'}, radii)


@pytest.mark.gpu
@pytest.mark.parametrize('radii', [1.0, [0.6, 1.0, 1.4]])
def test_outputs_of_the_device_path(radii):
    """cathode -> thruster (analytic test double, with the ion-velocity profile) -> plume on the GPU, device tensors straight
    into pem_to_xarray, one and three sweep radii."""
    import torch
    from hallthrusterpem_amd.data import pem_to_xarray
    from hallthrusterpem_amd.models import cathode_coupling, current_density, thruster_analytic
    n = 6
    x = {k: torch.from_numpy(v).cuda() for k, v in coupled_inputs(n, seed=8).items()}
    vcc = cathode_coupling({k: x[k] for k in ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')})['V_cc']
    th = thruster_analytic({'V_a': x['V_a'], 'V_cc': vcc, 'mdot_a': x['mdot_a'], 'a_1': x['a_1']}, num_cells=102)
    pl = current_density({**{k: x[k] for k in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')}, 'I_B0': th['I_B0'], 'T': th['T']},
                         sweep_radius=radii if np.ndim(radii) == 0 else np.array(radii))
    out = {'V_cc': vcc, 'I_d': th['I_d'], 'T': th['T'], 'u_ion': th['u_ion'], 'u_ion_coords': th['u_ion_coords'], **pl}
    ops = [{'P_b': float(x['P_b'][i]), 'V_a': float(x['V_a'][i]), 'mdot_a': float(x['mdot_a'][i])} for i in range(n)]
    host = {k: (v.cpu().numpy() if hasattr(v, 'cpu') else v) for k, v in out.items()}
    entries = pem_to_xarray(ops, out, np.atleast_1d(radii))
    _check(entries, ops, host, np.atleast_1d(np.asarray(radii, dtype=float)))
    # the corrected thrust is the one of the LAST radius
    assert _leaf(entries[0]['data']['thrust'])['val'] == np.atleast_1d(host['T_c'][0])[-1]
    assert np.allclose(_leaf(entries[0]['data']['ion current density'])['coords']['theta'], np.linspace(0, np.pi / 2, 91))
