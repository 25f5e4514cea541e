"""Parity of the HIP path (through the C ABI of libpem_hip.so) with the reference.

Two checkers: the golden vectors the reference itself produced (tests/golden), and the CPU oracle
(oracle/pem_oracle.c, pinned to those vectors by tests/test_oracle_golden.py) on seeded inputs.

Tolerance (BASELINE.json north_star: "within 1e-10 rel for the analytic cathode/plume models"):
  RTOL = 1e-10 relative on every finite output, NaN/inf patterns and invalid flags identical.
  div_angle = arccos(cos_div) is compared with conftest.div_err (relative 1e-10 unless the difference is
  within 8 ulp of cos_div, where arccos is ill-conditioned).
Observed on MI355X: see DESIGN.md "Parity" (cathode <= 2e-16, plume <= ~2e-12).
"""
import numpy as np
import pytest

from _inputs import cathode_inputs, coupled_inputs, plume_inputs
from conftest import div_err, load_golden, rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-10
PLUME_KEYS = ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex', 'I_B0')


@pytest.fixture(scope='module')
def pem():
    import hallthrusterpem_amd as pkg
    from hallthrusterpem_amd import _lib
    _lib.load()
    _lib.require_device()          # fail loudly: the GPU tier must never pass on a fallback
    return pkg


@pytest.fixture(scope='module')
def oc():
    from oracle import oracle_ctypes
    return oracle_ctypes


def _plume_in(g, prefix='in_'):
    d = {k: g[prefix + k] for k in PLUME_KEYS}
    if prefix + 'T' in g:
        d['T'] = g[prefix + 'T']
    return d


# ---------------------------------------------------------------------------------------------- golden vectors
def test_cathode_golden(pem):
    from hallthrusterpem_amd.models import cathode_coupling
    g = load_golden('cathode_random')
    pem.constants.set_torr_2_pa(float(g['TORR_2_PA']))
    out = cathode_coupling({k: g['in_' + k] for k in ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')})
    assert rel_err(out['V_cc'], g['out_V_cc']) <= RTOL
    assert np.all(out['V_cc'] >= 0) and np.all(out['V_cc'] <= 100)                  # tests/test_cathode.py:24
    for name in ('cathode_edges', 'cathode_wild'):
        g = load_golden(name)
        out = cathode_coupling({k: g['in_' + k] for k in ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')})
        assert rel_err(out['V_cc'], g['out_V_cc']) <= RTOL
    g = load_golden('cathode_edges')
    s = cathode_coupling({'P_b': 10e-6, 'V_a': 300, 'T_e': 3, 'V_vac': 30, 'Pstar': 20e-6, 'P_T': 50e-6})
    assert s['V_cc'].shape == (1,) and rel_err(s['V_cc'], g['scalar_out_V_cc']) <= RTOL   # test_cathode.py:14-15
    sw = cathode_coupling({'P_b': g['sweep_in_P_b'], 'V_a': 300, 'T_e': 1.33, 'V_vac': 31.6, 'Pstar': 24.6e-6,
                           'P_T': 10.2e-6})                                                 # test_cathode.py:27-31
    assert rel_err(sw['V_cc'], g['sweep_out_V_cc']) <= RTOL
    with pytest.raises(KeyError):
        cathode_coupling({'P_b': 1e-5})


@pytest.mark.parametrize('name', ['plume_random_r1', 'plume_priors_r1', 'plume_alpha_sweep', 'plume_random_r5',
                                  'plume_edges', 'plume_edges_r3', 'plume_cancel', 'plume_cancel_r3', 'plume_wild', 'plume_pressure_sweep'])
def test_plume_golden(pem, name):
    from hallthrusterpem_amd.models import current_density
    g = load_golden(name)
    pem.constants.set_torr_2_pa(float(g['TORR_2_PA']))
    radii = g['radii'] if g['radii'].size > 1 else float(g['radii'][0])
    out = current_density(_plume_in(g), sweep_radius=radii)
    assert out['j_ion'].shape == g['out_j_ion'].shape and out['div_angle'].shape == g['out_div_angle'].shape
    assert rel_err(out['j_ion'], g['out_j_ion']) <= RTOL
    assert div_err(out['div_angle'], g['out_div_angle']) <= RTOL
    if 'out_T_c' in g:
        assert rel_err(out['T_c'], g['out_T_c']) <= RTOL
    else:
        assert 'T_c' not in out
    assert out['j_ion_coords'].shape == tuple(g['out_coords_shape'])
    assert np.array_equal(out['j_ion_coords'].flat[0], g['out_coords0'])
    if name == 'plume_random_r5':                                                    # tests/test_plume.py:35,43-44
        assert out['j_ion'].shape == (96, 91, 5) and out['j_ion'].min() >= 0 and out['j_ion'].max() <= 5e3
    if name == 'plume_pressure_sweep':                                               # tests/test_plume.py:84-98
        from scipy.integrate import simpson
        theta = np.linspace(0, np.pi / 2, 91)
        cur = 2 * np.pi * simpson(out['j_ion'] * np.sin(theta), x=theta, axis=-1)
        assert np.sqrt(np.sum((cur - cur.mean()) ** 2) / np.sum(cur ** 2)) < 1e-4


def test_plume_fuzz_golden(pem):
    """The reference's outputs on 1560 fuzzed samples (tests/golden/plume_fuzz.npz), incl. the deep-underflow regimes:
    exp() flushed to zero in a narrow beam's tail, denormal and infinite amplitudes."""
    from conftest import wild_plume_errors
    from hallthrusterpem_amd.models import current_density
    g = load_golden('plume_fuzz')
    k = float(g['TORR_2_PA'])
    pem.constants.set_torr_2_pa(k)
    inputs = {q[3:]: g[q] for q in g if q.startswith('in_')}
    out = current_density(inputs, sweep_radius=1.0)
    want = {q: g['out_' + q] for q in ('j_ion', 'div_angle', 'T_c')}
    err = wild_plume_errors(inputs, out, want, k)
    assert err['compared'] > 500 and err['j_ion'] <= RTOL and err['div_angle'] <= RTOL and err['T_c'] <= RTOL


def test_plume_shapes_golden(pem):
    from hallthrusterpem_amd.models import current_density
    g = load_golden('plume_shapes')
    pem.constants.set_torr_2_pa(float(g['TORR_2_PA']))
    out = current_density({k: float(g['scalar_in_' + k]) for k in PLUME_KEYS})      # all scalars -> leading axis 1
    assert out['j_ion'].shape == (1, 91) and out['div_angle'].shape == (1,) and out['j_ion_coords'].shape == (1,)
    assert rel_err(out['j_ion'], g['scalar_out_j_ion']) <= RTOL
    out = current_density({k: g['nd_in_' + k] for k in PLUME_KEYS}, sweep_radius=g['nd_radii'])
    assert out['j_ion'].shape == (3, 4, 91, 2) and out['div_angle'].shape == (3, 4, 2)
    assert out['j_ion_coords'].shape == (3, 4)
    assert rel_err(out['j_ion'], g['nd_out_j_ion']) <= RTOL
    assert div_err(out['div_angle'], g['nd_out_div_angle']) <= RTOL
    with pytest.raises(KeyError):
        current_density({'P_b': 1e-5})


# ---------------------------------------------------------------------------------------------- against the oracle
def test_cathode_config1_lhs(pem, oc):
    """BASELINE.json configs[0]: 1e4 Latin-hypercube samples of cathode_coupling."""
    from hallthrusterpem_amd.models import cathode_coupling
    x = cathode_inputs(10_000, seed=0, lhs=True)
    k = pem.constants.TORR_2_PA
    got = cathode_coupling(x)['V_cc']
    want = oc.cathode(x['P_b'], x['V_a'], x['T_e'], x['V_vac'], x['Pstar'], x['P_T'], k)
    assert rel_err(got, want) <= RTOL


@pytest.mark.parametrize('lanes', [2, 4, 8])
@pytest.mark.parametrize('n', [1, 63, 65, 1000, 4099])
def test_plume_vs_oracle_every_variant_ragged(pem, oc, lanes, n):
    from hallthrusterpem_amd import _lib
    from hallthrusterpem_amd.models import current_density
    lib = _lib.load()
    assert lib.pem_set_lanes_per_sample(lanes) == lanes
    try:
        x = plume_inputs(n, seed=100 + n, priors=(n % 2 == 0))
        out = current_density(x)
        ref = oc.plume(*[x[k] for k in PLUME_KEYS], pem.constants.TORR_2_PA, T=x['T'])
        assert rel_err(out['j_ion'], ref['j_ion'][:, :, 0]) <= RTOL
        assert div_err(out['div_angle'], ref['div_angle'][:, 0]) <= RTOL
        assert rel_err(out['T_c'], ref['T_c'][:, 0]) <= RTOL
    finally:
        lib.pem_set_lanes_per_sample(0)


def test_plume_general_radii_vs_oracle(pem, oc):
    from hallthrusterpem_amd.models import current_density
    x = plume_inputs(777, seed=7, priors=False)
    radii = np.random.default_rng(8).random(25) * 0.2 + 1                            # tests/test_plume.py:31
    out = current_density(x, sweep_radius=radii)
    ref = oc.plume(*[x[k] for k in PLUME_KEYS], pem.constants.TORR_2_PA, T=x['T'], radii=radii)
    assert out['j_ion'].shape == (777, 91, 25)
    assert rel_err(out['j_ion'], ref['j_ion']) <= RTOL
    assert div_err(out['div_angle'], ref['div_angle']) <= RTOL
    assert rel_err(out['T_c'], ref['T_c']) <= RTOL


def test_thruster_stage_vs_oracle(pem, oc):
    from hallthrusterpem_amd.models import thruster_analytic
    x = coupled_inputs(5000, seed=3)
    vcc = np.random.default_rng(4).uniform(0, 60, 5000)
    got = thruster_analytic({'V_a': x['V_a'], 'V_cc': vcc, 'mdot_a': x['mdot_a'], 'a_1': x['a_1']})
    want = oc.thruster(x['V_a'], vcc, x['mdot_a'], x['a_1'])
    for k in want:
        assert rel_err(got[k], want[k]) <= 1e-15, k        # same IEEE operations, no transcendental but sqrt


@pytest.mark.parametrize('n', [1, 64, 100_000, 1_250_000])      # the last one is BASELINE configs[2]'s per-GPU shard
def test_coupled_vs_oracle(pem, oc, n):
    oc.set_threads(16)
    from hallthrusterpem_amd.models import pem_v0_coupled
    x = coupled_inputs(n, seed=2)
    got = pem_v0_coupled(x)
    want = oc.coupled(x, pem.constants.TORR_2_PA)
    assert rel_err(got['V_cc'], want['V_cc']) <= RTOL
    assert rel_err(got['I_B0'], want['I_B0']) <= 1e-15 and rel_err(got['T'], want['T']) <= RTOL
    assert rel_err(got['j_ion'], want['j_ion']) <= RTOL
    assert div_err(got['div_angle'], want['div_angle']) <= RTOL
    assert rel_err(got['T_c'], want['T_c']) <= RTOL
    assert np.array_equal(got['invalid'], want['invalid'])
    # reduced-QoI mode returns the same scalars without ever writing the profile
    red = pem_v0_coupled(x, profile=False)
    assert 'j_ion' not in red
    for k in ('V_cc', 'I_B0', 'T'):
        assert np.array_equal(red[k], got[k], equal_nan=True), k
    # ... the divergence integrals come from the Simpson-functional tables there (csrc: simpson_functionals), not from
    # the 91-term sums: held to the oracle like the full mode, and to the full mode within rounding
    assert div_err(red['div_angle'], want['div_angle']) <= RTOL and rel_err(red['T_c'], want['T_c']) <= RTOL
    # (arccos amplifies the ~1e-15 difference of cos_div by 1 / div_angle^2: narrow beams reach a few 1e-13)
    assert div_err(red['div_angle'], got['div_angle']) <= 1e-11 and rel_err(red['T_c'], got['T_c']) <= 1e-12
    assert np.array_equal(red['invalid'], got['invalid'])


def test_empty_batch(pem):
    from hallthrusterpem_amd.models import cathode_coupling, current_density
    e = np.empty(0)
    assert cathode_coupling({k: e for k in ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')})['V_cc'].shape == (0,)
    out = current_density({k: e for k in PLUME_KEYS})
    assert out['j_ion'].shape == (0, 91) and out['div_angle'].shape == (0,)


# ---------------------------------------------------------------------------------------------- device-resident path
def test_device_tensors_match_host_path(pem):
    import torch
    from hallthrusterpem_amd.models import cathode_coupling, current_density, pem_v0_coupled
    x = coupled_inputs(30_001, seed=11)
    xd = {k: torch.from_numpy(v).cuda() for k, v in x.items()}
    h = pem_v0_coupled(x)
    d = pem_v0_coupled(xd)
    torch.cuda.synchronize()
    for k in ('V_cc', 'I_B0', 'T', 'j_ion', 'div_angle', 'T_c', 'invalid'):
        assert d[k].is_cuda
        assert np.array_equal(d[k].cpu().numpy(), h[k], equal_nan=True), k           # same kernel: bit-identical
    c = cathode_coupling({k: xd[k] for k in ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')})['V_cc']
    assert np.array_equal(c.cpu().numpy(), h['V_cc'])
    p = plume_inputs(5000, seed=12)
    pd = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
    a, b = current_density(p), current_density(pd)
    assert np.array_equal(b['j_ion'].cpu().numpy(), a['j_ion'], equal_nan=True)
    assert np.array_equal(b['T_c'].cpu().numpy(), a['T_c'], equal_nan=True)


@pytest.mark.parametrize('n', [1, 63, 64, 65, 4096 + 37, 1_250_000])
def test_tile_interleaved_inputs_are_the_same_evaluation(pem, oc, n):
    """pem_coupled_tiled_f64_dev reads the 15 inputs from [tiles][15][64] blocks instead of 15 arrays: bit-identical to
    pem_coupled_f64_dev (profile and reduced-QoI mode, whole batch and range launches), and held to the oracle itself."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    oc.set_threads(16)
    x = coupled_inputs(n, seed=31)
    for profile in (True, False):
        a = CoupledBatch(n, profile=profile, layout='soa')
        b = CoupledBatch(n, profile=profile, layout='tile')
        a.set_inputs(x)
        b.set_inputs(x)
        assert b.inputs.shape == ((n + 63) // 64, 15, 64)
        assert torch.equal(b.inputs_soa(), a.inputs)
        a.run()
        b.run()
        torch.cuda.synchronize()
        oa, ob = a.outputs(), b.outputs()
        for k in oa:
            assert np.array_equal(oa[k].cpu().numpy(), ob[k].cpu().numpy(), equal_nan=True), (k, profile)
        if n > 300:                    # range launches of the tiled batch: from a multiple of 64 on
            c = CoupledBatch(n, profile=profile, layout='tile')
            c.set_inputs(x)
            cut = 128 * (n // 300)
            c.run(first=0, count=cut)
            c.run(first=cut)
            torch.cuda.synchronize()
            for k, v in c.outputs().items():
                assert np.array_equal(v.cpu().numpy(), oa[k].cpu().numpy(), equal_nan=True), (k, 'range')
            with pytest.raises(ValueError):
                c.run(first=2, count=10)
    want = oc.coupled(x, pem.constants.TORR_2_PA)
    b = CoupledBatch(n, layout='tile')
    b.load_soa(torch.stack([torch.from_numpy(np.ascontiguousarray(x[k])) for k in COUPLED_INPUTS]).cuda())
    b.run()
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in b.outputs().items()}
    for k in ('V_cc', 'j_ion', 'T_c'):
        assert rel_err(got[k], want[k]) <= RTOL, k
    assert div_err(got['div_angle'], want['div_angle']) <= RTOL and np.array_equal(got['invalid'], want['invalid'])


def test_mixed_profile_ranges_need_a_multiple_of_four(pem):
    """364-byte fp32 profile rows: a range launch must start 16-byte aligned, i.e. at a multiple of 4 samples -- an even
    `first` that is not one used to reach the library and come back as a generic INVALID_ARG."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    x = coupled_inputs(1000, seed=5)
    whole = CoupledBatch(1000, mixed=True)
    whole.set_inputs(x)
    whole.run()
    parts = CoupledBatch(1000, mixed=True)
    parts.set_inputs(x)
    with pytest.raises(ValueError, match='multiple of 4'):
        parts.run(first=2, count=100)
    with pytest.raises(ValueError, match='multiple of 2'):
        CoupledBatch(1000).run(first=3, count=100)
    parts.run(first=0, count=500)
    parts.run(first=500)
    torch.cuda.synchronize()
    assert torch.equal(whole.j_ion, parts.j_ion) and torch.equal(whole.qoi, parts.qoi)


# ---------------------------------------------------------------------------------------------- full-size properties
def test_config2_full_size_properties(pem, oc):
    """BASELINE.json configs[1]: 1e6 MC samples of the plume model, R = 1, fp64.  (a) ALL 1e6 samples against the
    oracle (OpenMP, a fraction of a second on the box's CPU share), (b) shard invariance (samples are independent:
    evaluating two halves equals evaluating the whole, bit for bit), (c) determinism, (d) the reference's own
    invariant, total current = I_B0 (tests/test_plume.py:91-98), (e) range, (f) linearity in I_B0."""
    import torch
    from hallthrusterpem_amd.models import current_density
    n = 1_000_000
    x = plume_inputs(n, seed=1, priors=True)
    xd = {k: torch.from_numpy(v).cuda() for k, v in x.items()}
    out = current_density(xd)
    j = out['j_ion']
    oc.set_threads(16)
    ref = oc.plume(*[x[k] for k in PLUME_KEYS], pem.constants.TORR_2_PA, T=x['T'])
    assert rel_err(j.cpu().numpy(), ref['j_ion'][:, :, 0]) <= RTOL
    assert div_err(out['div_angle'].cpu().numpy(), ref['div_angle'][:, 0]) <= RTOL
    assert rel_err(out['T_c'].cpu().numpy(), ref['T_c'][:, 0]) <= RTOL
    del ref
    # j_ion is linear in the beam current: scaling I_B0 by a power of two scales every output entry exactly
    x4 = dict(xd)
    x4['I_B0'] = xd['I_B0'] * 4.0
    j4 = current_density(x4)
    assert torch.equal(j4['j_ion'], 4.0 * j) and torch.equal(j4['div_angle'], out['div_angle'])
    half = n // 2 + 17
    lo = current_density({k: v[:half] for k, v in xd.items()})
    hi = current_density({k: v[half:] for k, v in xd.items()})
    assert torch.equal(torch.cat([lo['j_ion'], hi['j_ion']]), j)
    assert torch.equal(torch.cat([lo['div_angle'], hi['div_angle']]), out['div_angle'])
    again = current_density(xd)
    assert torch.equal(again['j_ion'], j) and torch.equal(again['T_c'], out['T_c'])
    assert bool((j >= 0).all()) and bool((j <= 5e3).all())
    # total current: 2 pi R^2 Int j sin(theta) dtheta = I_B0 wherever the 1-degree Simpson rule resolves the beam
    theta = torch.linspace(0, np.pi / 2, 91, dtype=torch.float64, device='cuda')
    w = torch.full((91,), 2.0, dtype=torch.float64, device='cuda')
    w[1::2] = 4.0
    w[0] = w[-1] = 1.0
    w *= (np.pi / 2 / 90) / 3
    cur = 2 * np.pi * (j * torch.sin(theta) * w).sum(-1)
    a1 = torch.clamp(xd['c2'] * xd['P_b'] * pem.constants.TORR_2_PA + xd['c3'], max=np.pi / 2)
    resolved = a1 > 0.1
    rel = ((cur - xd['I_B0']).abs() / xd['I_B0'])[resolved]
    assert float(rel.max()) < 1e-4


# ---------------------------------------------------------------------------------------------- mixed precision
def test_mixed_precision_profile_tolerance(pem, oc):
    """BASELINE.json configs[4]: fp64 -> fp32 mixed run compared with fp64 on identical inputs.
    The mixed mode keeps the fp64 arithmetic and rounds the profile once to fp32, so the scalars are bit-identical
    and every profile entry is within half an fp32 ulp (6e-8 relative) of the fp64 result."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    n = 200_003
    x = coupled_inputs(n, seed=21)
    full, mixed = CoupledBatch(n), CoupledBatch(n, mixed=True)
    full.set_inputs(x)
    mixed.set_inputs(x)
    full.run()
    mixed.run()
    torch.cuda.synchronize()
    assert mixed.j_ion.dtype == torch.float32 and mixed.bytes_per_eval == 508
    assert torch.equal(full.qoi, mixed.qoi) and torch.equal(full.invalid, mixed.invalid)
    assert torch.equal(full.j_ion.float(), mixed.j_ion)                 # exactly the correctly rounded fp64 value
    rel = ((mixed.j_ion.double() - full.j_ion) / full.j_ion).abs()
    assert float(rel.max()) <= 2.0 ** -24 and float(rel.flatten().kthvalue(int(0.999 * rel.numel())).values) <= 2.0 ** -24


# ---------------------------------------------------------------------------------------------- thruster profile + filters
def test_thruster_profile_and_filters_on_device(pem, oc):
    import torch
    from hallthrusterpem_amd.models.thruster import check_thruster_outputs, thruster_analytic
    x = coupled_inputs(3000, seed=31)
    vcc = np.random.default_rng(32).uniform(0, 60, 3000)
    ins = {'V_a': x['V_a'], 'V_cc': vcc, 'mdot_a': x['mdot_a'], 'a_1': x['a_1']}
    host = thruster_analytic(ins, num_cells=102)                                 # tests/test_thruster.py:185 sizes
    dev = thruster_analytic({k: torch.from_numpy(v).cuda() for k, v in ins.items()}, num_cells=102)
    z, u = oc.thruster_uion(host['v_exh'], 0.0, 0.08, 102)
    assert host['u_ion'].shape == (3000, 102) and rel_err(host['u_ion'], u) <= 1e-14
    assert np.allclose(host['u_ion_coords'], z, rtol=4e-16, atol=0)
    assert np.array_equal(dev['u_ion'].cpu().numpy(), host['u_ion'])
    # filters: device result == numpy result == what the reference would raise for
    T = host['T'].copy()
    IB = host['I_B0'].copy()
    T[5], IB[7] = -1e-3, -0.5
    uu = host['u_ion'].copy()
    uu[11, :20] = 1e9                                   # peak at z < threshold -> shock-like
    uu[12, 40] = np.nan                                 # np.argmax lands on the NaN (z = 0.0317 < 0.04)
    uu[13, 90] = np.nan                                 # NaN beyond the threshold
    outs = {'T': T, 'I_B0': IB, 'u_ion': uu, 'u_ion_coords': host['u_ion_coords']}
    bad_h = check_thruster_outputs(outs, shock_threshold=0.04)
    bad_d = check_thruster_outputs({k: torch.from_numpy(v).cuda() for k, v in outs.items()}, shock_threshold=0.04)
    assert bad_d.is_cuda and np.array_equal(bad_d.cpu().numpy(), bad_h)
    assert bad_h[[5, 7, 11, 12]].all() and not bad_h[13] and bad_h.sum() == 4
    no_shock = check_thruster_outputs({'T': torch.from_numpy(T).cuda(), 'I_B0': torch.from_numpy(IB).cuda()})
    assert int(no_shock.sum()) == 2


# ---------------------------------------------------------------------------------------------- C-ABI error behaviour
def test_c_abi_argument_errors_and_fallback_paths(pem, oc):
    """API misuse returns status codes with a message (never a crash); a misaligned j_ion falls back to the
    general kernel and still matches."""
    import ctypes as C
    import torch
    from hallthrusterpem_amd import _lib
    lib = _lib.load()
    x = plume_inputs(1000, seed=41)
    d = {k: torch.from_numpy(v).cuda() for k, v in x.items()}
    ptr = lambda t: C.c_void_p(t.data_ptr())                                          # noqa: E731
    ins = [ptr(d[k]) for k in PLUME_KEYS]
    j = torch.empty(1000 * 91 + 1, dtype=torch.float64, device='cuda')
    div = torch.empty(1000, dtype=torch.float64, device='cuda')
    tc = torch.empty(1000, dtype=torch.float64, device='cuda')
    rad = np.array([1.0])
    rp = C.c_void_p(rad.ctypes.data)
    k = pem.constants.TORR_2_PA
    # T without T_c
    rc = lib.pem_plume_f64_dev(1000, 1, rp, k, *ins, ptr(d['T']), ptr(j), ptr(div), None, None, None)
    assert rc == _lib.PEM_ERR_INVALID_ARG and b'T and T_c' in lib.pem_last_error()
    # no radii / NULL arrays
    assert lib.pem_plume_f64_dev(1000, 0, rp, k, *ins, None, ptr(j), ptr(div), None, None, None) == _lib.PEM_ERR_INVALID_ARG
    assert lib.pem_plume_f64_dev(1000, 1, rp, k, None, *ins[1:], None, ptr(j), ptr(div), None, None, None) == _lib.PEM_ERR_INVALID_ARG
    assert lib.pem_cathode_f64_dev(10, None, None, None, None, None, None, k, None, None) == _lib.PEM_ERR_INVALID_ARG
    with pytest.raises(_lib.PemHipError):
        _lib.check(_lib.PEM_ERR_INVALID_ARG)
    # misaligned profile pointer (8 bytes off a 16-byte boundary): general kernel, same numbers
    j_mis = j[1:]
    assert j_mis.data_ptr() % 16 == 8
    rc = lib.pem_plume_f64_dev(1000, 1, rp, k, *ins, ptr(d['T']), ptr(j_mis), ptr(div), ptr(tc), None, None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = oc.plume(*[x[kk] for kk in PLUME_KEYS], k, T=x['T'])
    assert rel_err(j_mis.cpu().numpy().reshape(1000, 91), ref['j_ion'][:, :, 0]) <= RTOL
    assert div_err(div.cpu().numpy(), ref['div_angle'][:, 0]) <= RTOL
    # coupled insists on alignment (its profile stores are 16-byte pieces)
    c = coupled_inputs(100, seed=42)
    cd = [ptr(torch.from_numpy(c[kk]).cuda()) for kk in ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T', 'mdot_a', 'a_1',
                                                          'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')]
    o = [torch.empty(100, dtype=torch.float64, device='cuda') for _ in range(5)]
    rc = lib.pem_coupled_f64_dev(100, k, 1.0, *cd, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(j_mis), ptr(o[3]), ptr(o[4]), None, None)
    assert rc == _lib.PEM_ERR_INVALID_ARG and b'aligned' in lib.pem_last_error()
    # lanes knob only accepts supported values
    assert lib.pem_set_lanes_per_sample(3) == 4 and lib.pem_set_lanes_per_sample(2) == 2 and lib.pem_set_lanes_per_sample(0) == 4


def test_large_batch_index_arithmetic(pem):
    """2.5e7 samples in reduced-QoI mode (3.6 GB of inputs): 64-bit indexing, persistent loop over 390k tiles;
    first / last / strided samples against a small evaluation of the same inputs."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.sampling import Design
    n = 25_000_007
    big = CoupledBatch(n, profile=False)
    Design(seed=77).fill(big.inputs)
    big.run()
    idx = torch.tensor([0, 1, 63, 64, 12_345_678, n - 65, n - 64, n - 2, n - 1], device='cuda')
    small = CoupledBatch(idx.numel(), profile=False)
    small.inputs.copy_(big.inputs[:, idx])
    small.run()
    torch.cuda.synchronize()
    assert torch.equal(big.qoi[:, idx], small.qoi)
    assert bool(torch.isfinite(big.qoi).all())


def test_host_entry_points_from_a_thread_pool(pem, oc):
    """SURVEY section 8b, threading: the reference evaluates models on Thread pools (gen_data.py:448-456); ctypes
    releases the GIL, so the host-pointer entry points really run concurrently.  Every call must return exactly what
    it returns alone, on the process's device (new threads start on HIP device 0; pem_init pins the choice)."""
    from concurrent.futures import ThreadPoolExecutor
    import hallthrusterpem_amd
    from hallthrusterpem_amd.models import cathode_coupling, current_density
    hallthrusterpem_amd.set_device(0)
    jobs = []
    for seed in range(24):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(1, 5000))
        if seed % 2:
            jobs.append((cathode_coupling, {'P_b': 10.0 ** rng.uniform(-8, -4, n), 'V_a': rng.uniform(200, 400, n),
                                            'T_e': rng.uniform(1, 5, n), 'V_vac': rng.uniform(0, 60, n),
                                            'Pstar': rng.uniform(1e-5, 1e-4, n), 'P_T': rng.uniform(1e-5, 1e-4, n)}))
        else:
            jobs.append((current_density, {'P_b': 10.0 ** rng.uniform(-8, -4, n), 'c0': rng.uniform(0, 1, n),
                                           'c1': rng.uniform(0.1, 0.9, n), 'c2': rng.uniform(-15, 15, n),
                                           'c3': rng.uniform(0.2, 1.57, n), 'c4': 10.0 ** rng.uniform(18, 22, n),
                                           'c5': 10.0 ** rng.uniform(14, 18, n), 'sigma_cex': rng.uniform(51e-20, 58e-20, n),
                                           'I_B0': rng.uniform(2, 8, n), 'T': rng.uniform(0.05, 0.1, n)}))
    alone = [f(x) for f, x in jobs]
    with ThreadPoolExecutor(max_workers=8) as pool:
        for _ in range(3):
            together = list(pool.map(lambda job: job[0](job[1]), jobs))
            for a, b in zip(alone, together):
                for k in a:
                    if k != 'j_ion_coords':
                        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_reduced_mode_table_and_loop_paths_agree_sample_by_sample(pem, oc):
    """Reduced-QoI mode: "plain" samples (amplitudes >= 0, j_cex > 0, beams wider than QA_MIN) take the divergence
    integrals from tables, the others from the 91-term loop -- per sample, so a result never depends on which other
    samples share its tile.  Mixed tiles, narrow beams on both sides of QA_MIN, invalid and NaN samples vs the oracle."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    n = 64 * 40 + 17
    x = coupled_inputs(n, seed=9)
    rng = np.random.default_rng(9)
    row = {k: i for i, k in enumerate(COUPLED_INPUTS)}
    x['c2'] = np.where(rng.random(n) < 0.5, 0.0, x['c2'])
    narrow = rng.random(n) < 0.3
    x['c3'] = np.where(narrow, rng.uniform(0.005, 0.06, n), x['c3'])          # alpha1 around QA_MIN = 0.03
    x['c2'] = np.where(narrow, 0.0, x['c2'])
    x['c0'] = np.where(rng.random(n) < 0.1, rng.uniform(-0.5, 1.5, n), x['c0'])   # negative amplitudes: loop path
    x['c3'][5], x['c2'][5] = -0.3, 0.0                                        # alpha1 <= 0: invalid
    x['c1'][70] = np.nan
    x['c3'][130:194] = 0.7                                                    # one whole tile of plain samples
    x['c2'][130:194], x['c0'][130:194] = 0.0, 0.4
    want = oc.coupled(x, pem.constants.TORR_2_PA)
    b = CoupledBatch(n, profile=False)
    b.set_inputs(x)
    b.run()
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in b.outputs().items()}
    assert np.array_equal(got['invalid'], want['invalid']) and got['invalid'][5] and got['invalid'].sum() > 20
    assert np.array_equal(np.isnan(got['div_angle']), np.isnan(want['div_angle'])) and np.isnan(got['div_angle'][70])
    assert div_err(got['div_angle'], want['div_angle']) <= RTOL and rel_err(got['T_c'], want['T_c']) <= RTOL
    # the same samples in another order (other tile mates): bit-identical results
    perm = rng.permutation(n)
    b2 = CoupledBatch(n, profile=False)
    b2.inputs.copy_(b.inputs[:, torch.from_numpy(perm).cuda()])
    b2.run()
    torch.cuda.synchronize()
    assert np.array_equal(b2.qoi.cpu().numpy(), b.qoi.cpu().numpy()[:, perm], equal_nan=True)


@pytest.mark.parametrize('R', [2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 16, 17, 25, 31, 32, 33, 47, 64, 65, 70, 256, 257])
def test_sweep_radius_counts_across_the_kernel_switches(pem, oc, R):
    """sweep_radius arrays: the recurrence kernel for 2..8 radii (register blocks of 2 / 4 / 8 radii, 8- and 16-byte stores),
    the wave-per-sample kernel for 9..12 and 65..256 (whose radius loop runs in chunks of 64 lanes), the staged kernel for
    13..64 (four / three / two / one samples in flight per wave: switches at 16, 21 and 32; odd counts start every second sample's runs on an odd double), the lane-per-sample kernel above -- every count at a switch against the oracle, invalid samples included, also
    with a ragged tile (n = 333)."""
    from hallthrusterpem_amd.models import current_density
    n = 333
    x = plume_inputs(n, seed=40 + R, priors=False)            # tests/test_plume.py ranges
    x['c3'][:7], x['c2'][:7] = -0.1, 0.0                      # alpha1 <= 0: invalid samples
    x['c0'][7:11] = 1.3                                       # negative beam amplitude: j_ion <= 0 somewhere
    radii = np.linspace(0.4, 2.0, R)
    out = current_density(x, sweep_radius=radii)
    ref = oc.plume(*[x[k] for k in PLUME_KEYS], pem.constants.TORR_2_PA, T=x['T'], radii=radii)
    assert out['j_ion'].shape == (n, 91, R) and out['div_angle'].shape == (n, R) and out['T_c'].shape == (n, R)
    assert rel_err(out['j_ion'], ref['j_ion']) <= RTOL
    assert div_err(out['div_angle'], ref['div_angle']) <= RTOL and rel_err(out['T_c'], ref['T_c']) <= RTOL
    assert ref['invalid'].any() and np.array_equal(np.all(out['j_ion'].reshape(n, -1) == 1e-20, axis=1), ref['invalid'])


@pytest.mark.parametrize('R,ts,rmid_min', [(17, 63, None), (17, 32, None), (25, 64, None), (25, 32, None), (33, 64, None), (33, 63, None),
                                           (11, 60, 11), (12, 64, 11), (13, 32, 11), (16, 64, 11)])
def test_staged_radii_kernel_at_production_tile_sizes(pem, oc, monkeypatch, R, ts, rmid_min):
    """plume_rmid_kernel (13..64 sweep radii by default) as large batches run it -- the host picks tiles of 8 samples for every
    n < 131072, so the cases above it never saw tiles of 16 / 32 / 64 (63 at three samples per wave), several groups per tile, or
    shuffles from lanes >= 8 -- and the instantiations for four and five samples per wave, reachable through
    PEM_RMID_MIN only (11..16 radii; six and seven were dropped).  Against the oracle, invalid samples and a ragged tile included."""
    from hallthrusterpem_amd.models import current_density
    monkeypatch.setenv('PEM_RMID_TS', str(ts))
    if rmid_min is not None:
        monkeypatch.setenv('PEM_RMID_MIN', str(rmid_min))
    n = 1000 + R
    x = plume_inputs(n, seed=90 + R + ts, priors=False)
    x['c3'][:7], x['c2'][:7] = -0.1, 0.0
    x['c0'][7:11] = 1.3
    x['c3'][-1], x['c2'][-1] = -0.2, 0.0                      # an invalid sample in the ragged last tile
    radii = np.linspace(0.4, 2.0, R)
    out = current_density(x, sweep_radius=radii)
    ref = oc.plume(*[x[k] for k in PLUME_KEYS], pem.constants.TORR_2_PA, T=x['T'], radii=radii)
    assert rel_err(out['j_ion'], ref['j_ion']) <= RTOL
    assert div_err(out['div_angle'], ref['div_angle']) <= RTOL and rel_err(out['T_c'], ref['T_c']) <= RTOL
    assert ref['invalid'].any() and np.array_equal(np.all(out['j_ion'].reshape(n, -1) == 1e-20, axis=1), ref['invalid'])
