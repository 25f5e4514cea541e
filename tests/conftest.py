import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(GOLDEN / f'{name}.npz') as z:
        return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def rel_err(got, want, floor=0.0):
    """max |got-want| / max(|want|, floor) over finite entries; NaN/inf patterns must match exactly."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), 'NaN pattern differs'
    fin = np.isfinite(want)
    assert np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]), 'inf pattern differs'
    if not fin.any():
        return 0.0
    den = np.maximum(np.abs(want[fin]), floor) if floor else np.abs(want[fin])
    den = np.where(den == 0, 1.0, den)
    return float(np.max(np.abs(got[fin] - want[fin]) / den))


def div_err(got, want, cos_ulps=8):
    """Error metric for div_angle = arccos(cos_div) (plume.py:127).

    arccos is ill-conditioned at cos_div -> 1 (beams narrower than the 1-degree grid): one ulp of
    cos_div moves the angle by 1.1e-16/sin(angle), and by sqrt(2.2e-16) at the pole itself.  An entry whose
    cosine is within `cos_ulps` ulps of the reference's counts as exact; otherwise its relative angle error
    is returned.
    """
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), 'NaN pattern differs'
    fin = np.isfinite(want)
    if not fin.any():
        return 0.0
    d = np.abs(got[fin] - want[fin])
    w = np.abs(want[fin])
    # |cos(got) - cos(want)| = 2 |sin((got + want) / 2) sin((got - want) / 2)|: exact also at the pole, where it is d^2 / 2
    in_cos_noise = 2.0 * np.abs(np.sin(0.5 * (got[fin] + want[fin])) * np.sin(0.5 * (got[fin] - want[fin]))) <= cos_ulps * 1.1102230246251565e-16
    rel = np.where(in_cos_noise, 0.0, d / np.where(w == 0, 1.0, w))
    return float(rel.max())


def wild_plume_errors(inputs, got, want, torr2pa, radius=1.0):
    """Compare plume results on inputs far outside the priors (tests/golden/plume_fuzz.npz) under the per-entry rules of
    tests/parity_rules.py: NaN / inf patterns and invalid rows identical; every finite entry within 1e-10 of its value
    plus TAU of the magnitudes of the terms it is summed from (taken from the oracle's decomposition of the sample).
    got / want: dicts with j_ion (n, 91), div_angle (n,), T_c (n,).  Returns the worst errors, scaled so that 1e-10 passes."""
    import parity_rules as pr
    from oracle import oracle_ctypes as oc
    x = {k: np.asarray(v, dtype=np.float64) for k, v in inputs.items()}
    with np.errstate(all='ignore'):
        terms = oc.plume_terms(*[x[q] for q in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex', 'I_B0')], torr2pa, radii=(radius,))
        bounds = pr.plume_bounds(terms, x['I_B0'])
    r = pr.j_ion_error(got['j_ion'], want['j_ion'], bounds)
    d = pr.divergence_error(got['div_angle'], want['div_angle'], got['T_c'], want['T_c'], bounds)
    return {'j_ion': r['err'], 'div_angle': d['err_div'], 'T_c': d['err_tc'], 'compared': int(bounds['comparable'].sum())}
