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
    cos_div moves the angle by 1.1e-16/sin(angle).  An entry whose angle error is within `cos_ulps`
    ulps of cos_div counts as exact; otherwise its relative angle error is returned.
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
    in_cos_noise = d * np.maximum(np.sin(w), d) <= cos_ulps * 1.1102230246251565e-16
    rel = np.where(in_cos_noise, 0.0, d / np.where(w == 0, 1.0, w))
    return float(rel.max())


def wild_plume_errors(inputs, got, want, torr2pa, radius=1.0):
    """Compare plume results on inputs far outside the priors (tests/golden/plume_fuzz.npz; tools/fuzz_parity.py states
    the rules): NaN / inf patterns and invalid rows identical; j_ion to 1e-10 of the value plus the rounding floors of
    1 - exp(-x) and of the sample's largest entry; div_angle / T_c where they are defined by normal-range arithmetic
    (beam amplitude not denormal, both amplitudes of one sign, beams wider than a quarter grid step).
    got / want: dicts with j_ion (n, 91), div_angle (n,), T_c (n,).  Returns the worst errors."""
    x = {k: np.asarray(v, dtype=np.float64) for k, v in inputs.items()}
    gj, wj = np.asarray(got['j_ion'], dtype=np.float64).reshape(-1, 91), np.asarray(want['j_ion'], dtype=np.float64).reshape(-1, 91)
    assert np.array_equal(np.isnan(gj), np.isnan(wj)), 'j_ion NaN pattern differs'
    assert np.array_equal(np.isinf(gj), np.isinf(wj)) and np.array_equal(np.sign(gj[np.isinf(gj)]), np.sign(wj[np.isinf(wj)]))
    assert np.array_equal(np.all(gj == 1e-20, axis=1), np.all(wj == 1e-20, axis=1)), 'invalid rows differ'
    with np.errstate(all='ignore'):
        peak = np.nan_to_num(np.max(np.where(np.isfinite(wj), np.abs(wj), 0.0), axis=1, keepdims=True))
        P_B0 = x['P_b'] * torr2pa
        one_minus_decay = np.abs(1.0 - np.exp(-radius * (x['c4'] * P_B0 + x['c5']) * x['sigma_cex']))[:, None]
        floor = (np.abs(x['I_B0'])[:, None] / (2 * np.pi * radius ** 2) * (8 * np.finfo(float).eps + 1e-13 * one_minus_decay)
                 + 1e-13 * peak)
        fin = np.isfinite(wj) & np.isfinite(floor)
        err_j = float(np.max((np.abs(gj - wj) / (np.abs(wj) + 1e10 * floor + 1e-300))[fin], initial=0.0))
        P_B = x['P_b'] * torr2pa
        base = x['I_B0'] * np.exp(-radius * (x['c4'] * P_B + x['c5']) * x['sigma_cex']) / radius ** 2
        a1 = np.minimum(x['c2'] * P_B + x['c3'], np.pi / 2)
        ok = (~((np.abs(base) < 1e-280) & (base != 0.0)) & (x['c0'] >= 0) & (x['c0'] <= 1)
              & ~(np.maximum(np.abs(a1), np.abs(a1 / x['c1'])) < 0.0044))
    gd, wd = np.asarray(got['div_angle']).reshape(-1), np.asarray(want['div_angle']).reshape(-1)
    gt, wt = np.asarray(got['T_c']).reshape(-1), np.asarray(want['T_c']).reshape(-1)
    assert np.array_equal(np.isnan(gt[ok]), np.isnan(wt[ok])), 'T_c NaN pattern differs'
    assert np.array_equal(np.isnan(gd[ok]), np.isnan(wd[ok])), 'div_angle NaN pattern differs'
    return {'j_ion': err_j, 'div_angle': div_err(gd[ok], wd[ok]), 'T_c': rel_err(gt[ok], wt[ok]), 'compared': int(ok.sum())}
