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
