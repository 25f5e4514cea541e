"""Differential fuzzing against the oracle on inputs far outside the priors (tools/fuzz_parity.py): every NaN / inf /
invalid pattern identical, every finite value inside the per-entry bound of tests/parity_rules.py (1e-10 of the result
plus 3e-13 of the magnitudes of the terms it is summed from).  Seeds 0..59 include the three that exposed deep-underflow
differences while the kernels were being written:
  seed 2   beam amplitude in the denormal range -> the reference's 0/0 = NaN in div_angle (reduced-QoI tables);
  seed 6   sigma_cex = 0 and a narrow beam -> exp() exactly 0 in the reference's tail, j_ion = 0, sample invalid;
  seed 53  negative density, exp(+703) -> infinite amplitude, which must become NaN where the reference's exp() is 0.
The named seeds are the ones round-1 campaigns ended red on ("what's weak" 1-2), kept as regression cases:
  seed 65   R = 1 fast path, beams of opposite sign (c0 outside [0, 1]): div_angle off by 2.6e-9 relative;
  seed 867  three radii, beams of opposite sign: div_angle off by 5e-6 before the wave-per-sample kernel summed such
            samples angle by angle as the reference does;
  seed 940  three radii: j_ion 1.3e-9 relative where a negative j_cex cancels the beams;
  seed 1100 coupled, full profile: j_ion 1.01e-10 relative, same cancellation;
  seed 269  five radii, a sample whose only non-zero beam is narrower than the 1-degree grid: cos_div = 1 - 2 eps in the
            oracle and 1 on the device, both inside the 4 eps bound, i.e. div_angle = 3.0e-8 against 0.  Round 2's campaign
            stopped here because the CHECKER mapped the angle difference back to the cosine as d^2 (it is d^2 / 2 at the
            pole); `divergence_error` / `conftest.div_err` evaluate |cos a - cos b| exactly since.  The seed keeps the R = 5
            path and the metric under the driver-run suite; `test_divergence_metric_at_the_pole` pins the metric itself.
  seed 5160 (round 4) R = 1, a beam one grid step wide (alpha1 = 0.016 rad, c1 at the edge of the wild range): the terms that
            carry the Simpson sums have u^2 = (alpha / a)^2 of 1 .. 5, each reproduced to 1.5 eps u^2 at best, and the two sums come
            out 8 eps apart -- cos_div = 0.999996, so 2.3e-10 of a divergence angle of 0.0028 rad.  No cancellation (cond = 1): the
            bound's TAU x (cond - 1) part did not cover term errors; `plume_bounds` now adds their |term|-weighted mean."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
NAMED_SEEDS = ['65', '269', '867', '940', '1100', '5160']


@pytest.mark.gpu
def test_wild_inputs_match_the_oracle(monkeypatch):
    sys.path.insert(0, str(ROOT / 'tools'))
    import fuzz_parity
    monkeypatch.setattr(sys, 'argv', ['fuzz_parity.py', '--seeds', '60', '--seed-list', *NAMED_SEEDS, '--n', '20000'])
    fuzz_parity.main()


def test_divergence_metric_at_the_pole():
    """The two comparisons of div_angle = arccos(cos_div) (`conftest.div_err`, `parity_rules.divergence_error`) at the pole
    cos_div -> 1, with the pair seed 269 produced: cos_div = 1 - 2 eps on one side, exactly 1 on the other.  The angles are
    arccos(1 - 2 eps) = 3.0e-8 and 0 -- relative difference 1 -- while the cosines differ by 2 eps, inside every bound used
    (4 eps for a quotient of two rounded sums): the pair must pass, and a pair whose cosines differ by more must not."""
    import parity_rules as pr
    from conftest import div_err
    eps = pr.EPS                                        # 2^-52, the eps of parity_rules
    a = np.arccos(np.array([1.0 - 2 * eps]))            # 2.98e-08
    z = np.array([0.0])
    assert 2.9e-8 < a[0] < 3.0e-8
    # conftest.div_err: 8 ulp of cos_div by default
    assert div_err(a, z) == 0.0 and div_err(z, a) == 0.0
    # 2 eps is more than cos_ulps = 1 allows: the angle's own error is reported (relative; absolute where the wanted angle is 0)
    assert div_err(z, a, cos_ulps=1) == 1.0 and div_err(a, z, cos_ulps=1) == a[0]
    far = np.arccos(np.array([1.0 - 64 * eps]))
    assert div_err(z, far) == 1.0 and div_err(far, z) == far[0] > 1e-7
    # away from the pole the same 2 eps of the cosine is 1e-16 of the angle and passes as a relative error
    mid = np.arccos(np.array([0.5])), np.arccos(np.array([0.5 - 2 * eps]))
    assert 0.0 <= div_err(mid[0], mid[1]) < 1e-15
    # parity_rules.divergence_error with the bound of a plain sample (cos_rel = 4 eps)
    bounds = {'comparable': np.array([True]), 'cos_div': np.array([1.0]), 'cos_rel': np.array([4 * eps]), 'cond_cos': np.array([1.0])}
    for got, want in ((a, z), (z, a)):
        r = pr.divergence_error(got, want, None, None, bounds)
        assert r['err_div'] == 0.0, r
    r = pr.divergence_error(z, far, None, None, bounds)
    assert r['err_div'] == 1.0                           # 64 eps of the cosine is outside the bound: the angle's own error is reported
    # the mapping that stopped round 2's campaign, for the record: d^2 overstates the cosine difference at the pole by 2
    d = float(a[0])
    assert abs(2 * np.sin(0.5 * d) ** 2 - 2 * eps) < 1e-3 * eps and d * d > 3.9 * eps
