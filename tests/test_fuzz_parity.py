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
  seed 1100 coupled, full profile: j_ion 1.01e-10 relative, same cancellation."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
NAMED_SEEDS = ['65', '867', '940', '1100']


@pytest.mark.gpu
def test_wild_inputs_match_the_oracle(monkeypatch):
    sys.path.insert(0, str(ROOT / 'tools'))
    import fuzz_parity
    monkeypatch.setattr(sys, 'argv', ['fuzz_parity.py', '--seeds', '60', '--seed-list', *NAMED_SEEDS, '--n', '20000'])
    fuzz_parity.main()
