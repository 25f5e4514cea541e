"""Second, independent restatement of the plume normaliser in numpy/scipy.  TEST INFRASTRUCTURE ONLY.

`normaliser_erfi` evaluates the literal complex-erfi bracket of the reference (src/hallmd/models/
plume.py:64-73) with scipy.special.erfi; `normaliser_quad` evaluates the integral it equals with
scipy.integrate.quad.  tests/test_oracle_golden.py uses both to pin oracle_normaliser() in
pem_oracle.c, whose Gauss-Legendre method is shared by nothing else in this repository.
"""
import numpy as np
from scipy import integrate, special


def normaliser_erfi(alpha):
    a = np.asarray(alpha, dtype=np.float64)
    with np.errstate(all='ignore'):
        gauss = np.exp(-(a / 2) ** 2)
        bracket = (2 * special.erfi(a / 2) + special.erfi((1j * np.pi - a ** 2) / (2 * a))
                   - special.erfi((1j * np.pi + a ** 2) / (2 * a)))
        return np.pi ** 1.5 / 2 * a * gauss * bracket


def normaliser_quad(alpha: float) -> float:
    a = abs(float(alpha))
    hi = min(np.pi / 2, 8.0 * a)
    val, _ = integrate.quad(lambda t: np.exp(-(t / a) ** 2) * np.sin(t), 0.0, hi, epsabs=0, epsrel=2e-14, limit=200)
    return 2 * np.pi * val
