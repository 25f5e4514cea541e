"""Seeded synthetic inputs shared by the tests (ranges: tests/test_cathode.py:19-21, tests/test_plume.py:20-28,
scripts/pem_v0/pem_v0_SPT-100.yml priors -- SURVEY.md Appendix A)."""
import numpy as np


def cathode_inputs(n, seed=0, lhs=False):
    rng = np.random.default_rng(seed)
    if lhs:   # Latin hypercube: one stratum per sample and dimension, independently permuted
        u = np.stack([(rng.permutation(n) + rng.random(n)) / n for _ in range(6)])
    else:
        u = rng.random((6, n))
    return {'P_b': 10 ** (u[0] * 4 - 8), 'V_a': u[1] * 200 + 200, 'T_e': u[2] * 4 + 1, 'V_vac': u[3] * 60,
            'Pstar': u[4] * 90e-6 + 10e-6, 'P_T': u[5] * 90e-6 + 10e-6}


def plume_inputs(n, seed=1, priors=True, with_T=True):
    rng = np.random.default_rng(seed)
    u = rng.random((10, n))
    x = {'P_b': 10 ** (u[0] * 4 - 8), 'c1': u[2] * 0.8 + 0.1, 'c2': u[3] * 30 - 15,
         'c4': 10 ** (u[5] * 4 + 18), 'c5': 10 ** (u[6] * 4 + 14), 'sigma_cex': u[7] * 7e-20 + 51e-20,
         'I_B0': u[8] * 6 + 2}
    if priors:      # yml:221-246
        x['c0'] = u[1]
        x['c3'] = u[4] * (1.570796 - 0.2) + 0.2
    else:           # tests/test_plume.py:21,24 (reaches alpha1 <= 0)
        x['c0'] = u[1] * 0.8 + 0.1
        x['c3'] = u[4] + 0.1
    if with_T:
        x['T'] = u[9] * 0.1 + 0.02
    return x


def coupled_inputs(n, seed=2):
    rng = np.random.default_rng(seed)
    u = rng.random((15, n))
    return {'P_b': 10 ** (u[0] * 4 - 8), 'V_a': u[1] * 200 + 200, 'T_e': u[2] * 4 + 1, 'V_vac': u[3] * 60,
            'Pstar': u[4] * 90e-6 + 10e-6, 'P_T': u[5] * 90e-6 + 10e-6,
            'mdot_a': u[6] * 5e-6 + 2e-6, 'a_1': 10 ** (u[7] * 1.5 - 2.5),
            'c0': u[8], 'c1': u[9] * 0.8 + 0.1, 'c2': u[10] * 30 - 15, 'c3': u[11] * (1.570796 - 0.2) + 0.2,
            'c4': 10 ** (u[12] * 4 + 18), 'c5': 10 ** (u[13] * 4 + 14), 'sigma_cex': u[14] * 7e-20 + 51e-20}
