#!/usr/bin/env python3
"""Wall time per call of the drop-in callables (numpy in -> numpy out) against batch size: the regime amisc uses
while training (a few to a few thousand samples per call)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd.models import cathode_coupling, current_density, pem_v0_coupled
from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
rng = np.random.default_rng(0)


def wall(fn, reps):
    for _ in range(5): fn()
    t = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t) / reps * 1e6


for n in (1, 10, 100, 1000, 10_000, 100_000):
    cat = {'P_b': 10.0 ** rng.uniform(-8, -4, n), 'V_a': rng.uniform(200, 400, n), 'T_e': rng.uniform(1, 5, n),
           'V_vac': rng.uniform(0, 60, n), 'Pstar': rng.uniform(1e-5, 1e-4, n), 'P_T': rng.uniform(1e-5, 1e-4, n)}
    plu = {'P_b': cat['P_b'], 'c0': rng.uniform(0, 1, n), 'c1': rng.uniform(0.1, 0.9, n), 'c2': rng.uniform(-15, 15, n),
           'c3': rng.uniform(0.2, 1.57, n), 'c4': 10.0 ** rng.uniform(18, 22, n), 'c5': 10.0 ** rng.uniform(14, 18, n),
           'sigma_cex': rng.uniform(51e-20, 58e-20, n), 'I_B0': rng.uniform(2, 8, n)}
    cpl = {**cat, **{k: plu[k] for k in ('c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')},
           'mdot_a': rng.uniform(2e-6, 7e-6, n), 'a_1': rng.uniform(0.003, 0.1, n)}
    reps = 200 if n <= 10_000 else 30
    print(f'n={n:7d}: cathode_coupling {wall(lambda: cathode_coupling(cat), reps):8.1f} us   '
          f'current_density {wall(lambda: current_density(plu), reps):8.1f} us   '
          f'pem_v0_coupled {wall(lambda: pem_v0_coupled(cpl), reps):8.1f} us')
