#!/usr/bin/env python3
"""Time the REFERENCE's NumPy path in the build container: `cathode_coupling` + `current_density`
imported from /root/reference exactly as tests/golden/make_golden.py imports them, seeded inputs, chunks of 1e5 samples,
best of 3 warm runs, one process (NumPy elementwise code is single-threaded: one effective core).

Writes profiles/reference_numpy_baseline.json, which bench.py replays as cpu_baseline.reference_numpy -- the reference
itself cannot travel to the GPU box.  Usage:  python tests/golden/time_reference.py [n_total] [chunk]
"""
import json
import os
import platform
import sys
import time
from pathlib import Path

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT / 'tests'))

import numpy as np

import make_golden as mg          # the loader of the reference's two model files behind the in-memory pem_core
from _inputs import coupled_inputs  # the seeded PEM-v0 prior draws every parity test uses


def main():
    n_total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000
    chunk = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000
    cathode, plume, _thruster, _const = mg._load_reference()
    x = coupled_inputs(n_total, seed=10)            # the inputs bench.py's cpu_baseline leg times the C oracle on

    def coupled_chunk(lo, hi):
        sl = {k: np.ascontiguousarray(v[lo:hi]) for k, v in x.items()}
        vcc = cathode.cathode_coupling({k: sl[k] for k in ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')})['V_cc']
        # the analytic thruster stage of tests/sim_hallthruster.jl:35-48 is a handful of flops per sample: it is left out of the
        # timing, the plume takes the prior's own I_B0 range instead (as BASELINE.md section 2 timed the two models)
        ins = {k: sl[k] for k in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')}
        ins['I_B0'] = 2.0 + 6.0 * (np.arange(hi - lo) % 1000) / 1000.0
        ins['T'] = np.full(hi - lo, 0.08)
        out = plume.current_density(ins, sweep_radius=1.0)
        return vcc, out['j_ion']

    def one_pass():
        t0 = time.perf_counter()
        for lo in range(0, n_total, chunk):
            vcc, j = coupled_chunk(lo, min(n_total, lo + chunk))
        return time.perf_counter() - t0, vcc, j

    one_pass()                                       # cold: imports, page faults
    runs = []
    for _ in range(int(os.environ.get('PEM_REF_PASSES', '3'))):
        dt, vcc, j = one_pass()
        runs.append(dt)
    assert np.isfinite(vcc).all() and j.shape[1] == 91
    best = min(runs)
    rec = {
        'value': n_total / best, 'unit': 'evals/s', 'cores': 1,
        'samples': n_total, 'chunk': chunk, 'runs_s': runs, 'best_s': best, 'seed': 10, 'TORR_2_PA': mg.TORR_2_PA,
        'what': 'hallmd.models.cathode.cathode_coupling + hallmd.models.plume.current_density (NumPy / SciPy; 91 angles, R = 1 at 1.0 m, T '
                'supplied) of /root/reference, imported as tests/golden/make_golden.py imports them; PEM-v0 prior draws '
                '(tests/_inputs.coupled_inputs, seed 10); chunks of `chunk` samples; best of the warm passes listed in runs_s (BASELINE.md section 3.1: best of 3; PEM_REF_PASSES adds more); one process',
        'host': {'cpu': next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), platform.processor()),
                 'logical_cpus': os.cpu_count(), 'python': platform.python_version(), 'numpy': np.__version__},
        'reference_lines': 'src/hallmd/models/cathode.py:16-38, src/hallmd/models/plume.py:21-159',
        'generator': 'tests/golden/time_reference.py',
    }
    out = ROOT / 'profiles' / 'reference_numpy_baseline.json'
    out.write_text(json.dumps(rec, indent=1) + '\n')
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
