#!/usr/bin/env python3
"""drivers.forward_uq at BASELINE configs[2] size (1e7 samples, one GPU): the resident-batch form against the form it
replaced (a scratch batch per 2^21 samples + copies of every result array), wall time including the sampling."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd import drivers, sampling
from hallthrusterpem_amd.batch import CoupledBatch, QOI_NAMES


def old_forward_uq(n, seed=0, profile=False, batch_size=1 << 21):
    design = sampling.Design(seed=seed)
    batch = CoupledBatch(batch_size, profile=profile)
    dev = batch.device
    out = {k: torch.empty(n, dtype=torch.float64, device=dev) for k in QOI_NAMES + ('I_B0', 'T')}
    out['invalid'] = torch.empty(n, dtype=torch.bool, device=dev)
    out['x'] = torch.empty((design.ndim, n), dtype=torch.float64, device=dev)
    for off in range(0, n, batch_size):
        m = min(batch_size, n - off)
        if m < batch_size:
            batch = CoupledBatch(m, device=dev, profile=profile)
        batch.run_mc(design, first_index=off, write_inputs=True)
        res = batch.outputs()
        sl = slice(off, off + m)
        for k in QOI_NAMES + ('I_B0', 'T', 'invalid'):
            out[k][sl] = res[k]
        out['x'][:, sl] = batch.inputs
    return out


def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); del r
    return best


n = 10_000_000
for prof in (False, True):
    a = wall(lambda: old_forward_uq(n, seed=2, profile=prof))
    b = wall(lambda: drivers.forward_uq(n, seed=2, profile=prof))
    x, y = old_forward_uq(n, seed=2, profile=prof), drivers.forward_uq(n, seed=2, profile=prof)
    same = all(torch.equal(x[k], y[k]) for k in x)
    print(f'forward_uq, 1e7 samples, profile={prof}: scratch batch + copies {a * 1e3:.2f} ms -> resident batch, in-place ranges {b * 1e3:.2f} ms '
          f'({n / b / 1e9:.2f} G evals/s incl. sampling and allocation); results identical: {same}')
