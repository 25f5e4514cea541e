#!/usr/bin/env python3
"""drivers.forward_uq at BASELINE configs[2] size (1e7 samples, one GPU): launches of `batch_size` samples over ranges of the
resident batch against one launch over all of it.  python tools/forward_uq_batch_probe.py"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers


def wall(fn, reps=7, warm_ms=40.0):
    """best wall time of `reps` calls, after calls for `warm_ms` (the clocks ramp for ~15 ms after an idle period: profiles/warmup_r03.txt)"""
    t0 = time.perf_counter()
    while True:
        r = fn(); torch.cuda.synchronize(); del r
        if time.perf_counter() - t0 >= 1e-3 * warm_ms:
            break
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); del r
    return best


n = 10_000_000
for keep in (True, False):
    for bs in (n, 1 << 22, 1 << 21, 1 << 20, n):
        t = wall(lambda: drivers.forward_uq(n, seed=2, keep_profile=keep, batch_size=bs))
        print(f'forward_uq n={n} keep_profile={keep} batch_size={bs:>9}: {t * 1e3:7.3f} ms = {n / t / 1e9:5.2f} G evals/s', flush=True)
for keep in (True, False):
    t = wall(lambda: drivers.forward_uq(n, seed=2, keep_profile=keep, keep_inputs=False))
    print(f'forward_uq n={n} keep_profile={keep} one launch, keep_inputs=False: {t * 1e3:7.3f} ms = {n / t / 1e9:5.2f} G evals/s', flush=True)
