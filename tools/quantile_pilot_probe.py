#!/usr/bin/env python3
"""pem_quantiles_f64_dev: the pilot form (every 32nd row brackets the ranks, two passes over the data) against the four passes,
same process, interleaved, on forward-UQ outputs.  PEM_QUANTILE_PILOT is read at every call.
    python tools/quantile_pilot_probe.py [stride ...]       (default strides: 0 = four passes, 16, 32, 64)"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers, _lib

strides = [int(x) for x in sys.argv[1:]] or [0, 16, 32, 64]
lib = _lib.load()


def wall(fn, reps=7):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); del r
    return best


for n, m in ((1_250_000, 91), (10_000_000, 91), (10_000_000, 1), (10_000_000, 3), (600_000, 91)):
    out = drivers.forward_uq(n, seed=2, keep_profile=(m == 91))
    a = out['j_ion'] if m == 91 else (out['T_c'] if m == 1 else torch.stack([out['V_cc'], out['div_angle'], out['T_c']], dim=1).contiguous())
    for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0]):
        ref = None
        for s in strides:
            os.environ['PEM_QUANTILE_PILOT'] = str(s)
            got = drivers.column_percentiles(a, pcts)
            path = lib.pem_quantiles_last_path()
            if ref is None:
                ref = got
            same = bool(torch.equal(got.view(torch.int64), ref.view(torch.int64)))
            t = wall(lambda: drivers.column_percentiles(a, pcts))
            print(f'n={n:>9} m={m:>3} percentiles {str(pcts):<18} stride {s:>3} (path {path}): {t * 1e3:8.3f} ms = {a.numel() * 8 / t / 1e12:5.2f} TB/s per '
                  f'reading of the data; equal to the four passes bit for bit: {same}', flush=True)
    del out, a
    torch.cuda.empty_cache()
