#!/usr/bin/env python3
"""drivers.column_percentiles (exact selection, csrc/pem_quantile.hip) on forward-UQ sized outputs: wall time per call, the
HBM rate of its four streaming passes, torch.quantile beside it where that accepts the size."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers


def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0); del r
    return best


for n, m in ((1_000_000, 91), (10_000_000, 91), (10_000_000, 1), (1_000_000, 3)):
    out = drivers.forward_uq(n, seed=2, keep_profile=(m == 91))
    a = out['j_ion'] if m == 91 else (out['T_c'] if m == 1 else torch.stack([out['V_cc'], out['div_angle'], out['T_c']], dim=1).contiguous())
    for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0]):
        t = wall(lambda: drivers.column_percentiles(a, pcts))
        line = f'n={n:>9} m={m:>3} percentiles {pcts}: {t * 1e3:8.3f} ms ({4 * a.numel() * 8 / t / 1e12:5.2f} TB/s over four passes)'
        try:
            q = torch.tensor([p / 100 for p in pcts], dtype=torch.float64, device='cuda')
            tt = wall(lambda: torch.quantile(a, q, dim=0), reps=2)
            same = torch.allclose(torch.quantile(a, q, dim=0).reshape(len(pcts), -1), drivers.column_percentiles(a, pcts).reshape(len(pcts), -1), rtol=1e-12, atol=0)
            line += f' | torch.quantile {tt * 1e3:9.2f} ms, same to 1e-12: {same}'
        except RuntimeError as e:
            line += f' | torch.quantile: {str(e)[:60]}'
        print(line, flush=True)
    del out, a
    torch.cuda.empty_cache()

# the multi-rank form on ONE rank (no collective here): the four-pass selection, and the level loop of rounds 1-2
from hallthrusterpem_amd.percentiles import column_percentiles_sharded
for n in (1_250_000, 10_000_000):
    a = drivers.forward_uq(n, seed=2, keep_profile=True)['j_ion']
    for method in ('select', 'levels'):
        t = wall(lambda: column_percentiles_sharded(a, [5.0, 50.0, 95.0], method=method), reps=5)
        print(f'sharded form ({method}), one rank, n={n} m=91 percentiles [5, 50, 95]: {t * 1e3:.2f} ms', flush=True)
    del a
