#!/usr/bin/env python3
"""One forward-UQ campaign of BASELINE configs[2] on one GPU, stage by stage: sample + evaluate (drivers.forward_uq, profile kept),
NaN / interquartile-range masks (drivers.filter_outputs = gen_data.py:125-174) and the 5 / 50 / 95 % bands
(drivers.percentile_bands = monte_carlo.py:363-658).  Wall time per stage, best of five, results left on the device.
    python tools/campaign_probe.py [n]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000


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


out = drivers.forward_uq(n, seed=2, keep_profile=True)
keep = {k: out[k] for k in ('V_cc', 'div_angle', 'T_c', 'j_ion')}
t_model = wall(lambda: drivers.forward_uq(n, seed=2, keep_profile=True))
t_filter = wall(lambda: drivers.filter_outputs(keep))
t_bands = wall(lambda: drivers.percentile_bands(out))
total = t_model + t_filter + t_bands
t_stats = wall(lambda: drivers.campaign_statistics(keep))
t_qj = wall(lambda: drivers.column_percentiles(out['j_ion'], [25.0, 75.0, 5.0, 50.0, 95.0]))
t_qs = wall(lambda: drivers.column_percentiles(drivers._stacked_rows([out[k] for k in ('V_cc', 'div_angle', 'T_c')]), [25.0, 75.0, 5.0, 50.0, 95.0]))
print(f'forward-UQ campaign, {n} coupled samples, one MI355X (fp64, 91-point profile kept):')
print(f'  sample + evaluate (forward_uq)          {t_model * 1e3:8.2f} ms   {n / t_model / 1e9:6.2f} G evals/s')
print(f'  NaN / IQR masks (filter_outputs)        {t_filter * 1e3:8.2f} ms')
print(f'  5 / 50 / 95 % bands (percentile_bands)  {t_bands * 1e3:8.2f} ms')
print(f'  whole campaign                          {total * 1e3:8.2f} ms   {n / total / 1e9:6.2f} G evals/s')
print(f'  masks + bands from ONE selection of five quantiles per variable (campaign_statistics) {t_stats * 1e3:8.2f} ms'
      f'   [five quantiles of j_ion {t_qj * 1e3:.2f} ms, of the three scalars in one call {t_qs * 1e3:.2f} ms]')
print(f'  whole campaign, that way                {(t_model + t_stats) * 1e3:8.2f} ms   {n / (t_model + t_stats) / 1e9:6.2f} G evals/s')
t_fused = wall(lambda: drivers.forward_uq_statistics(n, seed=2, keep_profile=True))
t_fused_np = wall(lambda: drivers.forward_uq_statistics(n, seed=2, keep_profile=False))
r = drivers.forward_uq_statistics(n, seed=2, keep_profile=True)
print(f'  forward_uq_statistics: sample + evaluate + count on chip, masks, bands (fused={r["fused"]})  {t_fused * 1e3:8.2f} ms   {n / t_fused / 1e9:6.2f} G evals/s')
print(f'  the same without a stored profile (scalar masks, all bands)                               {t_fused_np * 1e3:8.2f} ms   {n / t_fused_np / 1e9:6.2f} G evals/s')
