#!/usr/bin/env python3
"""Randomised stress of drivers.forward_uq_statistics (the campaign statistics counted inside the evaluation launch, the scalar QoIs
selected on the library's side stream) against the separate calls over a stored profile -- filter_outputs + percentile_bands, themselves
held to numpy by tools/quantile_stress.py: sample counts 4096 ... 3e6 (ragged tiles, tiny pilots), seeds, percentile sets with and
without the premask, with and without a stored profile, priors that make a share of the samples invalid (declines).  Every band and mask
must be EQUAL.  python tools/fused_stress.py [--cases 200] [--seed 0]; `profiles/fused_stress_r04.txt`."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers, sampling      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--cases', type=int, default=200)
ap.add_argument('--seed', type=int, default=0)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
sets = [(5.0, 50.0, 95.0), (2.5, 97.5), (50.0,), (1.0, 10.0, 90.0, 99.0), (0.0, 100.0), (33.3, 66.6), (5.0, 50.0, 95.0)]
fused_n = declined_n = 0
t0 = time.time()
for case in range(args.cases):
    n = int(rng.choice([4096, 5000, 9999, 20_001, 65_536, 131_073, 400_000, 1_000_003, 3_000_000], p=[.1, .1, .1, .15, .15, .15, .15, .07, .03]))
    seed = int(rng.integers(0, 2 ** 31))
    pct = sets[int(rng.integers(0, len(sets)))]
    keep = bool(rng.integers(0, 2))
    pri = None
    if rng.random() < 0.15:                                  # a share of the samples invalid (alpha1 <= 0): ties at 1e-20 -> declines
        pri = dict(sampling.PEM_V0_PRIORS)
        pri['c3'] = sampling.Prior(sampling.UNIFORM, float(rng.uniform(-0.8, 0.1)), 1.1, 'stress')
    got = drivers.forward_uq_statistics(n, seed=seed, keep_profile=keep, percentiles=pct, priors=pri)
    ref = drivers.forward_uq(n, seed=seed, keep_profile=True, keep_inputs=False, priors=pri) if (pri is not None or not keep) else None
    prof = got['j_ion'] if keep else ref['j_ion']
    own = {k: got[k] for k in ('V_cc', 'div_angle', 'T_c')}
    want_b = drivers.percentile_bands(dict(own, j_ion=prof), percentiles=list(pct))
    nan_a, outl_a = drivers.filter_outputs(dict(own, j_ion=prof))
    where = f'case {case}: n={n} seed={seed} pct={pct} keep={keep} priors={"wild c3" if pri else "default"} fused={got["fused"]} premasked={got["premasked"]}'
    for k in want_b:
        same = torch.equal(got['bands'][k], want_b[k]) or torch.equal(torch.nan_to_num(got['bands'][k], nan=1e300), torch.nan_to_num(want_b[k], nan=1e300))
        assert same, f'{where}: bands of {k} differ'
    for k in nan_a:
        assert torch.equal(got['nan_idx'][k], nan_a[k]), f'{where}: nan_idx of {k} differs'
        assert torch.equal(got['outlier_idx'][k], outl_a[k]), f'{where}: outlier_idx of {k} differs'
    fused_n += int(got['fused'])
    declined_n += int(not got['fused'])
    if (case + 1) % 20 == 0:
        print(f'{case + 1} cases: all equal ({fused_n} answered on chip, {declined_n} declined); {time.time() - t0:.0f} s', flush=True)
print(f'{args.cases} cases (seed {args.seed}): every band and mask of forward_uq_statistics equal to the separate calls over the stored profile; '
      f'{fused_n} answered on chip, {declined_n} declined and fell back')
