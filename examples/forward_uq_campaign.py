#!/usr/bin/env python3
"""A forward-UQ campaign on one MI355X, as scripts/pem_v0/monte_carlo.py and scripts/gen_data.py run it on the CPU:
sample the PEM-v0 priors, evaluate cathode -> thruster (analytic test double) -> plume, mark NaN / outlier samples, and take the
5 / 50 / 95 % bands of every output.  Everything stays on the device; the percentiles equal numpy's bit for bit.

One driver call (round 4): the profile's percentiles and outlier counts are taken inside the evaluation launch
(drivers.forward_uq_statistics); the three separate calls of round 3 -- forward_uq, filter_outputs, percentile_bands -- give the
same numbers from a stored profile and are shown for comparison.

    python examples/forward_uq_campaign.py [n_samples]          (default 1e6; 1e7 = BASELINE configs[2], about 5 ms of GPU time)
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers          # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
t0 = time.perf_counter()
out = drivers.forward_uq_statistics(n, seed=0, keep_profile=True)       # V_cc, div_angle, T_c, I_B0, T, invalid, j_ion + nan_idx, outlier_idx, bands
discard = drivers.discard_mask(out['nan_idx'], out['outlier_idx'], discard_outliers=True)     # gen_data.py:177-215
bands = out['bands']                                                    # monte_carlo.py:363-658: (3, ...) per output
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'{n} samples in {1e3 * dt:.1f} ms (first call: includes allocations); {int(discard.sum())} samples flagged NaN / outlier'
      f' (percentiles counted inside the evaluation launch: {out["fused"]}, outlier counts too: {out["premasked"]})')
# the same campaign as three calls over the stored profile (gen_data.py:125-174, monte_carlo.py:363-658): identical numbers
fields = {k: out[k] for k in ('V_cc', 'div_angle', 'T_c', 'j_ion')}
nan_idx, outlier_idx = drivers.filter_outputs(fields)
bands3 = drivers.percentile_bands(out)
same = all(torch.equal(bands[k], bands3[k]) for k in bands3) and all(torch.equal(outlier_idx[k], out['outlier_idx'][k]) for k in outlier_idx)
print(f'  the separate calls over the stored profile give the same bands and masks: {same}')
for k in ('V_cc', 'div_angle', 'T_c'):
    lo, med, hi = (float(v) for v in bands[k])
    print(f'  {k:<10} 5 % {lo:.6g}   50 % {med:.6g}   95 % {hi:.6g}')
j = bands['j_ion']
print(f'  j_ion      median on the axis {float(j[1, 0]):.4g} A/m^2, at 90 degrees {float(j[1, -1]):.4g} A/m^2')
