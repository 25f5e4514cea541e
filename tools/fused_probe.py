#!/usr/bin/env python3
"""drivers.forward_uq_statistics under rocprofv3 (tools/kstats.sh): python tools/fused_probe.py [n] [keep_profile 0/1] [reps]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keep = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for _ in range(3):
    r = drivers.forward_uq_statistics(n, seed=2, keep_profile=keep)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    r = drivers.forward_uq_statistics(n, seed=2, keep_profile=keep)
torch.cuda.synchronize()
print(f'forward_uq_statistics n={n} keep_profile={keep} fused={r["fused"]}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms')
if keep and len(sys.argv) > 4 and sys.argv[4] == 'check':
    # the same statistics from the stored profile (two-pass selection + masks pass, themselves held to numpy by tests/test_quantiles.py
    # and tools/quantile_stress.py): identical doubles, identical masks
    fields = {k: r[k] for k in ('V_cc', 'div_angle', 'T_c', 'j_ion')}
    nan_idx, outlier_idx = drivers.filter_outputs(fields)
    bands = drivers.percentile_bands(r)
    same = all(torch.equal(bands[k], r['bands'][k]) for k in bands) and all(torch.equal(outlier_idx[k], r['outlier_idx'][k]) for k in outlier_idx) \
        and all(torch.equal(nan_idx[k], r['nan_idx'][k]) for k in nan_idx)
    print(f'  equal to filter_outputs + percentile_bands over the stored profile (bands of {len(bands)} outputs, masks of {len(outlier_idx)}): {same}')
    if n <= 2_000_000:
        import numpy as np
        j = r['j_ion'].cpu().numpy()
        print('  j_ion bands equal np.percentile bit for bit:', bool(np.array_equal(r['bands']['j_ion'].cpu().numpy(), np.percentile(j, [5.0, 50.0, 95.0], axis=0))))
    sys.exit(0 if same else 1)
