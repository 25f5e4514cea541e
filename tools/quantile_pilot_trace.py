#!/usr/bin/env python3
"""For rocprofv3 (tools/profile_cmd.sh <tag> tools/quantile_pilot_trace.py): the pilot form of pem_quantiles_f64_dev on a
1e7 x 91 profile array, [5, 50, 95] and [25, 75], four calls each, then the four passes (PEM_QUANTILE_PILOT=0) once each."""
import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
a = drivers.forward_uq(n, seed=2, keep_profile=True)['j_ion']
for stride, reps in (('32', 4), ('0', 1)):
    os.environ['PEM_QUANTILE_PILOT'] = stride
    for _ in range(reps):
        drivers.column_percentiles(a, [5.0, 50.0, 95.0])
        drivers.column_percentiles(a, [25.0, 75.0])
torch.cuda.synchronize()
