#!/usr/bin/env python3
"""Five quantiles of a 1e7 x 91 profile in one selection (rocprofv3 --kernel-trace --stats target).  python tools/quantile5_probe.py [n] [reps]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
out = drivers.forward_uq(n, seed=2, keep_profile=True)
for pcts in ([25.0, 75.0, 5.0, 50.0, 95.0], [5.0, 50.0, 95.0], [25.0, 75.0]):
    for _ in range(3):
        drivers.column_percentiles(out['j_ion'], pcts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        drivers.column_percentiles(out['j_ion'], pcts)
    torch.cuda.synchronize()
    print(pcts, f'{(time.perf_counter() - t0) / reps * 1e3:.2f} ms')
