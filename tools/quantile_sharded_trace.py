#!/usr/bin/env python3
"""The sharded percentile selection on one rank, for rocprofv3 --kernel-trace: 5 / 50 / 95 % of the 91-point profiles of a
forward-UQ campaign of N samples (argv[1], default 1.25e6: one GPU's shard of BASELINE configs[2]), five times."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers                                   # noqa: E402
from hallthrusterpem_amd.percentiles import column_percentiles_sharded    # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_250_000
a = drivers.forward_uq(n, seed=2, keep_profile=True)['j_ion']
for _ in range(5):
    column_percentiles_sharded(a, [5.0, 50.0, 95.0])
print('done', n)
