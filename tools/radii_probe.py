#!/usr/bin/env python3
"""Throughput of current_density with a sweep_radius array (tests/test_plume.py:31 uses 25 radii): device path."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'tests'))
from _inputs import plume_inputs
from hallthrusterpem_amd.models import current_density
for n, R in ((100_000, 25), (400_000, 5), (400_000, 4), (600_000, 3), (1_000_000, 2)):
    x = {k: torch.as_tensor(v).cuda() for k, v in plume_inputs(n, seed=3).items()}
    radii = np.linspace(0.5, 1.5, R)
    for _ in range(2): out = current_density(x, sweep_radius=radii)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): out = current_density(x, sweep_radius=radii)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    by = n * (91 * R + 2 * R) * 8
    print(f'n={n} R={R}: {ms*1e3:.0f} us per call, {by/ms/1e6:.0f} GB/s of output, {n/ms/1e3:.1f} M samples/s')
