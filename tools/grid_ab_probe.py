#!/usr/bin/env python3
"""Launch-size sensitivity of the persistent grids (set PEM_BALANCED_GRID=0/1): reduced-QoI, fused Monte-Carlo, fused
likelihood and fused compression launches of 1.25e6 samples, microseconds per launch."""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.sampling import Design
n = 1_250_000
d = Design(seed=2)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
red = CoupledBatch(n, profile=False, thruster_qoi=False); d.fill(red.inputs)
full = CoupledBatch(n, profile=True, thruster_qoi=False); d.fill(full.inputs)
mixed = CoupledBatch(n, profile=True, mixed=True, thruster_qoi=False); d.fill(mixed.inputs)
print(f"PEM_BALANCED_GRID={os.environ.get('PEM_BALANCED_GRID', 'default')}: reduced {t(red.run):.1f} us | fused MC reduced {t(lambda: red.run_mc(d)):.1f} us | "
      f"profile (one batch) {t(full.run):.1f} us | fused MC profile {t(lambda: full.run_mc(d)):.1f} us | mixed {t(mixed.run):.1f} us")
