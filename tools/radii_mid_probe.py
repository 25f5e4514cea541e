#!/usr/bin/env python3
"""plume_rmid_kernel (9..64 sweep radii) by radius count: TB/s of output written, the default (samples, passes) packing against the
one-pass packing of round 3 (PEM_RMID_SP).  python tools/radii_mid_probe.py [R ...]"""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'tests'))
from _inputs import plume_inputs
from hallthrusterpem_amd.models import current_density

Rs = [int(a) for a in sys.argv[1:]] or [17, 18, 22, 25, 27, 31, 32, 33, 40, 44, 48, 50, 64]


def rate(n, R):
    x = {k: torch.as_tensor(v).cuda() for k, v in plume_inputs(n, seed=3).items()}
    radii = np.linspace(0.5, 1.5, R)
    for _ in range(3):
        out = current_density(x, sweep_radius=radii)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            out = current_density(x, sweep_radius=radii)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 4)
    return n * (91 * R + 2 * R) * 8 / best / 1e9     # TB/s


for R in Rs:
    n = max(20_000, int(2.5e9 / (91 * R * 8)) // 64 * 64)       # ~2.5 GB of profile per call
    os.environ.pop('PEM_RMID_SP', None)
    new = rate(n, R)
    os.environ['PEM_RMID_SP'] = f'{max(1, 64 // R)},1'
    old = rate(n, R)
    os.environ.pop('PEM_RMID_SP', None)
    print(f'R = {R:2d} (n = {n}): {new:5.2f} TB/s of output with the default packing, {old:5.2f} with {max(1, 64 // R)} samples in one pass (round 3)')
