#!/usr/bin/env python3
"""current_density with 17 ... 64 sweep radii (the staged kernel, csrc/pem_kernels.hip plume_rmid_kernel): TB/s of output per radius count.
Set PEM_HIP_LIB=build_variants/libpem_<name>.so (tools/build_variant_fast.sh) to compare variants, one process per variant, interleaved
by the caller; `profiles/radii_mid_r04.txt`."""
import os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from _inputs import plume_inputs
from hallthrusterpem_amd.models import current_density
for R in (tuple(int(r) for r in os.environ['PEM_PROBE_RADII'].split(',')) if os.environ.get('PEM_PROBE_RADII') else (17, 25, 32, 33, 48, 64)):
    n = max(20_000, int(2.5e9 / (91 * R * 8)) // 64 * 64)
    x = {k: torch.as_tensor(v).cuda() for k, v in plume_inputs(n, seed=3).items()}
    radii = np.linspace(0.5, 1.5, R)
    for _ in range(3): out = current_density(x, sweep_radius=radii)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4): out = current_density(x, sweep_radius=radii)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 4)
    print(os.environ.get('PEM_HIP_LIB', 'tree')[-20:], f'R={R}: {n * (91 * R + 2 * R) * 8 / best / 1e9:.2f} TB/s')
