#!/usr/bin/env python3
"""plume_rmid_kernel (9..64 sweep radii: rows staged in LDS, several samples in flight per wave) against the wave-per-sample
kernel it replaces there (PEM_RADII_MID=0): device path of current_density into preallocated outputs, ~1.8 GB of profiles per
call, four output sets in rotation.  Run once per mode (the library reads the switch once)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'tests'))
from _inputs import plume_inputs                      # noqa: E402
from hallthrusterpem_amd import _lib, constants       # noqa: E402

lib = _lib.load()
print(f'# PEM_RADII_MID={os.environ.get("PEM_RADII_MID", "1")}')
for R in (9, 12, 16, 17, 25, 32, 33, 48, 64):
    n = int(1.8e9 / (91 * R * 8))
    x = {k: torch.as_tensor(v).cuda() for k, v in plume_inputs(n, seed=3).items()}
    radii = np.linspace(0.5, 1.5, R)
    sets = [(torch.empty((n, 91, R), dtype=torch.float64, device='cuda'), torch.empty((n, R), dtype=torch.float64, device='cuda'),
             torch.empty((n, R), dtype=torch.float64, device='cuda'), torch.empty(n, dtype=torch.uint8, device='cuda')) for _ in range(4)]
    p = lambda t: C.c_void_p(t.data_ptr())            # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(i):
        j, d, tc, inv = sets[i % 4]
        _lib.check(lib.pem_plume_f64_dev(n, R, C.c_void_p(radii.ctypes.data), constants.TORR_2_PA, *[p(x[k]) for k in
                   ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex', 'I_B0', 'T')], p(j), p(d), p(tc), p(inv), st))
    for i in range(4):
        call(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(12):
        call(i)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 12
    by = n * (91 * R + 2 * R) * 8
    print(f'R={R:3d} n={n:7d}: {ms * 1e3:7.0f} us per call, {by / ms / 1e9:5.2f} TB/s of output', flush=True)
    del sets, x
    torch.cuda.empty_cache()
