#!/usr/bin/env python3
"""One sweep-radius count through pem_plume_f64_dev, a few launches (for rocprofv3): python tools/radii_one.py R"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / 'tests'))
from _inputs import plume_inputs                      # noqa: E402
from hallthrusterpem_amd import _lib, constants       # noqa: E402

lib = _lib.load()
R = int(sys.argv[1])
n = int(1.8e9 / (91 * R * 8))
x = {k: torch.as_tensor(v).cuda() for k, v in plume_inputs(n, seed=3).items()}
radii = np.linspace(0.5, 1.5, R)
sets = [(torch.empty((n, 91, R), dtype=torch.float64, device='cuda'), torch.empty((n, R), dtype=torch.float64, device='cuda'),
         torch.empty((n, R), dtype=torch.float64, device='cuda'), torch.empty(n, dtype=torch.uint8, device='cuda')) for _ in range(3)]
p = lambda t: C.c_void_p(t.data_ptr())            # noqa: E731
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for i in range(6):
    j, d, tc, inv = sets[i % 3]
    _lib.check(lib.pem_plume_f64_dev(n, R, C.c_void_p(radii.ctypes.data), constants.TORR_2_PA, *[p(x[k]) for k in
               ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex', 'I_B0', 'T')], p(j), p(d), p(tc), p(inv), st))
torch.cuda.synchronize()
print('done', R, n)
