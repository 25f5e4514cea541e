#!/usr/bin/env python3
"""Time pem_coupled_latent_f64_dev (set PEM_HIP_LIB to compare build variants)."""
import os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.compression import SVDCompression
from hallthrusterpem_amd.sampling import Design
n, rank = 1_250_000, 6
c = SVDCompression(norm=os.environ.get('NORM', 'log10'), rank=rank)
c.basis = torch.from_numpy(np.linalg.qr(np.random.default_rng(0).standard_normal((91, rank)))[0].copy()).cuda()
b = CoupledBatch(n, profile=False, thruster_qoi=False)
Design(seed=2).fill(b.inputs)
out = torch.empty((n, rank), dtype=torch.float64, device='cuda')
for _ in range(3): b.run_latent(c, out=out)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): b.run_latent(c, out=out)
e.record(); torch.cuda.synchronize()
print(os.environ.get('PEM_HIP_LIB'), os.environ.get('NORM', 'log10'), f'{a.elapsed_time(e) / 20 * 1e3:.1f} us')
