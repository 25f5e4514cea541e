#!/usr/bin/env python3
"""Throughput of the SVD compress / reconstruct kernels against their HBM roofline."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd import drivers
from hallthrusterpem_amd.compression import SVDCompression
n = 1_250_000
res = drivers.forward_uq(n, seed=2, keep_profile=True)
j = res['j_ion']
for norm in ('log10', 'none'):
    c = SVDCompression(norm=norm, reconstruction_tol=0.01).fit(j[:50_000])
    def t(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    z = c.compress(j)
    ms_c = t(lambda: c.compress(j)); ms_r = t(lambda: c.reconstruct(z))
    by = n * (91 + c.rank) * 8
    print(f'norm={norm:5s} rank={c.rank}: compress {ms_c*1e3:7.1f} us {by/ms_c/1e6:6.0f} GB/s | reconstruct {ms_r*1e3:7.1f} us {by/ms_r/1e6:6.0f} GB/s  (incl. output alloc)')

# fused: coupled evaluation + compression in one launch (the profile never exists in memory)
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.sampling import Design
c = SVDCompression(norm='log10', reconstruction_tol=0.01).fit(j[:50_000])
two = CoupledBatch(n, profile=True, thruster_qoi=False)
Design(seed=2).fill(two.inputs)
fused = CoupledBatch(n, profile=False, thruster_qoi=False)
fused.inputs.copy_(two.inputs)
out = torch.empty((n, c.rank), dtype=torch.float64, device='cuda')
ms2 = t(lambda: (two.run(), c.compress(two.j_ion)))
ms1 = t(lambda: fused.run_latent(c, out=out))
print(f'coupled -> compress(log10, rank {c.rank}): two launches {ms2*1e3:.1f} us; fused pem_coupled_latent {ms1*1e3:.1f} us '
      f'({n/ms1/1e6:.2f} G evals/s)')
