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
