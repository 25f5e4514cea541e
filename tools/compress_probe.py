#!/usr/bin/env python3
"""svd_compress_kernel alone (for rocprofv3 counter passes: tools/profile_cmd.sh <tag> tools/compress_probe.py [norm])."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers
from hallthrusterpem_amd.compression import SVDCompression
norm = sys.argv[1] if len(sys.argv) > 1 else 'log10'
n = 1_250_000
j = drivers.forward_uq(n, seed=2, keep_profile=True)['j_ion']
c = SVDCompression(norm=norm, reconstruction_tol=0.01).fit(j[:50_000])
for _ in range(3):
    z = c.compress(j)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    z = c.compress(j)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(f'compress norm={norm} rank={c.rank}: {ms * 1e3:.1f} us, {n * (91 + c.rank) * 8 / ms / 1e6:.0f} GB/s')
