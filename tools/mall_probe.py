#!/usr/bin/env python3
"""Does the 256 MB Infinity Cache help the shard-sized bench?  The bench re-evaluates ONE 1.25e6-sample batch (150 MB of
inputs, 0.94 GB of results) every step; here K batches with their own buffers are evaluated round-robin, so that for K >= 2
a batch's inputs have been evicted by the time it comes round again (K x 1.09 GB between two visits).  Prints the mean
kernel time per launch for K = 1, 2, 4, 8 and for one launch over the K = 8 total (1e7 samples)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.sampling import Design
n = 1_250_000
batches = []
for k in range(8):
    b = CoupledBatch(n, thruster_qoi=False)
    Design(seed=2).fill(b.inputs, first_index=k * n)
    batches.append(b)
def timed(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps
for K in (1, 2, 4, 8):
    def sweep():
        for b in batches[:K]: b.run()
    ms = timed(sweep, 40 // K + 4) / K
    print(f'K = {K} batches round-robin: {ms * 1e3:7.1f} us per 1.25e6-sample launch = {872 * n / ms / 1e6:6.0f} GB/s algorithmic')
del batches
torch.cuda.empty_cache()
big = CoupledBatch(8 * n, thruster_qoi=False)
Design(seed=2).fill(big.inputs)
ms = timed(big.run, 8)
print(f'one launch over 1e7 samples:  {ms * 1e3 / 8:7.1f} us per 1.25e6 samples      = {872 * 8 * n / ms / 1e6:6.0f} GB/s algorithmic')
