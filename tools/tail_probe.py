#!/usr/bin/env python3
"""Launch-boundary cost of the coupled kernel in the streaming regime (8 batches in rotation): samples per launch chosen so
that the persistent grid (2048 waves of 64-sample tiles) gets exactly 9, 9.54 (the 1.25e6-sample shard) or 10 tiles per
wave, and 2 / 4 shards per launch."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.sampling import Design
for n in (9 * 2048 * 64, 1_250_000, 10 * 2048 * 64, 2_500_000, 5_000_000):
    K = max(2, 10_000_000 // n)
    batches = []
    for k in range(K):
        b = CoupledBatch(n, thruster_qoi=False)
        Design(seed=2).fill(b.inputs, first_index=k * n)
        batches.append(b)
    def sweep():
        for b in batches: b.run()
    for _ in range(2): sweep()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 6
    a.record()
    for _ in range(reps): sweep()
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / (reps * K)
    print(f'n = {n:8d} ({n / (2048 * 64):5.2f} tiles per wave), {K} batches in rotation: {ms * 1e3:7.1f} us per launch, {ms * 1e3 / n * 1.25e6:6.1f} us per 1.25e6 samples, '
          f'{872 * n / ms / 1e6:5.0f} GB/s')
    del batches, b
    torch.cuda.empty_cache()
