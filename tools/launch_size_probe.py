#!/usr/bin/env python3
"""What a range launch of the coupled kernel costs as a function of its size and grid (round 3; the N > 1 pipeline cuts a
shard into such launches).  Streaming regime: 8 batches of 1.25e6 samples in rotation, every launch evaluates samples
[0, n) of the next batch.  Run twice: PEM_BALANCED_GRID=1 (default) and =0 (always the full grid).

    python tools/launch_size_probe.py [--streams 2]
"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import _lib                      # noqa: E402
from hallthrusterpem_amd.batch import CoupledBatch        # noqa: E402
from hallthrusterpem_amd.sampling import Design           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, default=1)
ap.add_argument('--reps', type=int, default=25)
ap.add_argument('--sizes', type=str, default='')
ap.add_argument('--walk', action='store_true', help='the launches of size n walk through the whole 1.25e6-sample batch (a chunked step) instead of re-evaluating its first n samples')
args = ap.parse_args()
N = 1_250_000
batches = []
for k in range(8):
    b = CoupledBatch(N, thruster_qoi=False)
    Design(seed=2).fill(b.inputs, first_index=k * N)
    batches.append(b)
cus, per = _lib.coupled_occupancy(1)
full = cus * per * 4 * 64
bal = _lib.persistent_grid(N, cus, per)[1]
print(f'# PEM_GRID_MULT={os.environ.get("PEM_GRID_MULT")} PEM_BALANCED_GRID={os.environ.get("PEM_BALANCED_GRID", "1")}, {cus} CUs x {per} workgroups, full round {full} samples, '
      f'balanced round of the shard {bal}; streams {args.streams}')
side = [torch.cuda.Stream() for _ in range(args.streams)]
sizes = [bal * r for r in (1, 2, 3, 5)] + [full * r for r in (1, 2, 3, 5)] + [312512, 625920, 624080, 625024, 1_000_000, N]
if args.sizes:
    sizes = [int(v) for v in args.sizes.split(',')]
for n in sizes:
    grid, spr = _lib.persistent_grid(n, cus, per)
    cnt = [0]

    def sweep():
        for b in batches:
            for first in (range(0, N, n) if args.walk else (0,)):
                st = side[cnt[0] % len(side)]
                cnt[0] += 1
                b.run(first=first, count=min(n, N - first), stream=st)
    for _ in range(2):
        sweep()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(args.reps):
        sweep()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / (args.reps * 8) * 1e6
    if args.walk:
        print(f'walk: pieces of {n:8d} samples ({-(-N // n):3d} launches per 1.25e6-sample step): {us:7.1f} us per step, {872 * N / us / 1e6:5.2f} TB/s')
        continue
    print(f'n = {n:8d} = {n / spr:5.2f} rounds of a {grid:3d}-workgroup grid: {us:7.1f} us per launch, {us / n * N:6.1f} us per 1.25e6 samples, '
          f'{872 * n / us / 1e6:5.2f} TB/s')
