#!/usr/bin/env python3
"""Kernels whose README / DESIGN rates were first measured on ONE buffer set, again with eight sets in rotation (no help from the
256 MB Infinity Cache): the jion likelihood kernel (one pass over 0.92 GB of profiles), the fp32 and fp64 reduced-QoI kernels
and the mixed-precision profile mode."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd import drivers
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.fp32 import CoupledBatchF32
from hallthrusterpem_amd.likelihood import JionLikelihood
from hallthrusterpem_amd.sampling import Design
n, NB = 1_250_000, 8


def t(fn, reps=40):
    for i in range(NB): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


js = [drivers.forward_uq(n, seed=2 + i, keep_profile=True)['j_ion'] for i in range(NB)]
rng = np.random.default_rng(0)
alpha = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, (8, 40)), axis=1)
lk = JionLikelihood(alpha, np.ones((8, 40)), np.ones((8, 40)))
rot, one = t(lambda i: lk.per_sample(js[i % NB])), t(lambda i: lk.per_sample(js[0]))
print(f'jion likelihood kernel (736 B/sample): {rot:.1f} us = {n*736/rot/1e6:.2f} TB/s in rotation | {one:.1f} us = {n*736/one/1e6:.2f} TB/s on one buffer')
del js
d = Design(seed=2)
for name, make, bpe in (('reduced QoIs, fp64 (144 B/eval)', lambda: CoupledBatch(n, profile=False, thruster_qoi=False), 144),
                        ('reduced QoIs, fp32 (72 B/eval)', lambda: CoupledBatchF32(n), 72),
                        ('mixed: fp64 arithmetic, fp32 profile (508 B/eval)', lambda: CoupledBatch(n, profile=True, mixed=True, thruster_qoi=False), 508)):
    bs = [make() for _ in range(NB)]
    src = CoupledBatch(n, profile=False, thruster_qoi=False)
    for i, b in enumerate(bs):
        Design(seed=2 + i).fill(src.inputs)
        b.inputs.copy_(src.inputs)
    rot, one = t(lambda i: bs[i % NB].run()), t(lambda i: bs[0].run())
    print(f'{name}: {rot:.1f} us = {n*bpe/rot/1e6:.2f} TB/s in rotation | {one:.1f} us = {n*bpe/one/1e6:.2f} TB/s on one batch')
    del bs
