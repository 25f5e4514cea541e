#!/usr/bin/env python3
"""The fused Saltelli launch (pem_saltelli_f64_dev / _f32_dev) for BASELINE configs[4]'s 2e7-evaluation design (1 428 572 base
samples x 14): microseconds per launch by model and grid size."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import sampling                       # noqa: E402
from hallthrusterpem_amd.fp32 import saltelli_sums             # noqa: E402

fixed = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6}
pri = dict(sampling.PEM_V0_PRIORS)
for k, v in fixed.items():
    pri[k] = sampling.Prior(sampling.UNIFORM, v, v, 'fixed')
design = sampling.Design(priors=pri, seed=1)
varied = [i for i, k in enumerate(design.names) if k not in fixed]
n_base = 1_428_572
for precision in ('fp64', 'fp32'):
    for n_blocks in (256, 512, 768, 1024, 2048):
        for _ in range(2):
            saltelli_sums(design, varied, n_base, n_blocks=n_blocks, precision=precision)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            sums, flags = saltelli_sums(design, varied, n_base, n_blocks=n_blocks, precision=precision)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print(f'{precision} model, {n_blocks:5d} workgroups: {ms:6.3f} ms per {n_base * (len(varied) + 2):.3g} evaluations = {n_base * (len(varied) + 2) / ms / 1e6:6.2f} G evals/s', flush=True)
