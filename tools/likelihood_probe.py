#!/usr/bin/env python3
"""Throughput of the jion likelihood kernel against its HBM roofline (one pass over j_ion, 728 B per sample)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd import drivers
from hallthrusterpem_amd.likelihood import JionLikelihood
n, Ne, Na = 1_250_000, 8, 40
res = drivers.forward_uq(n, seed=2, keep_profile=True)
j = res['j_ion']
rng = np.random.default_rng(0)
alpha = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, (Ne, Na)), axis=1)
lk = JionLikelihood(alpha, np.ones((Ne, Na)), np.ones((Ne, Na)))
for _ in range(3): lk.per_sample(j)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): lk.per_sample(j)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(f'jion_loglik: n={n} Ne={Ne} Na={Na}: {ms*1e3:.1f} us per call, {n*736/ms/1e6:.0f} GB/s algorithmic (736 B/sample), {n/ms/1e3:.0f} M samples/s')

# fused: coupled evaluation + likelihood in one launch, profile never written (152 B per evaluation)
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.sampling import Design
two = CoupledBatch(n, profile=True, thruster_qoi=False)
Design(seed=2).fill(two.inputs)
fused = CoupledBatch(n, profile=False, thruster_qoi=False)
fused.inputs.copy_(two.inputs)
out = torch.empty(n, dtype=torch.float64, device='cuda')
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms2 = t(lambda: (two.run(), lk.per_sample(two.j_ion)))
ms1 = t(lambda: fused.run_loglik(lk, out=out))
ms0 = t(lambda: fused.run())
print(f'coupled -> loglik, two launches: {ms2*1e3:.1f} us; fused pem_coupled_loglik: {ms1*1e3:.1f} us '
      f'({n/ms1/1e6:.2f} G evals/s); coupled without profile: {ms0*1e3:.1f} us')
