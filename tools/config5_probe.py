#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU: Saltelli design of 2e7 coupled evaluations (12 varied inputs, operating point
held: N_base (d + 2) = 1,428,572 x 14), first-order + total Sobol' indices incl. the thruster QoI post-process, and the
fp64 -> mixed (fp32 profile) tolerance check on identical inputs."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd import drivers
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.models.thruster import check_thruster_outputs
from hallthrusterpem_amd.sampling import Design
fixed = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6}
n_base = 1_428_572
drivers.sobol_indices(10_000, seed=1, fixed=fixed)                      # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
res = drivers.sobol_indices(n_base, seed=1, fixed=fixed, batch_size=1 << 21)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'Saltelli: {res["evaluations"]} coupled evaluations in {dt*1e3:.1f} ms wall = {res["evaluations"]/dt/1e9:.2f} G evals/s '
      f'(sampling + evaluation + reductions, reduced-QoI mode, 1 GPU)')
for q in ('V_cc', 'div_angle', 'T_c'):
    top = sorted(zip(res['inputs'], res['ST'][q].tolist()), key=lambda kv: -kv[1])[:3]
    print(f'  {q}: sum S1 = {float(res["S1"][q].sum()):.3f}; largest total indices: ' + ', '.join(f'{k} {v:.3f}' for k, v in top))
# thruster QoI post-process on a batch of the design (thruster.py:490-502 filters)
b = CoupledBatch(2_000_000, profile=False)
Design(seed=1).fill(b.inputs)
b.run()
bad = check_thruster_outputs({'T': b.T, 'I_B0': b.I_B0})
print(f'thruster filters on 2e6 samples: {int(bad.sum())} non-physical')
# fp64 vs mixed on identical inputs
n = 2_000_000
f64, mix = CoupledBatch(n), CoupledBatch(n, mixed=True)
Design(seed=2).fill(f64.inputs); mix.inputs.copy_(f64.inputs)
f64.run(); mix.run(); torch.cuda.synchronize()
rel = ((mix.j_ion.double() - f64.j_ion) / f64.j_ion).abs().flatten()
k = int(0.999 * rel.numel())
print(f'mixed vs fp64 on {n} identical samples: scalars bit-identical = {bool(torch.equal(f64.qoi, mix.qoi))}; '
      f'j_ion relative error max {float(rel.max()):.3e}, 99.9th percentile {float(rel.kthvalue(k).values):.3e} (fp32 half-ulp = 5.96e-08)')
