#!/usr/bin/env python3
"""PCIe-inclusive rate of the HOST-pointer entry points (numpy in, numpy out), for DESIGN.md section 6."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from _inputs import coupled_inputs, cathode_inputs, plume_inputs
from hallthrusterpem_amd.models import pem_v0_coupled, cathode_coupling, current_density
n = 1_250_000
x = coupled_inputs(n, seed=2)
pem_v0_coupled({k: v[:1000] for k, v in x.items()})
for prof in (True, False):
    best = min((lambda t0: (pem_v0_coupled(x, profile=prof), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(3))
    print(f'pem_v0_coupled host path profile={prof}: n={n} {best*1e3:.1f} ms  {n/best/1e6:.2f} M evals/s  {((872 if prof else 144)*n)/best/1e9:.2f} GB/s algorithmic')
c = cathode_inputs(10_000, seed=0, lhs=True)
cathode_coupling(c)
t0 = time.perf_counter()
for _ in range(100): cathode_coupling(c)
print(f'cathode_coupling host path config 1 (1e4 LHS): {(time.perf_counter()-t0)/100*1e6:.1f} us per call')
p = plume_inputs(1_000_000, seed=1)
current_density({k: v[:1000] for k, v in p.items()})
best = min((lambda t0: (current_density(p), time.perf_counter() - t0)[1])(time.perf_counter()) for _ in range(3))
print(f'current_density host path config 2 (1e6): {best*1e3:.1f} ms  {1e6/best/1e6:.2f} M evals/s')
import torch
cd = {k: torch.from_numpy(v).cuda() for k, v in c.items()}
cathode_coupling(cd); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200): cathode_coupling(cd)
b.record(); torch.cuda.synchronize()
print(f'cathode_coupling device path config 1: {a.elapsed_time(b)/200*1e3:.1f} us per call (launch-latency-bound)')
pd = {k: torch.from_numpy(v).cuda() for k, v in p.items()}
current_density(pd); torch.cuda.synchronize()
a.record()
for _ in range(20): current_density(pd)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)/20
print(f'current_density device path config 2 (1e6, with T): {ms*1e3:.1f} us per call  {1e6/ms/1e3:.0f} M evals/s  {824e6/ms/1e6:.0f} GB/s algorithmic (incl. wrapper allocs)')
