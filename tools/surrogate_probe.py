#!/usr/bin/env python3
"""BASELINE.json configs[3] shape: adaptive surrogate training, then 5e5 candidate points through the true model and
through the batched sparse-grid predict kernel."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd.surrogate import SparseGridSurrogate
from hallthrusterpem_amd.batch import CoupledBatch
FIXED = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6, 'a_1': 0.01, 'sigma_cex': 55e-20, 'c4': 1e20, 'c5': 1e16}
VARIED = ('T_e', 'V_vac', 'Pstar', 'P_T', 'c0', 'c1', 'c2', 'c3')
s = SparseGridSurrogate(VARIED, FIXED)
t0 = time.perf_counter()
hist = s.refine(max_iter=40, num_refine=1000, seed=0)
torch.cuda.synchronize()
print(f'adaptive fit: 40 iterations, {len(s.index_set)} active indices, {len(s.candidates)} candidates, {s.model_evals} true-model '
      f'evaluations, {time.perf_counter() - t0:.2f} s wall; last indicator {hist[-1][1]:.2e}')
n = 500_000
g = torch.Generator(device='cuda'); g.manual_seed(1)
t = torch.rand((len(VARIED), n), dtype=torch.float64, device='cuda', generator=g) * 2 - 1
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms_p = timeit(lambda: s.predict(t))
x = {k: np.full(n, v) for k, v in FIXED.items()}
x.update(s.to_physical(t.cpu().numpy()))
batch = CoupledBatch(n, profile=False)
batch.set_inputs(x)
ms_m = timeit(batch.run)
truth = batch.qoi.clone()
pred = s.predict(t)
err = (torch.linalg.norm(pred - truth, dim=1) / torch.linalg.norm(truth, dim=1)).cpu().numpy()
nb = s._tables[3]
nodes_total = int(s._tables[2].numel() // len(s.qoi))
levels = s._tables[0].cpu().numpy()[:, 2 + 3:2 + 6]
print(f'combination: {nb} grids, {nodes_total} grid nodes in total (sum of the grids\' sizes) -> {nodes_total * len(s.qoi)} FMAs per point; level histogram of active dims {np.bincount(levels[levels > 0].ravel())}')
print(f'5e5 candidate points: true model (reduced QoIs) {ms_m*1e3:.1f} us = {n/ms_m/1e6:.2f} G evals/s | surrogate predict '
      f'({nb} grids in the combination) {ms_p*1e3:.1f} us = {n/ms_p/1e6:.2f} G points/s | relative L2 error {err}')
