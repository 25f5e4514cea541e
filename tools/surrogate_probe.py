#!/usr/bin/env python3
"""Adaptive sparse-grid training of scalars + j_ion latents (BASELINE configs[3]): error against the true model and time per
refinement step as the index set grows.  python tools/surrogate_probe.py [iters] [num_refine] [max_level] [max_active]"""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.surrogate import SparseGridSurrogate

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
num_refine = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
max_level = int(sys.argv[3]) if len(sys.argv) > 3 else 4
max_active = int(sys.argv[4]) if len(sys.argv) > 4 else 5
fixed = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6, 'a_1': 0.01, 'sigma_cex': 55e-20, 'c4': 1e20, 'c5': 1e16}
varied = ('T_e', 'V_vac', 'Pstar', 'P_T', 'c0', 'c1', 'c2', 'c3')
t0 = time.perf_counter()
s = SparseGridSurrogate(varied, fixed, qoi=('V_cc', 'div_angle', 'T_c', 'j_ion'), max_level=max_level, max_active=max_active)
print(f'compression rank {s.compression.rank} (relative error {s.compression.relative_error:.4f}), n_out {s.n_out}, set-up {time.perf_counter() - t0:.2f} s')
n = 500_000
g = torch.Generator(device='cuda'); g.manual_seed(1)
t = torch.rand((len(varied), n), dtype=torch.float64, device='cuda', generator=g) * 2 - 1
x = {k: np.full(n, v) for k, v in fixed.items()}
x.update(s.to_physical(t.cpu().numpy()))
b = CoupledBatch(n, profile=True)
b.set_inputs(x); b.run(); torch.cuda.synchronize()
lt = torch.log10(b.j_ion)
done = 0
for step in (10, 20, 40, 80, 120, 160, 200, 300, 400):
    if step > iters:
        break
    t1 = time.perf_counter()
    s.refine(max_iter=step - done, num_refine=num_refine, seed=done)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    done = step
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); y = s.predict_fields(t); e1.record(); torch.cuda.synchronize()
    errs = {k: float(torch.linalg.norm(y[k] - b.qoi[i]) / torch.linalg.norm(b.qoi[i])) for i, k in enumerate(('V_cc', 'div_angle', 'T_c'))}
    ej = float(torch.linalg.norm(torch.log10(y['j_ion']) - lt) / torch.linalg.norm(lt))
    na = max(sum(1 for l in bb if l > 0) for bb in s.index_set); lv = max(max(bb) for bb in s.index_set)
    print(f'{step:4d} iterations ({dt:6.2f} s, {s.model_evals} model evaluations, {len(s.index_set)} indices, <= {na} active dims, level <= {lv}): '
          f'V_cc {errs["V_cc"]:.2e} div_angle {errs["div_angle"]:.2e} T_c {errs["T_c"]:.2e} log10 j_ion {ej:.2e} | predict+reconstruct 5e5 points {e0.elapsed_time(e1):.2f} ms')
