#!/usr/bin/env python3
"""BASELINE configs[4] in single precision: the model's tolerance report and the fused Saltelli launch against the fp64
block-by-block driver (same design).  Prints one JSON document (committed as profiles/fp32_report_<round>.json)."""
import json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers
from hallthrusterpem_amd.fp32 import compare_with_fp64
rep = {'model': compare_with_fp64(1_250_000)}
fixed = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6}
n_base = 1_428_572
out = {}
for name, kw in (('fp64_block_by_block', dict(fused=False)), ('fp64', {}), ('fp32', dict(precision='fp32'))):
    drivers.sobol_indices(10_000, seed=1, fixed=fixed, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        res = drivers.sobol_indices(n_base, seed=1, fixed=fixed, batch_size=1 << 21, **kw)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out[name] = res
    rep[f'saltelli_{name}'] = {'evaluations': res['evaluations'], 'wall_ms': 1e3 * best, 'evals_per_s': res['evaluations'] / best}
rep['fused_fp64_vs_block_by_block'] = rep['saltelli_fp64_block_by_block']['wall_ms'] / rep['saltelli_fp64']['wall_ms']
rep['saltelli_speedup'] = rep['saltelli_fp64']['wall_ms'] / rep['saltelli_fp32']['wall_ms']
rep['largest_index_difference'] = max(float((out['fp64'][k][q] - out['fp32'][k][q]).abs().max()) for k in ('S1', 'ST') for q in ('V_cc', 'div_angle', 'T_c'))
rep['non_physical'], rep['invalid'] = out['fp32']['non_physical'], out['fp32']['invalid']
print(json.dumps(rep, indent=1))
