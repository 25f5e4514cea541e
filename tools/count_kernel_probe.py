#!/usr/bin/env python3
"""Time the evaluate-and-count launch alone (drivers.forward_uq_statistics's dominant kernel) through HIP events around whole
calls minus nothing -- simply the call, many times; use with PEM_HIP_LIB=build_variants/... for A/B.  python tools/count_kernel_probe.py [n] [keep]"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keep = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
for _ in range(4):
    r = drivers.forward_uq_statistics(n, seed=2, keep_profile=keep)
torch.cuda.synchronize()
best = 1e9
for _ in range(8):
    t0 = time.perf_counter(); r = drivers.forward_uq_statistics(n, seed=2, keep_profile=keep); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(os.environ.get('PEM_HIP_LIB', 'tree'), f'keep={keep} fused={r["fused"]} whole call best {best * 1e3:.2f} ms')
