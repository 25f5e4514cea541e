#!/usr/bin/env python3
"""drivers.forward_uq_statistics under rocprofv3 (tools/kstats.sh): python tools/fused_probe.py [n] [keep_profile 0/1] [reps]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keep = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for _ in range(3):
    r = drivers.forward_uq_statistics(n, seed=2, keep_profile=keep)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    r = drivers.forward_uq_statistics(n, seed=2, keep_profile=keep)
torch.cuda.synchronize()
print(f'forward_uq_statistics n={n} keep_profile={keep} fused={r["fused"]}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms')
