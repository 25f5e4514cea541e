#!/usr/bin/env python3
"""Where the HOST time of drivers.forward_uq_statistics goes (cProfile, 30 calls): python tools/fused_pyprofile.py [n] [keep]"""
import cProfile, pstats, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
keep = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
for _ in range(3):
    drivers.forward_uq_statistics(n, seed=2, keep_profile=keep)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    drivers.forward_uq_statistics(n, seed=2, keep_profile=keep)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
