#!/usr/bin/env python3
"""Fused Monte-Carlo mode (inputs generated in-kernel) against sample-then-evaluate, per 1.25e6-sample step."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.sampling import Design
n = 1_250_000
d = Design(seed=2)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for prof in (True, False):
    b = CoupledBatch(n, profile=prof, thruster_qoi=False)
    def two():
        d.fill(b.inputs); b.run()
    ms2 = t(two); ms_s = t(lambda: d.fill(b.inputs)); ms1 = t(lambda: b.run_mc(d)); ms1w = t(lambda: b.run_mc(d, write_inputs=True))
    print(f'profile={prof}: sample {ms_s*1e3:.1f} us + evaluate = {ms2*1e3:.1f} us | fused {ms1*1e3:.1f} us ({n/ms1/1e6:.2f} G evals/s) | fused + inputs written {ms1w*1e3:.1f} us')
