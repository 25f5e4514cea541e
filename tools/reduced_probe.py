#!/usr/bin/env python3
"""A few launches of the profile-less coupled kernel (reduced-QoI mode) -- a target for tools/profile_cmd.sh."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.sampling import Design
b = CoupledBatch(1_250_000, profile=False, thruster_qoi=False)
Design(seed=2).fill(b.inputs)
for _ in range(12): b.run()
torch.cuda.synchronize()
