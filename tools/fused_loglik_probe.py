import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from hallthrusterpem_amd import _lib
from hallthrusterpem_amd.batch import CoupledBatch
from hallthrusterpem_amd.sampling import Design
from hallthrusterpem_amd.likelihood import JionLikelihood
n, Ne, Na = 1_250_000, 8, int(os.environ.get('NA', 40))
rng = np.random.default_rng(0)
alpha = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, (Ne, Na)), axis=1)
lk = JionLikelihood(alpha, np.ones((Ne, Na)), np.ones((Ne, Na)))
fused = CoupledBatch(n, profile=False, thruster_qoi=False)
Design(seed=2).fill(fused.inputs)
out = torch.empty(n, dtype=torch.float64, device='cuda')
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print(os.environ.get('PEM_HIP_LIB'), 'Na', Na, 'fused %.1f us' % (t(lambda: fused.run_loglik(lk, out=out)) * 1e3), 'no-profile %.1f us' % (t(fused.run) * 1e3))
