#!/usr/bin/env python3
"""Launch-bound loops under a hipGraph: (a) BASELINE configs[0] -- 1e4-sample LHS design + cathode_coupling per
iteration; (b) one posterior evaluation and (c) one Metropolis step of hallthrusterpem_amd.calibration.
Eager launches vs graph replay, microseconds per iteration."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import ctypes as C
from hallthrusterpem_amd import _lib, constants
from hallthrusterpem_amd.calibration import JionPosterior, Metropolis, capture_graph
from hallthrusterpem_amd.sampling import Design, PEM_V0_PRIORS


def wall(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


# (a) config 1: LHS over the six cathode inputs + one cathode launch
names = ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T')
design = Design(names=names, seed=0)
n = 10_000
x = torch.empty((6, n), dtype=torch.float64, device='cuda')
v = torch.empty(n, dtype=torch.float64, device='cuda')
lib = _lib.load()
p = lambda t: C.c_void_p(t.data_ptr())
def cathode_iter():
    design.fill(x, method='lhs', n_total=n)
    _lib.check(lib.pem_cathode_f64_dev(n, *[p(x[i]) for i in range(6)], constants.TORR_2_PA, p(v),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
eager = wall(cathode_iter, 2000)
g, _ = capture_graph(cathode_iter, torch.device('cuda', 0))
graph = wall(g.replay, 2000)
print(f'config 1 (1e4 LHS + cathode_coupling): eager {eager:.1f} us/iter, hipGraph replay {graph:.1f} us/iter '
      f'({n / graph:.0f} M evals/s)')

# (b), (c) calibration: 64 chains x 100 nuisance draws x 8 conditions = 51200 samples per step
rng = np.random.default_rng(0)
K, M, Ne, Na = 64, 100, 8, 40
operating = np.stack([10.0 ** rng.uniform(-6, -4.5, Ne), rng.uniform(250, 350, Ne), rng.uniform(4e-6, 6e-6, Ne)], axis=1)
alpha = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, (Ne, Na)), axis=1)
y = rng.lognormal(0, 1, (Ne, Na))
post = JionPosterior(('c0', 'c1', 'c2', 'c3', 'c4', 'c5'), operating, alpha, y, 0.2 * y + 0.05, n_chains=K, n_nuisance=M,
                     fresh_nuisance=False)
theta = torch.tensor([[0.4, 0.4, 3.0, 0.7, 1e20, 1e16]], dtype=torch.float64, device='cuda').expand(K, 6).contiguous()
eager = wall(lambda: post.log_posterior(theta), 300)
replay = post.capture()
graph = wall(lambda: replay(theta), 300)
print(f'posterior evaluation ({K} chains x {M} draws x {Ne} conditions = {post.n} samples): eager {eager:.1f} us, '
      f'hipGraph replay {graph:.1f} us ({post.n / graph:.0f} M evals/s)')
scale = [0.02, 0.02, 0.5, 0.03, 0.0, 0.0]
for use_graph in (False, True):
    mh = Metropolis(post, theta[0].cpu().numpy(), scale, seed=1, use_graph=use_graph)
    us = wall(lambda: mh.run(1, keep=False), 300)
    print(f'Metropolis step, {"hipGraph replay" if use_graph else "eager"}: {us:.1f} us per step of {K} chains')
