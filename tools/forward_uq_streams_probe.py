"""drivers.forward_uq (fused Monte-Carlo launches of 2^21 samples) with its launches on one stream or dealt onto two: wall time per\ncampaign of 1e7 / 5e6 samples, with and without the profile (profiles/launch_amortisation_r03.txt)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from hallthrusterpem_amd import drivers
for n in (10_000_000, 5_000_000):
    for keep in (False, True):
        for streams in (1, 2, 1, 2):
            for _ in range(2):
                out = drivers.forward_uq(n, seed=2, keep_profile=keep, streams=streams)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = drivers.forward_uq(n, seed=2, keep_profile=keep, streams=streams)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            print(f'forward_uq n={n} keep_profile={keep} streams={streams}: {ms:.3f} ms = {n / ms / 1e6:.2f} G evals/s', flush=True)
            del out
            torch.cuda.empty_cache()
