#!/usr/bin/env python3
"""How much of the short kernels' distance to the HBM roofline is launch ramp and tail: the reduced-QoI kernels (fp64: 144 B per
evaluation, VALU-bound; fp32: 72 B, HBM-bound by design) and the mixed-precision mode at 1.25e6 samples per launch (22-135 us) and at
1e7 (eight times the work per launch), isolated launches and launches dealt onto two streams; buffers in rotation throughout."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd.batch import CoupledBatch          # noqa: E402
from hallthrusterpem_amd.fp32 import CoupledBatchF32        # noqa: E402
from hallthrusterpem_amd.sampling import Design             # noqa: E402

side = [torch.cuda.Stream(), torch.cuda.Stream()]
for name, make, bpe in (('reduced QoIs, fp64 arithmetic (144 B/eval)', lambda n: CoupledBatch(n, profile=False, thruster_qoi=False), 144),
                        ('reduced QoIs, fp32 arithmetic (72 B/eval)', lambda n: CoupledBatchF32(n), 72),
                        ('mixed: fp64 arithmetic, fp32 profile (508 B/eval)', lambda n: CoupledBatch(n, profile=True, mixed=True, thruster_qoi=False), 508)):
    for n, nb in ((1_250_000, 8), (10_000_000, 4)):
        bs = [make(n) for _ in range(nb)]
        src = CoupledBatch(n, profile=False, thruster_qoi=False)
        for i, b in enumerate(bs):
            Design(seed=2 + i).fill(src.inputs)
            b.inputs.copy_(src.inputs)
        del src
        out = []
        for ns in (1, 2):
            reps = 25 * nb
            for i in range(nb):
                bs[i].run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(reps):
                if ns == 1:
                    bs[i % nb].run()
                else:
                    bs[i % nb].run(stream=side[i % 2])
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
            out.append(f'{ns} stream{"s" if ns > 1 else ""}: {us:8.1f} us = {n * bpe / us / 1e6:5.2f} TB/s, {n / us / 1e3:6.2f} G evals/s')
        print(f'{name}, n = {n:8d}: ' + ' | '.join(out), flush=True)
        del bs
        torch.cuda.empty_cache()
