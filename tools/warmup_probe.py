"""How long after process start does a launch of the bench kernel reach its steady duration?  The driver runs
`bench.py --steps 20 --warmup 5`: 25 launches, 5 ms of GPU work, after a start-up that is mostly host work.  This probe sets
the batches up as bench.py does and times EVERY launch of a fresh process with HIP events (one stream), printing the mean of
consecutive blocks, then the same after one second of idling, to tell first-touch effects (once per buffer) from clock ramps
(again after every idle period).
    python tools/warmup_probe.py [--launches 400] [--block 20]
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=400)
    ap.add_argument('--block', type=int, default=20)
    ap.add_argument('--n', type=int, default=1_250_000)
    args = ap.parse_args()
    import torch
    import bench
    from hallthrusterpem_amd.batch import CoupledBatch
    batches = []
    for k in range(8):
        b = CoupledBatch(args.n, profile=True, thruster_qoi=False, layout='tile')
        bench.synth_inputs(b, 0, 0, k)
        batches.append(b)
    torch.cuda.synchronize()

    def series(tag):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.launches)]
        t0 = time.perf_counter()
        for i, (a, b) in enumerate(evs):
            a.record()
            batches[i % 8].run()
            b.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ms = [a.elapsed_time(b) for a, b in evs]
        print(f'## {tag}: {args.launches} launches, wall {1e3 * wall / args.launches:.4f} ms per launch')
        for s in range(0, args.launches, args.block):
            blk = ms[s:s + args.block]
            print(f'launches {s:4d}..{s + len(blk) - 1:4d}: mean {1e3 * sum(blk) / len(blk):7.1f} us  min {1e3 * min(blk):7.1f}  max {1e3 * max(blk):7.1f}')

    series('fresh process (first touch of the result buffers in launches 0..7)')
    time.sleep(1.0)
    series('after 1 s idle')
    time.sleep(0.05)
    series('after 50 ms idle')


if __name__ == '__main__':
    main()
