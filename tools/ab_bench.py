#!/usr/bin/env python3
"""Interleaved A/B timing of libpem_hip.so variants in ONE process (cdna_hip_programming.md rule 24).

    python tools/ab_bench.py [--n 1250000] [--rounds 15] [--reps 8] name=path[:lanes[:waves_per_cu]] ...

Each variant is a shared library (tools/build_variant.sh) plus optional lanes-per-sample / waves-per-CU settings.
Every round times `reps` back-to-back coupled launches of every variant with HIP events; prints median and min."""
import argparse
import ctypes as C
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=1_250_000)
    ap.add_argument('--rounds', type=int, default=15)
    ap.add_argument('--reps', type=int, default=8)
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-invalid', action='store_true', help='pass NULL for the invalid-flag array')
    ap.add_argument('variants', nargs='+')
    args = ap.parse_args()

    import torch
    from hallthrusterpem_amd import _lib
    from hallthrusterpem_amd.batch import CoupledBatch
    import bench
    _lib.load()                     # maps torch's HIP runtime first; every variant then shares it
    _lib.require_device()
    batch = CoupledBatch(args.n, profile=not args.no_profile, thruster_qoi=False)
    if args.no_invalid:
        batch._out_ptrs[-1] = None
    bench.synth_inputs(batch, 2, 0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    variants = []
    for spec in args.variants:
        name, rest = spec.split('=', 1)
        parts = rest.split(':')
        lib = C.CDLL(str((ROOT / parts[0]).resolve()))
        for fn, (res, argt) in _lib.SIGNATURES.items():
            getattr(lib, fn).restype = res
            getattr(lib, fn).argtypes = argt
        lanes = int(parts[1]) if len(parts) > 1 and parts[1] else 0
        wpc = parts[2] if len(parts) > 2 else ''
        variants.append((name, lib, lanes, wpc))

    def launch(lib, lanes, wpc):
        if wpc:
            os.environ['PEM_WAVES_PER_CU'] = wpc
        else:
            os.environ.pop('PEM_WAVES_PER_CU', None)
        lib.pem_set_lanes_per_sample(lanes)
        rc = lib.pem_coupled_f64_dev(batch.n, 133.322, 1.0, *batch._in_ptrs, *batch._out_ptrs, stream)
        assert rc == 0, lib.pem_last_error()

    times = {v[0]: [] for v in variants}
    for name, lib, lanes, wpc in variants:          # warm-up
        for _ in range(3):
            launch(lib, lanes, wpc)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for name, lib, lanes, wpc in variants:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                launch(lib, lanes, wpc)
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b) / args.reps)
    bytes_per = (144 if args.no_profile else 872) * args.n
    for name, ts in times.items():
        med, mn = statistics.median(ts), min(ts)
        print(f'{name:24s} median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us   {bytes_per / med / 1e6:7.0f} GB/s (median)')


if __name__ == '__main__':
    main()
