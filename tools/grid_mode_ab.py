#!/usr/bin/env python3
"""Interleaved A/B of the coupled kernel's grid modes on ONE box, in ONE process, streaming regime (8 batches in rotation):
the balanced persistent grid (default), the full persistent grid, a one-shot grid (one tile per wave: PEM_GRID_MULT=0) and
grids of m x the resident slots.  The library reads PEM_GRID_MULT / PEM_WAVES_PER_CU at every launch."""
import argparse
import os
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd.batch import CoupledBatch        # noqa: E402
from hallthrusterpem_amd.sampling import Design           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--modes', default='default,PEM_GRID_MULT=1,PEM_GRID_MULT=0,PEM_GRID_MULT=2,PEM_GRID_MULT=4')
ap.add_argument('--rounds', type=int, default=6)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--layout', default='soa')
ap.add_argument('--streams', default='1,2')
ap.add_argument('--libs', default='', help='name=path,... : variant libraries (tools/build_variant.sh); a mode "lib:name" launches through one')
args = ap.parse_args()
N = 1_250_000
batches = []
for k in range(8):
    b = CoupledBatch(N, thruster_qoi=False, layout=args.layout)
    tmp = Design(seed=2).sample(N, first_index=k * N)
    b.load_soa(tmp)
    del tmp
    batches.append(b)
side = {1: [torch.cuda.current_stream()], 2: [torch.cuda.Stream(), torch.cuda.Stream()], 3: [torch.cuda.Stream() for _ in range(3)], 4: [torch.cuda.Stream() for _ in range(4)]}
NS = [int(v) for v in args.streams.split(',')]
modes = args.modes.split(',')
res = {(m, s): [] for m in modes for s in NS}
KEYS = ('PEM_GRID_MULT', 'PEM_WAVES_PER_CU', 'PEM_TAIL_TILES')


import ctypes as C                                         # noqa: E402
from hallthrusterpem_amd import _lib as L                  # noqa: E402
main_lib = L.load()
variants = {}
for spec in filter(None, args.libs.split(',')):
    name, path = spec.split('=')
    h = C.CDLL(str(Path(path).resolve()))
    for fn, (restype_, argtypes_) in L.SIGNATURES.items():
        getattr(h, fn).restype = restype_
        getattr(h, fn).argtypes = argtypes_
    variants[name] = h


def set_mode(m):
    for k in KEYS:
        os.environ.pop(k, None)
    L._lib = main_lib
    if m != 'default':
        for kv in m.split('+'):
            if kv.startswith('lib:'):
                L._lib = variants[kv[4:]]
                continue
            k, v = kv.split('=')
            os.environ[k] = v


for r in range(args.rounds):
    for m in modes:
        set_mode(m)
        for ns in NS:
            cnt = 0

            def sweep():
                global cnt
                for b in batches:
                    b.run(stream=side[ns][cnt % ns])
                    cnt += 1
            sweep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                sweep()
            torch.cuda.synchronize()
            res[(m, ns)].append((time.perf_counter() - t0) / (args.reps * 8) * 1e6)
set_mode('default')
print(f'# us per 1.25e6-sample launch (872 B per sample), layout {args.layout}: median [min .. max] over {args.rounds} interleaved rounds of {args.reps * 8} launches')
for m in modes:
    for ns in NS:
        v = sorted(res[(m, ns)])
        med = v[len(v) // 2]
        print(f'{m:40s} streams {ns}: {med:7.1f} [{v[0]:6.1f} .. {v[-1]:6.1f}] us = {872 * N / med / 1e6:5.2f} TB/s = {872 * N / med / 8e6:5.3f} of peak')
