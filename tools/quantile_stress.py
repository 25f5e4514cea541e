#!/usr/bin/env python3
"""Randomised stress of pem_quantiles_f64_dev against np.percentile (bit for bit): random shapes, percentile sets, pilot strides and
thresholds (PEM_QUANTILE_PILOT / PEM_QUANTILE_PILOT_MIN are read at every call), distributions with ties, constants, NaN, infinities,
signed zeros, sorted and periodic columns.  Also drivers.filter_outputs (masks kernel) against its numpy branch.
    python tools/quantile_stress.py [--cases 400] [--seed 0] [--pilot-only] [--sharded]"""
import argparse, os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import drivers, _lib

ap = argparse.ArgumentParser()
ap.add_argument('--cases', type=int, default=400)
ap.add_argument('--seed', type=int, default=0)
ap.add_argument('--pilot-only', action='store_true', help='every case with a pilot stride and a threshold below its size')
ap.add_argument('--sharded', action='store_true', help='also the sharded selection (pem_qsel_*) on this one rank, every case')
args = ap.parse_args()
lib = _lib.load()
rng = np.random.default_rng(args.seed)
paths = {0: 0, 1: 0, 2: 0}
for case in range(args.cases):
    n = int(rng.choice([1, 2, 3, 17, 129, 1000, 4097, 20_000, 100_003, 300_000]))
    m = int(rng.choice([1, 2, 3, 7, 31, 64, 65, 91, 128, 129, 200, 256, 257, 300]))
    if n * m > 6_000_000:
        n = max(1, 6_000_000 // m)
    kind = rng.integers(0, 6)
    a = rng.lognormal(0.0, float(rng.choice([0.1, 1.0, 3.0])), (n, m)) * np.where(rng.random((n, m)) < rng.choice([0.0, 0.3]), -1.0, 1.0)
    if kind == 1:                                    # heavy ties
        a[rng.random((n, m)) < rng.choice([0.1, 0.5, 0.9])] = rng.choice([1e-20, 0.0, 3.5])
    elif kind == 2:                                  # few distinct values, a constant column, signed zeros
        a = rng.integers(0, int(rng.choice([2, 5, 1000])), (n, m)).astype(np.float64)
        a[:, 0] = 7.25
        a[:, -1] = np.where(rng.random(n) < 0.5, 0.0, -0.0)
    elif kind == 3:                                  # sorted / reversed / periodic columns
        a = np.sort(a, axis=0)
        a[:, ::2] = a[::-1, ::2]
        if n > 64:
            a[::32, -1] = 1e9                        # defeats a stride-32 pilot in the last column
    elif kind == 4:                                  # NaN and infinities
        a[rng.random((n, m)) < 0.001] = np.inf
        a[rng.random((n, m)) < 0.001] = -np.inf
        if n > 3:
            a[rng.integers(0, n), rng.integers(0, m)] = np.nan
    elif kind == 5:                                  # denormals and huge values
        a[:, 0] *= 1e-310
        a[::2, -1] *= 1e300
    pcts = [[25.0, 75.0], [5.0, 50.0, 95.0], [50.0], [0.0, 100.0], list(np.round(rng.uniform(0, 100, int(rng.integers(1, 8))), 3)), [25.0, 75.0, 5.0, 50.0, 95.0]][int(rng.integers(0, 6))]
    os.environ['PEM_QUANTILE_PILOT'] = str(int(rng.choice([2, 7, 32, 64] if args.pilot_only else [0, 2, 7, 32, 64])))
    os.environ['PEM_QUANTILE_PILOT_MIN'] = str(int(rng.choice([1, 1000] if args.pilot_only else [1, 1000, 1 << 25])))
    d = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    with np.errstate(invalid='ignore'):
        want = np.percentile(a, pcts, axis=0)
    got = drivers.column_percentiles(d, pcts).cpu().numpy()
    paths[lib.pem_quantiles_last_path()] += 1
    if not np.array_equal(got, want, equal_nan=True):
        bad = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
        raise SystemExit(f'case {case}: n={n} m={m} kind={kind} pcts={pcts} pilot={os.environ["PEM_QUANTILE_PILOT"]} min={os.environ["PEM_QUANTILE_PILOT_MIN"]}: '
                         f'{len(bad)} entries differ, first {bad[0]}: got {got[tuple(bad[0])]!r} want {want[tuple(bad[0])]!r}')
    if args.sharded:
        from hallthrusterpem_amd.percentiles import column_percentiles_sharded
        got_s = column_percentiles_sharded(d, pcts)
        if not np.array_equal(got_s, want, equal_nan=True):
            raise SystemExit(f'case {case}: n={n} m={m} kind={kind} pcts={pcts}: the sharded selection differs from np.percentile')
    if case % 7 == 0:
        f = float(rng.choice([1.5, 0.0, 3.0]))
        nh, oh = drivers.filter_outputs({'v': a}, iqr_factor=f)
        nd, od = drivers.filter_outputs({'v': d}, iqr_factor=f)
        if not (np.array_equal(nh['v'], nd['v'].cpu().numpy()) and np.array_equal(oh['v'], od['v'].cpu().numpy())):
            raise SystemExit(f'case {case}: filter_outputs masks differ (n={n} m={m} kind={kind} f={f})')
    if case % 50 == 49:
        print(f'{case + 1} cases: all equal; last calls went {paths}', flush=True)
print(f'{args.cases} cases (seed {args.seed}): every percentile equal to np.percentile bit for bit, every mask equal; paths taken by the last call of each case: {paths}')
