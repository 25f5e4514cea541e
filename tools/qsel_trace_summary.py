#!/usr/bin/env python3
"""Per-kernel averages of the sharded selection's passes from rocprofv3 --kernel-trace --stats of tools/quantile_sharded_trace.py:
    python tools/qsel_trace_summary.py <dir with t<n>/run_kernel_stats.csv> n [n ...]"""
import csv, sys
root = sys.argv[1]
for n in map(int, sys.argv[2:]):
    print(f'== n = {n} samples x 91 columns ({n * 91 * 8 / 1e9:.2f} GB per pass)')
    for r in csv.DictReader(open(f'{root}/t{n}/run_kernel_stats.csv')):
        if 'qsel' in r['Name']:
            avg = float(r['AverageNs']) / 1e3
            name = r['Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
            name = name[:name.index('(')] if '(' in name else name
            stream = n * 91 * 8 / (avg * 1e-6) / 1e12
            rate = f'{stream:5.2f} TB/s' if any(k in name for k in ('hist', 'compact', 'minmax')) else ''
            print(f'{name:<32} calls {r["Calls"]:>4}  avg {avg:9.1f} us  {rate}')
