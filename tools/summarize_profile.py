#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 csv output of tools/profile_gpu.sh) into profiles/<tag>_summary.md and
profiles/traffic_<tag>.json (HBM bytes per launch of the coupled kernel, with the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md section HBM).   Usage: python tools/summarize_profile.py <tag> [samples_per_launch]"""
import csv
import os
import json
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 1_250_000
src = ROOT / 'gpurun_out' / f'prof_{tag}'
KERNEL = 'plume_r1_kernel'
lines = [f'# rocprofv3 summary `{tag}` (coupled bench, {n_samples} samples per launch)', '']


def find(pattern):
    return sorted(src.rglob(pattern))


# ---- kernel trace: per-kernel durations -------------------------------------------------------------
durs = defaultdict(list)
full_durs = []
for f in find('*kernel_trace.csv'):
    if 'trace_full' in str(f):     # the whole-config pass: only its long launches of the coupled kernel are of interest
        with open(f) as fd:
            for row in csv.DictReader(fd):
                d = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
                if KERNEL in row['Kernel_Name'] and d > 1_000_000:
                    full_durs.append(d)
        continue
    with open(f) as fd:
        for row in csv.DictReader(fd):
            durs[row['Kernel_Name']].append(int(row['End_Timestamp']) - int(row['Start_Timestamp']))
lines += ['## kernel trace (`rocprofv3 --kernel-trace --stats`)', '', '| kernel | calls | mean us | min us | max us | total ms |',
          '|---|---|---|---|---|---|']
kern_mean_us = None
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f'| `{k[:110]}` | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {min(v) / 1e3:.2f} | {max(v) / 1e3:.2f} | {sum(v) / 1e6:.3f} |')
    if KERNEL in k and kern_mean_us is None:
        steady = sorted(v)[: max(1, len(v) * 3 // 4)]
        kern_mean_us = sum(v) / len(v) / 1e3
lines.append('')
if full_durs:
    lines += [f'Whole-config launch (1e7 samples, 8.72 GB algorithmic; separate trace): {len(full_durs)} launches, mean '
              f'{sum(full_durs) / len(full_durs) / 1e3:.1f} us, min {min(full_durs) / 1e3:.1f} us = '
              f'{8.72e9 / (sum(full_durs) / len(full_durs) * 1e-9) / 1e12:.2f} TB/s.', '']
for f in [x for x in find('*kernel_stats.csv') if 'trace_full' not in str(x)]:
    lines += ['### rocprofv3 kernel_stats.csv (top rows, kernel names truncated)', '', '```']
    with open(f) as fd:
        for i, row in enumerate(csv.reader(fd)):
            if i > 6:
                break
            lines.append(','.join(c[:90] for c in row))
    lines += ['```', '']

# ---- counters ---------------------------------------------------------------------------------------
counters = defaultdict(list)
for f in find('*counter_collection.csv'):
    with open(f) as fd:
        for row in csv.DictReader(fd):
            if KERNEL in row.get('Kernel_Name', ''):
                counters[row['Counter_Name']].append(float(row['Counter_Value']))
lines += [f'## PMC, per launch of `{KERNEL}` (mean over dispatches; separate passes per block)', '', '| counter | mean | n |', '|---|---|---|']
mean = {}
for k, v in sorted(counters.items()):
    mean[k] = sum(v) / len(v)
    lines.append(f'| {k} | {mean[k]:.6g} | {len(v)} |')
lines.append('')
sys.path.insert(0, str(ROOT))
from bench import kernel_source_hash     # noqa: E402  (the digest bench.py checks before it replays this record)
rec = {'tag': tag, 'samples_per_launch': n_samples, 'kernel': KERNEL, 'kernel_mean_us': kern_mean_us,
       'kernel_srchash': kernel_source_hash(), 'layout': os.environ.get('PEM_PROFILE_LAYOUT', 'soa' if '--layout soa' in ' '.join(sys.argv) else 'tile')}   # (bench.py's default input layout: tile)
if 'FETCH_SIZE' in mean and 'WRITE_SIZE' in mean:
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly half the bytes of a wide coalesced
    # stream (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact for 16-B/lane streaming stores.
    fetch = mean['FETCH_SIZE'] * 1024 * 2
    write = mean['WRITE_SIZE'] * 1024
    algo = 872 * n_samples
    rec.update({'fetch_bytes_corrected': fetch, 'write_bytes': write, 'hbm_bytes_per_launch': fetch + write,
                'algorithmic_bytes_per_launch': algo, 'fetch_size_raw_kib': mean['FETCH_SIZE'],
                'write_size_raw_kib': mean['WRITE_SIZE']})
    lines += ['## HBM traffic per launch', '',
              f'* FETCH_SIZE {mean["FETCH_SIZE"]:.6g} KiB raw -> x1024 x2 (gfx950 correction) = {fetch / 1e6:.1f} MB '
              f'(algorithmic input bytes {120 * n_samples / 1e6:.1f} MB)',
              f'* WRITE_SIZE {mean["WRITE_SIZE"]:.6g} KiB raw -> x1024 = {write / 1e6:.1f} MB (algorithmic output bytes {752 * n_samples / 1e6:.1f} MB)',
              f'* total {(fetch + write) / 1e6:.1f} MB vs algorithmic {algo / 1e6:.1f} MB  (ratio {(fetch + write) / algo:.3f})', '']
    if kern_mean_us:
        lines.append(f'* at the traced mean duration {kern_mean_us:.1f} us: algorithmic {algo / kern_mean_us / 1e3:.0f} GB/s, '
                     f'measured traffic {(fetch + write) / kern_mean_us / 1e3:.0f} GB/s')
    (ROOT / 'profiles' / f'traffic_{tag}.json').write_text(json.dumps(rec, indent=1))
if 'SQ_WAVE_CYCLES' in mean:
    wc = mean['SQ_WAVE_CYCLES']
    lines += ['', '## SQ shares (of SQ_WAVE_CYCLES)', '']
    for k in ('SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_LDS'):
        if k in mean:
            lines.append(f'* {k} / SQ_WAVE_CYCLES = {mean[k] / wc:.3f}')
    if 'SQ_INSTS_VALU' in mean and 'SQ_WAVES' in mean:
        lines.append(f'* VALU instructions per wave = {mean["SQ_INSTS_VALU"] / mean["SQ_WAVES"]:.0f}')
out = ROOT / 'profiles' / f'{tag}_summary.md'
out.write_text('\n'.join(lines) + '\n')
print('\n'.join(lines))
