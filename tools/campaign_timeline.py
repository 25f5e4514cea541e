#!/usr/bin/env python3
"""The kernels of ONE forward_uq_statistics call in time order, from a rocprofv3 kernel trace of tools/fused_probe.py
(tools/kstats.sh <tag> tools/fused_probe.py ...): start, duration, idle gap before it, queue.  python tools/campaign_timeline.py <trace.csv>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'plume_r1_kernel<4, true, 1, true, 0' in r['Kernel_Name']]      # the pilot evaluation opens a call
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:8.1f} us +{(e - s) / 1e3:7.1f}  idle before {max(0.0, (s - prev_end) / 1e3):6.1f}  queue {r.get('Queue_Id', '?')}  {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e)
print(f'call to call: {(int(rows[b]["Start_Timestamp"]) - t0) / 1e3:.1f} us')
