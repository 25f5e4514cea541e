#!/usr/bin/env python3
"""Calibrate what this MI355X sustains for pure streaming writes/copies of the bench's size (torch kernels)."""
import torch
import sys
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for n in (1_250_000 * 94, 10_000_000 * 94):        # doubles written by one coupled launch: the 1.25e6-sample shard, the whole 1e7 config
    a = torch.empty(n, dtype=torch.float64, device='cuda')
    b = torch.empty(n, dtype=torch.float64, device='cuda')
    ms = t(lambda: a.fill_(1.5)); print(f'fill  {n*8/1e9:.2f} GB: {ms:.3f} ms  {n*8/ms/1e6:.0f} GB/s written')
    ms = t(lambda: b.copy_(a));   print(f'copy  {n*8/1e9:.2f} GB: {ms:.3f} ms  {2*n*8/ms/1e6:.0f} GB/s read+write')
    ms = t(lambda: a.sum());      print(f'sum   {n*8/1e9:.2f} GB: {ms:.3f} ms  {n*8/ms/1e6:.0f} GB/s read')
