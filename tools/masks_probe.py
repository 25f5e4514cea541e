#!/usr/bin/env python3
"""pem_campaign_masks_f64_dev alone at 1e7 samples: microseconds per call with the mask rows on 4-byte boundaries (words written, four
samples per thread) and one byte off (bytes written, one sample per thread).  python tools/masks_probe.py [n]"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from hallthrusterpem_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nv = 3
x = torch.randn((nv, n), dtype=torch.float64, device='cuda')
q = torch.tensor(np.percentile(x[:, :100000].cpu().numpy(), [25, 75, 5, 50, 95], axis=1), device='cuda').contiguous()
cd = torch.randint(0, 60, (n,), dtype=torch.uint8, device='cuda')
ud = torch.randint(0, 3, (n,), dtype=torch.uint8, device='cuda')
rows = torch.empty(65536, dtype=torch.int64, device='cuda')
count = torch.zeros(1, dtype=torch.int32, device='cuda')
p = lambda t: C.c_void_p(t.data_ptr())
vars_ = (C.c_void_p * nv)(*[x[i].data_ptr() for i in range(nv)])
lib = _lib.load()
for pad in (0, 1, 0, 1):
    pitch = (n + 3) // 4 * 4 + pad
    nan_o = torch.empty((nv + 1, pitch), dtype=torch.uint8, device='cuda')[:, :n]
    out_o = torch.empty((nv + 1, pitch), dtype=torch.uint8, device='cuda')[:, :n]
    call = lambda: _lib.check(lib.pem_campaign_masks_f64_dev(n, nv, vars_, p(q), q.stride(0), 0, 1, 1.5, p(nan_o), p(out_o), nan_o.stride(0), p(cd), p(ud), 68,
                                                             p(rows), p(count), 65536, None))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        call()
    b.record()
    torch.cuda.synchronize()
    print(f'pitch n + {pitch - n}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call ({"words" if pitch % 4 == 0 else "bytes"})')
