#!/usr/bin/env python3
"""The SVD compress / reconstruct kernels in the STREAMING regime: eight field buffers (7.3 GB) and eight latent buffers in
rotation, so that neither the 0.91 GB a launch reads nor the 0.91 GB it writes is helped by the 256 MB Infinity Cache --
beside the single-buffer numbers of tools/svd_probe.py (DESIGN.md section 6 explains the difference for the coupled kernel)."""
import ctypes as C
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hallthrusterpem_amd import _lib, drivers
from hallthrusterpem_amd.compression import SVDCompression
n, NB = 1_250_000, 8
fields = [drivers.forward_uq(n, seed=2 + i, keep_profile=True)['j_ion'] for i in range(NB)]
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())          # noqa: E731


def t(fn, reps=40):
    for i in range(NB): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for norm in ('log10', 'none'):
    c = SVDCompression(norm=norm, reconstruction_tol=0.01).fit(fields[0][:50_000])
    lat = [torch.empty((n, c.rank), dtype=torch.float64, device='cuda') for _ in range(NB)]
    rec = [torch.empty((n, 91), dtype=torch.float64, device='cuda') for _ in range(NB)]
    basis = c.basis.contiguous()
    comp = lambda i: _lib.check(lib.pem_svd_compress_f64_dev(n, 91, c.rank, int(c.norm), c.scale, p(fields[i % NB]), p(basis), p(lat[i % NB]), st))       # noqa: E731
    reco = lambda i: _lib.check(lib.pem_svd_reconstruct_f64_dev(n, 91, c.rank, int(c.norm), c.scale, p(lat[i % NB]), p(basis), p(rec[i % NB]), st))      # noqa: E731
    one_c = lambda i: _lib.check(lib.pem_svd_compress_f64_dev(n, 91, c.rank, int(c.norm), c.scale, p(fields[0]), p(basis), p(lat[0]), st))               # noqa: E731
    one_r = lambda i: _lib.check(lib.pem_svd_reconstruct_f64_dev(n, 91, c.rank, int(c.norm), c.scale, p(lat[0]), p(basis), p(rec[0]), st))               # noqa: E731
    by = n * (91 + c.rank) * 8
    mc, mr, sc, sr = t(comp), t(reco), t(one_c), t(one_r)
    print(f'norm={norm:5s} rank={c.rank}: compress {mc*1e3:6.1f} us {by/mc/1e6:5.0f} GB/s streaming ({sc*1e3:6.1f} us {by/sc/1e6:5.0f} GB/s one buffer) | '
          f'reconstruct {mr*1e3:6.1f} us {by/mr/1e6:5.0f} GB/s streaming ({sr*1e3:6.1f} us {by/sr/1e6:5.0f} GB/s one buffer)', flush=True)
    del lat, rec
