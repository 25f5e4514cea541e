import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from hallthrusterpem_amd.compression import SVDCompression
n,dof,rank=600_000,202,7
x=torch.rand((n,dof),dtype=torch.float64,device='cuda')*1e4+1e3
c=SVDCompression(norm='linear',scale=1e-3,rank=rank); c.basis=torch.from_numpy(np.linalg.qr(np.random.default_rng(0).standard_normal((dof,rank)))[0].copy()).cuda()
def t(fn,reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps
z=c.compress(x)
by=n*(dof+rank)*8
mc,mr=t(lambda:c.compress(x)),t(lambda:c.reconstruct(z))
print(f'dof=202 rank=7 linear: compress {mc*1e3:.1f} us {by/mc/1e6:.0f} GB/s | reconstruct {mr*1e3:.1f} us {by/mr/1e6:.0f} GB/s')
