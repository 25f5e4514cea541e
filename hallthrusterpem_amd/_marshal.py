"""Turn the reference's dict-of-arrays convention into the flat fp64 SoA buffers of the C ABI."""
import ctypes as C

import numpy as np


def is_torch(x) -> bool:
    return type(x).__module__.split('.')[0] == 'torch'


def any_device_tensor(values) -> bool:
    return any(is_torch(v) and v.is_cuda for v in values)


def loop_shape(values, at_least_1d=True) -> tuple:
    """Broadcast shape of the per-sample inputs; the reference's np.atleast_1d (cathode.py:34,
    plume.py:59) turns an all-scalar call into a loop of one."""
    shape = np.broadcast_shapes(*[tuple(v.shape) if hasattr(v, 'shape') else np.shape(v) for v in values])
    if at_least_1d and shape == ():
        shape = (1,)
    return tuple(shape)


def host_flat(x, shape) -> np.ndarray:
    """Contiguous float64 copy/view of x broadcast to `shape` and flattened."""
    if is_torch(x):
        x = x.detach().cpu().numpy()
    a = np.asarray(x, dtype=np.float64)
    if a.shape != tuple(shape):
        a = np.broadcast_to(a, shape)
    return np.ascontiguousarray(a).reshape(-1)


def dev_flat(x, shape, device):
    """Contiguous float64 CUDA tensor of x broadcast to `shape` and flattened (torch is plumbing here)."""
    import torch
    t = x if is_torch(x) else torch.as_tensor(np.asarray(x, dtype=np.float64))
    t = t.to(device=device, dtype=torch.float64)
    if tuple(t.shape) != tuple(shape):
        t = t.broadcast_to(shape)
    return t.contiguous().reshape(-1)


def np_ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def t_ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream_ptr(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def pick_device(values):
    for v in values:
        if is_torch(v) and v.is_cuda:
            return v.device
    raise ValueError('no CUDA tensor among the inputs')


PINNED_THRESHOLD_BYTES = 32 << 20


def host_empty(n: int, dtype=np.float64) -> np.ndarray:
    """Uninitialised flat host array for results.  Large results are taken from torch's caching pinned-memory
    allocator: a device-to-host copy into page-locked memory runs at PCIe rate (~50 GB/s) instead of the ~18 GB/s of
    pageable memory, and the allocator reuses the pinned block once the array is garbage-collected, so a sampling
    loop pays the pinning once.  Falls back to np.empty if pinned memory is not available."""
    nbytes = int(n) * np.dtype(dtype).itemsize
    if nbytes >= PINNED_THRESHOLD_BYTES:
        try:
            import torch
            tdtype = {np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
            return torch.empty(int(n), dtype=tdtype, pin_memory=True).numpy()
        except Exception:
            pass
    return np.empty(int(n), dtype=dtype)
