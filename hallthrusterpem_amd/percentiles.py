"""Exact percentiles over the sample axis when the samples are SHARDED over ranks (one process per GPU).

`np.percentile(arr, q, axis=0)` of gen_data.py:163-168 (IQR masks) and monte_carlo.py:363-658 (5 / 50 / 95 % bands) needs two
order statistics per column.  With all samples on one GPU `drivers.column_percentiles` copies the few candidates out and sorts
them (`pem_quantiles_f64_dev`); sharded over ranks that would mean gathering values.  Histograms add, values do not: here every
rank counts its own values inside the current key range of every wanted rank (`pem_range_hist_f64_dev`), the counts are
all-reduced (at most 144 KB), every rank picks the same bin and narrows the range to it, until the range is ONE key -- the
order statistic, exactly.  A campaign of 1e7 x 91 values on 8 GPUs moves ~1 MB over xGMI instead of 7.3 GB, in about nine
levels (each one streaming pass over the rank's shard).

The level logic is plain numpy on small arrays and takes its two local operations as callables, so the orchestration --
all-reduces, bin choice, inversion of the kernel's binning, numpy's interpolation -- runs under gloo on the CPU with a numpy
restatement of the histogram (tests/test_distributed_gloo.py), and the device kernel is held to that same restatement
(tests/test_quantiles.py).  Reference semantics: numpy's method 'linear', NaN in a column -> NaN (numpy/lib/_function_base_impl.py).
"""
import ctypes as C

import numpy as np

U64 = np.uint64
TOP = U64(1) << U64(63)
LDS_WORDS = 36864          # csrc/pem_quantile.hip: 32-bit counters per workgroup


# ---- the order-preserving image of a double, and the kernel's binning (restated; csrc/pem_quantile.hip range_scale) ----------
def key_of(x):
    b = np.ascontiguousarray(x, dtype=np.float64).view(U64)
    return np.where(b >> U64(63) != 0, ~b, b | TOP)


def value_of(k):
    k = np.ascontiguousarray(k, dtype=U64)
    return np.where(k >> U64(63) != 0, k & ~TOP, ~k).view(np.float64)


def range_scale(klo, khi, bins):
    """(shift, mult, identity) of the map key -> bin over [klo, khi], arrays of any common shape"""
    span = np.where(khi >= klo, khi - klo, U64(0)).astype(U64)
    shift = np.zeros(span.shape, dtype=U64)
    for _ in range(34):
        shift = np.where((span >> shift) >> U64(31) != 0, shift + U64(1), shift)
    d = span >> shift
    identity = d < U64(bins)
    mult = np.minimum((U64(bins) << U64(32)) // (d + U64(1)), U64(0xFFFFFFFF))
    return shift, mult, identity


def bin_interval(klo, khi, bins, b):
    """The keys of [klo, khi] that fall in bin b (arrays): the inverse of the kernel's map, clipped to the range."""
    shift, mult, identity = range_scale(klo, khi, bins)
    b = b.astype(U64)
    two32 = U64(1) << U64(32)
    d_lo = np.where(identity, b, (b * two32 + mult - U64(1)) // mult)                 # smallest d with floor(d mult / 2^32) == b
    d_hi = np.where(identity, b, ((b + U64(1)) * two32 + mult - U64(1)) // mult - U64(1))
    span = np.where(khi >= klo, khi - klo, U64(0)).astype(U64)
    last = d_hi >= (span >> shift)                       # the top bin ends at khi (and (d + 1) << shift may not fit 64 bits)
    lo = klo + (d_lo << shift)
    hi = np.where(last, khi, klo + (((np.where(last, U64(0), d_hi) + U64(1)) << shift) - U64(1)))
    return np.maximum(lo, klo), np.minimum(hi, khi)


def local_hist_numpy(a2d, klo, khi, bins):
    """hist[c][r][bin] of the finite-or-infinite values of a2d[:, c] whose key is in [klo[c][r], khi[c][r]] -- the numpy
    restatement of pem_range_hist_f64_dev (tests, and the CPU side of the gloo rehearsal)."""
    a2d = np.asarray(a2d, dtype=np.float64)
    m, nr = klo.shape
    hist = np.zeros((m, nr, bins), dtype=np.int64)
    shift, mult, identity = range_scale(klo, khi, bins)
    for c in range(m):
        col = a2d[:, c]
        k = key_of(col[~np.isnan(col)])
        for r in range(nr):
            kk = k[(k >= klo[c, r]) & (k <= khi[c, r])]
            d = (kk - klo[c, r]) >> shift[c, r]
            b = d if identity[c, r] else (d * mult[c, r]) >> U64(32)
            hist[c, r] = np.bincount(b.astype(np.int64), minlength=bins)[:bins]
    return hist


def local_minmax_numpy(a2d):
    a2d = np.asarray(a2d, dtype=np.float64)
    m = a2d.shape[1]
    kmin, kmax, nan = np.full(m, ~U64(0)), np.zeros(m, dtype=U64), np.zeros(m, dtype=np.int64)
    for c in range(m):
        col = a2d[:, c]
        nan[c] = int(np.isnan(col).any())
        k = key_of(col[~np.isnan(col)])
        if k.size:
            kmin[c], kmax[c] = k.min(), k.max()
    return kmin, kmax, nan


# ---- numpy's index arithmetic and interpolation (method 'linear') ----------------------------------------------------------------
def linear_ranks(n: int, percentiles):
    q = np.true_divide(np.atleast_1d(np.asarray(percentiles, dtype=np.float64)), np.float64(100))
    if not np.all((q >= 0) & (q <= 1)):
        raise ValueError('Percentiles must be in the range [0, 100]')
    virtual = (n - 1) * q
    prev = np.floor(virtual)
    nxt = prev + 1
    above = virtual >= n - 1
    prev[above], nxt[above] = -1, -1                       # "take the max value of the array": index -1
    gamma = virtual - prev                                  # (numpy takes the weight from the clipped index as well)
    return (np.where(prev < 0, n - 1, prev).astype(np.int64), np.where(nxt < 0, n - 1, nxt).astype(np.int64), gamma)


def lerp(a, b, t):
    """numpy's _lerp, operation for operation"""
    with np.errstate(invalid='ignore'):
        diff = b - a
        out = a + diff * t
        return np.where(t >= 0.5, b - diff * (1 - t), out)


# ---- the levels -----------------------------------------------------------------------------------------------------------------
def _all_reduce(arr, op, group):
    """all-reduce a small numpy int64 array over the process group (identity without one)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return arr
    backend = dist.get_backend(group)
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if backend == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=op, group=group)
    return t.cpu().numpy()


def sharded_percentiles(local_minmax, local_hist, n_local: int, m: int, percentiles, group=None, device_levels=None):
    """Percentiles (method 'linear') of the union of all ranks' rows, per column: array (len(percentiles), m), the same on every rank.

    local_minmax() -> (kmin[m], kmax[m] uint64 keys, has_nan[m]) of this rank's rows (kmin > kmax: none);
    local_hist(klo[m][nr], khi[m][nr] uint64, bins) -> counts [m][nr][bins] of this rank's keys inside the ranges, nr in {1, 2, 4, 6};
    device_levels(ranks[nr], klo0[m], khi0[m]) -> keys [m][nr]: the whole level loop done elsewhere (on the device), optional."""
    import torch.distributed as dist
    n = int(_all_reduce(np.array([n_local], dtype=np.int64), dist.ReduceOp.SUM, group)[0])
    if n == 0:
        raise ValueError('no samples')
    scalar = np.ndim(percentiles) == 0
    rank_prev, rank_next, gamma = linear_ranks(n, percentiles)
    nq = gamma.size
    kmin, kmax, nan = local_minmax()
    # unsigned keys through a signed all-reduce: flipping the top bit maps unsigned order to signed order
    smin = _all_reduce((kmin ^ TOP).view(np.int64), dist.ReduceOp.MIN, group).view(U64) ^ TOP
    smax = _all_reduce((kmax ^ TOP).view(np.int64), dist.ReduceOp.MAX, group).view(U64) ^ TOP
    nan = _all_reduce(np.asarray(nan, dtype=np.int64), dist.ReduceOp.MAX, group)
    empty = smin > smax
    out = np.empty((nq, m))
    for q0 in range(0, nq, 3):                                                    # three quantiles = six ranks per column and pass
        sel = slice(q0, min(q0 + 3, nq))
        ranks = np.stack([rank_prev[sel], rank_next[sel]], axis=1).reshape(-1)    # [prev0, next0, prev1, ...]
        nr = {2: 2, 4: 4, 6: 6}[ranks.size]
        if device_levels is not None:                                             # the same levels, resident on the device
            vals = value_of(device_levels(ranks, np.where(empty, U64(1), smin).astype(U64), np.where(empty, U64(0), smax).astype(U64)))
            for i, qi in enumerate(range(sel.start, sel.stop)):
                out[qi] = lerp(vals[:, 2 * i], vals[:, 2 * i + 1], gamma[qi])
            continue
        bins = 1
        while 2 * bins * m * nr <= LDS_WORDS and 2 * bins <= 4096:
            bins *= 2
        klo = np.repeat(np.where(empty, U64(1), smin)[:, None], nr, axis=1).astype(U64)
        khi = np.repeat(np.where(empty, U64(0), smax)[:, None], nr, axis=1).astype(U64)
        resid = np.broadcast_to(ranks, (m, nr)).astype(np.int64).copy()
        for _level in range(80):
            if not np.any(klo < khi):
                break
            hist = _all_reduce(np.asarray(local_hist(klo, khi, bins), dtype=np.int64), dist.ReduceOp.SUM, group)
            cum = np.cumsum(hist, axis=2)
            b = np.minimum((cum <= resid[:, :, None]).sum(axis=2), bins - 1)     # first bin whose cumulative count exceeds the rank
            before = np.where(b > 0, np.take_along_axis(cum, np.maximum(b - 1, 0)[:, :, None], axis=2)[:, :, 0], 0)
            live = klo < khi
            nlo, nhi = bin_interval(klo, khi, bins, b)
            resid = np.where(live, resid - before, resid)
            klo, khi = np.where(live, nlo, klo), np.where(live, nhi, khi)
        else:
            raise RuntimeError('percentile refinement did not converge')
        vals = value_of(klo)                                                      # [m][nr]: x_(prev), x_(next) per quantile
        for i, qi in enumerate(range(sel.start, sel.stop)):
            out[qi] = lerp(vals[:, 2 * i], vals[:, 2 * i + 1], gamma[qi])
    out[:, (nan != 0) | empty] = np.nan
    return out[0] if scalar else out


class DeviceColumns:
    """This rank's rows as a CUDA tensor (n_local, ...), with the two local operations of `sharded_percentiles` on the device."""

    def __init__(self, a):
        import torch
        from . import _lib
        self.lib = _lib
        flat = a.double().reshape(a.shape[0], -1) if a.shape[0] else a.double().reshape(0, int(np.prod(a.shape[1:])) or 1)
        self.flat = flat if flat.is_contiguous() else flat.contiguous()
        self.n, self.m = self.flat.shape
        self.trailing = tuple(a.shape[1:])
        self.stream = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
        self.torch = torch

    def _chunks(self):
        return [(c0, min(256, self.m - c0)) for c0 in range(0, self.m, 256)]

    def minmax(self):
        torch = self.torch
        kmin = torch.empty(self.m, dtype=torch.int64, device=self.flat.device)
        kmax, nan = torch.empty_like(kmin), torch.empty(self.m, dtype=torch.int32, device=self.flat.device)
        with torch.cuda.device(self.flat.device):
            for c0, mc in self._chunks():
                self.lib.check(self.lib.load().pem_key_minmax_f64_dev(
                    self.n, mc, C.c_void_p(self.flat.data_ptr() + 8 * c0), self.m, C.c_void_p(kmin.data_ptr() + 8 * c0),
                    C.c_void_p(kmax.data_ptr() + 8 * c0), C.c_void_p(nan.data_ptr() + 4 * c0), self.stream))
        return kmin.cpu().numpy().view(U64), kmax.cpu().numpy().view(U64), nan.cpu().numpy().astype(np.int64)

    def hist(self, klo, khi, bins):
        torch = self.torch
        nr = klo.shape[1]
        out = np.empty((self.m, nr, bins), dtype=np.int64)
        with torch.cuda.device(self.flat.device):
            for c0, mc in self._chunks():
                d_lo = torch.from_numpy(np.ascontiguousarray(klo[c0:c0 + mc]).view(np.int64)).to(self.flat.device)
                d_hi = torch.from_numpy(np.ascontiguousarray(khi[c0:c0 + mc]).view(np.int64)).to(self.flat.device)
                h = torch.empty((mc, nr, bins), dtype=torch.int32, device=self.flat.device)
                self.lib.check(self.lib.load().pem_range_hist_f64_dev(
                    self.n, mc, C.c_void_p(self.flat.data_ptr() + 8 * c0), self.m, nr, C.c_void_p(d_lo.data_ptr()),
                    C.c_void_p(d_hi.data_ptr()), bins, C.c_void_p(h.data_ptr()), self.stream))
                out[c0:c0 + mc] = h.cpu().numpy().view(np.uint32)
        return out


def _device_levels(cols, ranks, klo0, khi0, group):
    """The level loop with ranges, counts and decisions resident on the device (`pem_range_hist_f64_dev` -> all-reduce ->
    `pem_range_narrow_dev`): per level one streaming pass, one collective on a device tensor and one flag read -- no histogram
    crosses PCIe.  `ranks`: (nr,) wanted ranks of this pass, klo0 / khi0: (m,) the columns' global key range; returns the (m, nr) keys.  One chunk of columns (m <= 256)."""
    import torch
    import torch.distributed as dist
    lib, dev, m, nr = cols.lib.load(), cols.flat.device, cols.m, ranks.size
    bins = 1
    while 2 * bins * m * nr <= LDS_WORDS and 2 * bins <= 4096:
        bins *= 2
    klo = torch.from_numpy(np.repeat(klo0[:, None], nr, axis=1).view(np.int64).copy()).to(dev)
    khi = torch.from_numpy(np.repeat(khi0[:, None], nr, axis=1).view(np.int64).copy()).to(dev)
    resid = torch.from_numpy(np.broadcast_to(ranks, (m, nr)).astype(np.int64).copy()).to(dev)
    hist = torch.empty((m, nr, bins), dtype=torch.int32, device=dev)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    top = torch.tensor(-2 ** 63, dtype=torch.int64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())                                                       # noqa: E731
    with torch.cuda.device(dev):
        for _level in range(80):
            if not bool(((klo ^ top) < (khi ^ top)).any()):            # unsigned comparison of the keys
                break
            cols.lib.check(lib.pem_range_hist_f64_dev(cols.n, m, p(cols.flat), m, nr, p(klo), p(khi), bins, p(hist), cols.stream))
            if multi:
                if dist.get_backend(group) == 'nccl':
                    dist.all_reduce(hist, group=group)
                else:                                                   # gloo rehearsals: the counts take the host route
                    h = hist.cpu()
                    dist.all_reduce(h, group=group)
                    hist.copy_(h)
            cols.lib.check(lib.pem_range_narrow_dev(m * nr, bins, p(hist), p(klo), p(khi), p(resid), cols.stream))
        else:
            raise RuntimeError('percentile refinement did not converge')
    return klo.cpu().numpy().view(U64)


def column_percentiles_sharded(a, percentiles, group=None, on_device: bool = True):
    """`np.percentile(concatenation of every rank's `a`, percentiles, axis=0)`, bit for bit, on every rank: `a` is this rank's
    (n_local, ...) CUDA tensor (n_local may be 0 on some ranks).  Returns a numpy array (len(percentiles), ...)."""
    cols = DeviceColumns(a)          # (more than 256 columns: the local operations go through them 256 at a time)
    levels = (lambda ranks, klo0, khi0: _device_levels(cols, ranks, klo0, khi0, group)) if (on_device and cols.m <= 256) else None
    res = sharded_percentiles(cols.minmax, cols.hist, cols.n, cols.m, percentiles, group=group, device_levels=levels)
    return res.reshape(res.shape[:-1] + cols.trailing) if cols.trailing else res[..., 0]
