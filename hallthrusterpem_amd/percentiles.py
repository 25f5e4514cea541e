"""Exact percentiles over the sample axis when the samples are SHARDED over ranks (one process per GPU).

`np.percentile(arr, q, axis=0)` of gen_data.py:163-168 (IQR masks) and monte_carlo.py:363-658 (5 / 50 / 95 % bands) needs two
order statistics per column.  With all samples on one GPU `drivers.column_percentiles` selects them in four streaming passes
(`pem_quantiles_f64_dev`): min / max, a histogram per column, a second histogram inside the chosen bins, and a copy of the few
values left.  Values cannot be merged across GPUs without moving them, but every product of those passes can: min / max and the
two histograms ADD (all-reduce MIN / MAX / SUM of 91 x 256 and 91 x 6 x 64 counters), so every rank takes the same decisions,
and what is left per wanted rank -- 1 / (bins1 bins2) of a column, a few hundred keys -- is copied into fixed-length padded
lists, all-gathered, and selected from by one workgroup per list (`pem_qsel_*`, csrc/pem_quantile.hip).  Four passes over the
rank's shard and ~1 MB over xGMI for a campaign of 1e7 x 91 values on 8 GPUs, instead of all-gathering 7.3 GB of profiles.
(Rounds 1-2 narrowed a key range by 64 bins per pass instead -- eleven passes; that level loop is kept as the fallback for
lists that do not fit their cap: heavy ties such as the 1e-20 profile of invalid samples under a wanted rank.)

The stage logic exists twice on purpose: `DeviceColumns` runs the kernels on the rank's CUDA tensor, `NumpyColumns` restates
every stage -- the integer binning, the decisions, the padded lists -- in numpy, so that the orchestration (collectives, numpy's
index arithmetic and interpolation) runs under gloo on the CPU (tests/test_distributed_gloo.py) and the kernels are held to the
restatement stage by stage (tests/test_quantiles.py).  Reference semantics: numpy's method 'linear', NaN in a column -> NaN.
"""
import ctypes as C

import numpy as np

U64 = np.uint64
TOP = U64(1) << U64(63)
LDS_WORDS = 36864          # csrc/pem_quantile.hip: 32-bit counters per workgroup
MAX_COLUMNS = 256          # columns per kernel call (a lane owns up to four)
LIST_CAP = 65536           # longest padded candidate list per (column, target) and rank ...
GATHER_CAP_BYTES = 1 << 29 # ... and the most all the ranks' lists of one pass may weigh once gathered: beyond either the level loop takes over
PAD = ~U64(0)              # padding of the candidate lists: above the key of +inf


# ---- the order-preserving image of a double -------------------------------------------------------------------------------------
def key_of(x):
    b = np.ascontiguousarray(x, dtype=np.float64).view(U64)
    return np.where(b >> U64(63) != 0, ~b, b | TOP)


def value_of(k):
    k = np.ascontiguousarray(k, dtype=U64)
    return np.where(k >> U64(63) != 0, k & ~TOP, ~k).view(np.float64)


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


# ---- collectives on small arrays ---------------------------------------------------------------------------------------------------
def _dist(group):
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    return dist, on


def _all_reduce(arr, op, group):
    """all-reduce a small numpy int64 array over the process group (identity without one)"""
    import torch
    dist, on = _dist(group)
    if not on:
        return arr
    t = torch.from_numpy(np.array(arr, copy=True))          # (a copy: the collective works in place)
    if dist.get_backend(group) == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=op, group=group)
    return t.cpu().numpy()


def _reduce_keys(kmin, kmax, nan, group):
    """global kmin / kmax / has_nan from the ranks' own: unsigned keys go through a signed all-reduce with the top bit flipped"""
    import torch.distributed as dist
    smin = _all_reduce((np.asarray(kmin, dtype=U64) ^ TOP).view(np.int64), dist.ReduceOp.MIN, group).view(U64) ^ TOP
    smax = _all_reduce((np.asarray(kmax, dtype=U64) ^ TOP).view(np.int64), dist.ReduceOp.MAX, group).view(U64) ^ TOP
    return smin, smax, _all_reduce(np.asarray(nan, dtype=np.int64), dist.ReduceOp.MAX, group)


def pow2_at_most(x: int, cap: int) -> int:
    p = 1
    while 2 * p <= x and 2 * p <= cap:
        p *= 2
    return p


def qsel_bins(m: int, nt: int):
    """(bins1, bins2) of the two histograms for m columns and nt targets per column (`pem_qsel_bins`)"""
    return pow2_at_most(LDS_WORDS // m, 4096), pow2_at_most(LDS_WORDS // (m * nt), 4096)


def _list_len(need: int) -> int:
    p = 16
    while p < need:
        p *= 2
    return p


def _lists_fit(need: int, world: int, m: int, nt: int) -> bool:
    """Whether padded lists of `need` keys per (column, target) and rank are short enough to gather and select from."""
    return need <= LIST_CAP and world * m * nt * _list_len(need) * 8 <= GATHER_CAP_BYTES


# ---- the stages, restated in numpy (csrc/pem_quantile.hip: column_scale, bin_of, subbin_of, find_bin, qsel_*_kernel) ----------------
def column_scale(kmin, kmax, bins1):
    """(shift, mult) of key -> bin over [kmin, kmax] per column: d = (k - kmin) >> shift < 2^31, bin = floor(d mult / 2^32)"""
    kmin, kmax = np.asarray(kmin, dtype=U64), np.asarray(kmax, dtype=U64)
    span = np.where(kmax >= kmin, kmax - kmin, U64(0)).astype(U64)
    shift = np.zeros(span.shape, dtype=U64)
    for _ in range(34):
        shift = np.where((span >> shift) >> U64(31) != 0, shift + U64(1), shift)
    mult = np.minimum((U64(bins1) << U64(32)) // ((span >> shift) + U64(1)), U64(0xFFFFFFFF))
    return shift, mult


def _bins_of(keys, kmin, shift, mult, bins2):
    p = ((keys - kmin) >> shift) * mult
    return (p >> U64(32)).astype(np.int64), (((p & U64(0xFFFFFFFF)) * U64(bins2)) >> U64(32)).astype(np.int64)


def _find_bin(hist, rank):
    """first bin whose cumulative count exceeds `rank` (clamped to the last), the count below it, its own count"""
    cum = np.cumsum(hist.astype(np.int64))
    b = int(min((cum <= rank).sum(), hist.size - 1))
    return b, (int(cum[b - 1]) if b > 0 else 0), int(hist[b])


class NumpyColumns:
    """This rank's rows as a numpy array (n_local, m): every stage of the selection restated in numpy."""

    def __init__(self, a):
        a = np.asarray(a, dtype=np.float64)
        self.a = a.reshape(a.shape[0], -1) if a.shape[0] else a.reshape(0, int(np.prod(a.shape[1:])) or 1)
        self.n, self.m = self.a.shape
        self.trailing = tuple(a.shape[1:])
        self._keys = [key_of(col[~np.isnan(col)]) for col in self.a.T]

    def minmax(self):
        kmin, kmax, nan = np.full(self.m, ~U64(0)), np.zeros(self.m, dtype=U64), np.zeros(self.m, dtype=np.int64)
        for c, k in enumerate(self._keys):
            nan[c] = int(k.size < self.n)
            if k.size:
                kmin[c], kmax[c] = k.min(), k.max()
        return kmin, kmax, nan

    def hist1(self, kmin, kmax, bins1):
        shift, mult = column_scale(kmin, kmax, bins1)
        out = np.zeros((self.m, bins1), dtype=np.int64)
        for c, k in enumerate(self._keys):
            if k.size:
                out[c] = np.bincount(_bins_of(k, kmin[c], shift[c], mult[c], 1)[0], minlength=bins1)[:bins1]
        return out

    def hist2(self, kmin, kmax, bin1, bins1, bins2):
        shift, mult = column_scale(kmin, kmax, bins1)
        nt = bin1.shape[1]
        out = np.zeros((self.m, nt, bins2), dtype=np.int64)
        for c, k in enumerate(self._keys):
            b, sb = _bins_of(k, kmin[c], shift[c], mult[c], bins2) if k.size else (np.zeros(0, np.int64),) * 2
            for t in range(nt):
                if bin1[c, t] < 0 or bin1[c, t] in bin1[c, :t]:          # a shared bin is counted under the first target that has it
                    continue
                out[c, t] = np.bincount(sb[b == bin1[c, t]], minlength=bins2)[:bins2]
        return out

    def compact(self, kmin, kmax, bin1, bin2, done, bins1, bins2, list_len):
        shift, mult = column_scale(kmin, kmax, bins1)
        nt = bin1.shape[1]
        cand = np.full((self.m, nt, list_len), PAD, dtype=U64)
        cursor = np.zeros((self.m, nt), dtype=np.int64)
        for c, k in enumerate(self._keys):
            b, sb = _bins_of(k, kmin[c], shift[c], mult[c], bins2) if k.size else (np.zeros(0, np.int64),) * 2
            for t in range(nt):
                if done[c, t] or any((not done[c, u]) and bin1[c, u] == bin1[c, t] and bin2[c, u] == bin2[c, t] for u in range(t)):
                    continue
                mine = k[(b == bin1[c, t]) & (sb == bin2[c, t])]
                cursor[c, t] = mine.size
                cand[c, t, :min(mine.size, list_len)] = mine[:list_len]
        return cand, cursor

    # the two ways through a pass of up to six quantiles (`sharded_percentiles` calls these)
    def select_pass(self, ranks, kmin, kmax, group):
        return _numpy_pass(self, ranks, kmin, kmax, group)

    def levels_pass(self, ranks, kmin, kmax, group):
        return narrow_by_levels(lambda lo, hi, b: local_hist_numpy(self.a, lo, hi, b), ranks, kmin, kmax, group)


def decide1(kmin, kmax, hist1, resid):
    """per (column, target): bin1 that holds the rank, the rank inside it; constant / empty columns are done (answer = kmin)"""
    m, nt = resid.shape
    bin1 = np.full((m, nt), -1, dtype=np.int64)
    done = np.zeros((m, nt), dtype=bool)
    answer = np.zeros((m, nt), dtype=U64)
    resid = resid.copy()
    for c in range(m):
        if kmin[c] >= kmax[c]:
            done[c], answer[c] = True, kmin[c]
            continue
        for t in range(nt):
            bin1[c, t], before, _ = _find_bin(hist1[c], resid[c, t])
            resid[c, t] -= before
    return bin1, done, answer, resid


def decide2(hist2, hist2_local, bin1, done, resid):
    m, nt = resid.shape
    bin2 = np.full((m, nt), -1, dtype=np.int64)
    count = np.zeros((m, nt), dtype=np.int64)
    count_local = np.zeros((m, nt), dtype=np.int64)
    resid = resid.copy()
    for c in range(m):
        for t in range(nt):
            if done[c, t]:
                continue
            first = next(u for u in range(t + 1) if not done[c, u] and bin1[c, u] == bin1[c, t])
            bin2[c, t], before, count[c, t] = _find_bin(hist2[c, first], resid[c, t])
            resid[c, t] -= before
            count_local[c, t] = hist2_local[c, first, bin2[c, t]]
    return bin2, count, count_local, resid


def select_lists(gathered, bin1, bin2, done, resid, answer):
    """gathered: (world, m, nt, L) padded lists.  x_(resid) of the union of the ranks' lists of each target's (bin1, bin2)."""
    world, m, nt, _ = gathered.shape
    answer = answer.copy()
    for c in range(m):
        for t in range(nt):
            if done[c, t]:
                continue
            owner = next(u for u in range(t + 1) if not done[c, u] and bin1[c, u] == bin1[c, t] and bin2[c, u] == bin2[c, t])
            keys = gathered[:, c, owner, :].reshape(-1)
            keys = np.sort(keys[keys != PAD])
            # (an empty union: a rank past the column's last value, i.e. a column that holds a NaN -- its result is NaN whatever this is)
            answer[c, t] = keys[min(int(resid[c, t]), keys.size - 1)] if keys.size else PAD
    return answer


def _numpy_pass(cols: NumpyColumns, ranks, kmin, kmax, group):
    """One pass (up to six quantiles = twelve targets per column) of the selection on the numpy restatement, collectives over
    gloo.  Returns the (m, nt) keys, or None when the lists would not fit (`_lists_fit`: the level loop takes over)."""
    import torch
    import torch.distributed as dist
    _, on = _dist(group)
    m, nt = cols.m, ranks.size
    bins1, bins2 = qsel_bins(m, nt)
    resid = np.broadcast_to(ranks, (m, nt)).astype(np.int64).copy()
    h1 = _all_reduce(cols.hist1(kmin, kmax, bins1), dist.ReduceOp.SUM, group)
    bin1, done, answer, resid = decide1(kmin, kmax, h1, resid)
    h2_local = cols.hist2(kmin, kmax, bin1, bins1, bins2)
    h2 = _all_reduce(h2_local, dist.ReduceOp.SUM, group)
    bin2, _, count_local, resid = decide2(h2, h2_local, bin1, done, resid)
    need = int(_all_reduce(np.array([count_local.max(initial=0)], dtype=np.int64), dist.ReduceOp.MAX, group)[0])
    if not _lists_fit(need, dist.get_world_size(group) if on else 1, m, nt):
        return None
    L = _list_len(need)
    cand, cursor = cols.compact(kmin, kmax, bin1, bin2, done, bins1, bins2, L)
    assert cursor.max(initial=0) <= L
    if on:
        world = dist.get_world_size(group)
        mine = torch.from_numpy(cand.view(np.int64))
        pieces = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(pieces, mine, group=group)
        gathered = np.stack([p.numpy().view(U64) for p in pieces])
    else:
        gathered = cand[None]
    return select_lists(gathered, bin1, bin2, done, resid, answer)


# ---- the level loop of rounds 1-2 (fallback): narrow a key range per wanted rank by `bins` bins per streaming pass -------------------
def range_scale(klo, khi, bins):
    """(shift, mult, identity) of the map key -> bin over [klo, khi], arrays of any common shape (pem_range_hist_f64_dev)"""
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
    restatement of pem_range_hist_f64_dev"""
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
    return NumpyColumns(a2d).minmax()


def level_bins(chunk_columns: int, nr: int) -> int:
    """Bins per range and level: what fits the kernel's LDS counters for ONE call's columns (the local histogram goes through
    wide arrays 256 columns at a time, so the bound is the chunk's width, not the array's)."""
    bins = 1
    while 2 * bins * chunk_columns * nr <= LDS_WORDS and 2 * bins <= 4096:
        bins *= 2
    return bins


def narrow_by_levels(local_hist, ranks, kmin, kmax, group=None, chunk_columns=None):
    """keys [m][nr] of the wanted ranks by the level loop: local_hist(klo[m][nr], khi[m][nr] uint64, bins) -> this rank's counts
    [m][nr][bins]; the counts are all-reduced, every rank picks the same bin and narrows the range to it, until a range is ONE key."""
    import torch.distributed as dist
    m, nr = kmin.size, ranks.size
    empty = kmin > kmax
    bins = level_bins(min(m, MAX_COLUMNS) if chunk_columns is None else chunk_columns, nr)
    klo = np.repeat(np.where(empty, U64(1), kmin)[:, None], nr, axis=1).astype(U64)
    khi = np.repeat(np.where(empty, U64(0), kmax)[:, None], nr, axis=1).astype(U64)
    resid = np.broadcast_to(ranks, (m, nr)).astype(np.int64).copy()
    for _level in range(80):
        if not np.any(klo < khi):
            return klo
        hist = _all_reduce(np.asarray(local_hist(klo, khi, bins), dtype=np.int64), dist.ReduceOp.SUM, group)
        cum = np.cumsum(hist, axis=2)
        b = np.minimum((cum <= resid[:, :, None]).sum(axis=2), bins - 1)     # first bin whose cumulative count exceeds the rank
        before = np.where(b > 0, np.take_along_axis(cum, np.maximum(b - 1, 0)[:, :, None], axis=2)[:, :, 0], 0)
        live = klo < khi
        nlo, nhi = bin_interval(klo, khi, bins, b)
        resid = np.where(live, resid - before, resid)
        klo, khi = np.where(live, nlo, klo), np.where(live, nhi, khi)
    raise RuntimeError('percentile refinement did not converge')


# ---- the driver ------------------------------------------------------------------------------------------------------------------------
def sharded_percentiles(cols, percentiles, group=None, method: str = 'select'):
    """Percentiles (method 'linear') of the union of all ranks' rows, per column: array (len(percentiles), m), the same on every
    rank.  `cols`: this rank's rows as `NumpyColumns` or `DeviceColumns`.  method 'select': the four-pass selection, the level
    loop only where a candidate list would not fit; 'levels': the level loop throughout (rounds 1-2)."""
    import torch.distributed as dist
    n = int(_all_reduce(np.array([cols.n], dtype=np.int64), dist.ReduceOp.SUM, group)[0])
    if n == 0:
        raise ValueError('no samples')
    scalar = np.ndim(percentiles) == 0
    rank_prev, rank_next, gamma = linear_ranks(n, percentiles)
    nq, m = gamma.size, cols.m
    kmin, kmax, nan = cols.reduced_minmax(group) if hasattr(cols, 'reduced_minmax') else _reduce_keys(*cols.minmax(), group)
    empty = kmin > kmax
    out = np.empty((nq, m))
    # six quantiles = twelve ranks per column and pass (three for more than 128 columns: PEM_QUANTILE_MAX_Q / _WIDE): a campaign's
    # p25 / p75 (the IQR masks) and its 5 / 50 / 95 % bands share the four passes
    step = 6 if m <= 128 else 3
    for q0 in range(0, nq, step):
        sel = slice(q0, min(q0 + step, nq))
        ranks = np.stack([rank_prev[sel], rank_next[sel]], axis=1).reshape(-1)    # [prev0, next0, prev1, ...]
        keys = cols.select_pass(ranks, kmin, kmax, group) if method == 'select' else None
        if keys is None:                                                          # (the level loop takes six ranks at a time)
            keys = np.concatenate([cols.levels_pass(ranks[r0:r0 + 6], kmin, kmax, group) for r0 in range(0, ranks.size, 6)], axis=1)
        vals = value_of(keys)                                                     # [m][nt]: x_(prev), x_(next) per quantile
        for i, qi in enumerate(range(sel.start, sel.stop)):
            out[qi] = lerp(vals[:, 2 * i], vals[:, 2 * i + 1], gamma[qi])
    out[:, (nan != 0) | empty] = np.nan
    return out[0] if scalar else out


class DeviceColumns:
    """This rank's rows as a CUDA tensor (n_local, ...): the stages of the selection on the device (`pem_qsel_*`), state and
    collectives on device tensors -- per pass two small device-to-host reads (the list length, the answers)."""

    def __init__(self, a):
        import torch
        from . import _lib
        self.lib = _lib
        flat = a.double().reshape(a.shape[0], -1) if a.shape[0] else a.double().reshape(0, int(np.prod(a.shape[1:])) or 1)
        self.flat = flat if flat.is_contiguous() else flat.contiguous()
        self.n, self.m = self.flat.shape
        self.trailing = tuple(a.shape[1:])
        self.dev = self.flat.device
        self.stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        self.torch = torch
        self._top = torch.tensor(-2 ** 63, dtype=torch.int64, device=self.dev)

    def _chunks(self):
        return [(c0, min(MAX_COLUMNS, self.m - c0)) for c0 in range(0, self.m, MAX_COLUMNS)]

    def _data(self, c0):
        return C.c_void_p(self.flat.data_ptr() + 8 * c0)

    # -- collectives on device tensors (RCCL directly; gloo rehearsals take the host route) --
    def _reduce(self, t, op, group):
        dist, on = _dist(group)
        if not on:
            return t
        if dist.get_backend(group) == 'nccl':
            dist.all_reduce(t, op=op, group=group)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op, group=group)
            t.copy_(h)
        return t

    def _gather(self, t, group):
        torch = self.torch
        dist, on = _dist(group)
        if not on:
            return t[None], 1
        world = dist.get_world_size(group)
        if dist.get_backend(group) == 'nccl':
            out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t.contiguous(), group=group)
            return out, world
        h = t.cpu()
        pieces = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(pieces, h, group=group)
        return torch.stack(pieces).to(t.device), world

    def minmax_device(self):
        torch = self.torch
        kmin = torch.empty(self.m, dtype=torch.int64, device=self.dev)
        kmax, nan = torch.empty_like(kmin), torch.empty(self.m, dtype=torch.int32, device=self.dev)
        with torch.cuda.device(self.dev):
            for c0, mc in self._chunks():
                self.lib.check(self.lib.load().pem_qsel_minmax_f64_dev(
                    self.n, mc, self._data(c0), self.m, C.c_void_p(kmin.data_ptr() + 8 * c0), C.c_void_p(kmax.data_ptr() + 8 * c0),
                    C.c_void_p(nan.data_ptr() + 4 * c0), self.stream))
        return kmin, kmax, nan

    def minmax(self):
        kmin, kmax, nan = self.minmax_device()
        return kmin.cpu().numpy().view(U64), kmax.cpu().numpy().view(U64), nan.cpu().numpy().astype(np.int64)

    def reduced_minmax(self, group):
        """global kmin / kmax / has_nan (numpy) and the same keys kept on the device for the passes"""
        import torch.distributed as dist
        kmin, kmax, nan = self.minmax_device()
        _, on = _dist(group)
        if on:
            # unsigned keys through a signed MIN / MAX: the top bit flipped maps unsigned order to signed order
            self._reduce(kmin.bitwise_xor_(self._top), dist.ReduceOp.MIN, group).bitwise_xor_(self._top)
            self._reduce(kmax.bitwise_xor_(self._top), dist.ReduceOp.MAX, group).bitwise_xor_(self._top)
            self._reduce(nan, dist.ReduceOp.MAX, group)
        self._kmin, self._kmax = kmin, kmax
        return kmin.cpu().numpy().view(U64), kmax.cpu().numpy().view(U64), nan.cpu().numpy().astype(np.int64)

    def select_pass(self, ranks, kmin, kmax, group):
        """One pass of the selection (<= 6 quantiles; <= 3 for more than 128 columns) on the device, the columns MAX_COLUMNS at a time.  (m, nt) keys, or None
        when the candidate lists would not fit (`_lists_fit`)."""
        import torch.distributed as dist
        torch, lib = self.torch, self.lib.load()
        nt = int(ranks.size)
        p = lambda t: C.c_void_p(t.data_ptr())                                                       # noqa: E731
        if not hasattr(self, '_kmin'):
            self._kmin = torch.from_numpy(np.ascontiguousarray(kmin).view(np.int64).copy()).to(self.dev)
            self._kmax = torch.from_numpy(np.ascontiguousarray(kmax).view(np.int64).copy()).to(self.dev)
        out = np.empty((self.m, nt), dtype=U64)
        ranks_dev = torch.from_numpy(np.ascontiguousarray(ranks, dtype=np.int64)).to(self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        i64 = dict(dtype=torch.int64, device=self.dev)
        with torch.cuda.device(self.dev):
            for c0, mc in self._chunks():
                bins1, bins2 = qsel_bins(mc, nt)
                kmn, kmx = self._kmin[c0:c0 + mc], self._kmax[c0:c0 + mc]
                resid = ranks_dev.repeat(mc, 1).contiguous()
                hist1 = torch.empty((mc, bins1), **i32)
                self.lib.check(lib.pem_qsel_hist1_f64_dev(self.n, mc, self._data(c0), self.m, p(kmn), p(kmx), bins1, p(hist1), self.stream))
                self._reduce(hist1, dist.ReduceOp.SUM, group)
                bin1, done, answer = torch.empty((mc, nt), **i32), torch.empty((mc, nt), **i32), torch.zeros((mc, nt), **i64)
                self.lib.check(lib.pem_qsel_decide1_dev(mc, nt, p(kmn), p(kmx), p(hist1), bins1, p(resid), p(bin1), p(done), p(answer), self.stream))
                hist2_local = torch.empty((mc, nt, bins2), **i32)
                self.lib.check(lib.pem_qsel_hist2_f64_dev(self.n, mc, self._data(c0), self.m, p(kmn), p(kmx), nt, p(bin1), bins1, bins2,
                                                          p(hist2_local), self.stream))
                _, on = _dist(group)
                hist2 = self._reduce(hist2_local.clone(), dist.ReduceOp.SUM, group) if on else hist2_local
                bin2, count, count_local = torch.empty((mc, nt), **i32), torch.empty((mc, nt), **i64), torch.empty((mc, nt), **i32)
                self.lib.check(lib.pem_qsel_decide2_dev(mc, nt, p(hist2), p(hist2_local), bins2, p(bin1), p(done), p(resid), p(bin2), p(count),
                                                        p(count_local), self.stream))
                need = int(self._reduce(count_local.max().to(torch.int64).reshape(1), dist.ReduceOp.MAX, group).item())
                if not _lists_fit(need, dist.get_world_size(group) if on else 1, mc, nt):
                    return None
                L = _list_len(need)
                cand, cursor = torch.empty((mc * nt, L), **i64), torch.empty(mc * nt, **i32)
                self.lib.check(lib.pem_qsel_compact_f64_dev(self.n, mc, self._data(c0), self.m, p(kmn), p(kmx), nt, p(bin1), p(bin2), p(done),
                                                            bins1, bins2, L, p(cand), p(cursor), self.stream))
                gathered, world = self._gather(cand, group)
                self.lib.check(lib.pem_qsel_select_dev(mc, nt, world, L, p(gathered), p(bin1), p(bin2), p(done), p(resid), p(answer), self.stream))
                out[c0:c0 + mc] = answer.cpu().numpy().view(U64)
        return out

    # -- the level loop (fallback) --
    def hist(self, klo, khi, bins):
        torch = self.torch
        nr = klo.shape[1]
        out = np.empty((self.m, nr, bins), dtype=np.int64)
        with torch.cuda.device(self.dev):
            for c0, mc in self._chunks():
                d_lo = torch.from_numpy(np.ascontiguousarray(klo[c0:c0 + mc]).view(np.int64)).to(self.dev)
                d_hi = torch.from_numpy(np.ascontiguousarray(khi[c0:c0 + mc]).view(np.int64)).to(self.dev)
                h = torch.empty((mc, nr, bins), dtype=torch.int32, device=self.dev)
                self.lib.check(self.lib.load().pem_range_hist_f64_dev(
                    self.n, mc, self._data(c0), self.m, nr, C.c_void_p(d_lo.data_ptr()), C.c_void_p(d_hi.data_ptr()), bins,
                    C.c_void_p(h.data_ptr()), self.stream))
                out[c0:c0 + mc] = h.cpu().numpy().view(np.uint32)
        return out

    def levels_pass(self, ranks, kmin, kmax, group, on_device: bool = True):
        if on_device and self.m <= MAX_COLUMNS:
            empty = kmin > kmax
            return _device_levels(self, ranks, np.where(empty, U64(1), kmin).astype(U64), np.where(empty, U64(0), kmax).astype(U64), group)
        return narrow_by_levels(self.hist, ranks, kmin, kmax, group, chunk_columns=min(self.m, MAX_COLUMNS))


def _device_levels(cols, ranks, klo0, khi0, group):
    """The level loop with ranges, counts and decisions resident on the device (`pem_range_hist_f64_dev` -> all-reduce ->
    `pem_range_narrow_dev`): per level one streaming pass, one collective on a device tensor and one flag read -- no histogram
    crosses PCIe.  `ranks`: (nr,) wanted ranks of this pass, klo0 / khi0: (m,) the columns' global key range; returns the (m, nr) keys.  One chunk of columns (m <= 256)."""
    import torch
    import torch.distributed as dist
    lib, dev, m, nr = cols.lib.load(), cols.flat.device, cols.m, ranks.size
    bins = level_bins(m, nr)
    klo = torch.from_numpy(np.repeat(klo0[:, None], nr, axis=1).view(np.int64).copy()).to(dev)
    khi = torch.from_numpy(np.repeat(khi0[:, None], nr, axis=1).view(np.int64).copy()).to(dev)
    resid = torch.from_numpy(np.broadcast_to(ranks, (m, nr)).astype(np.int64).copy()).to(dev)
    hist = torch.empty((m, nr, bins), dtype=torch.int32, device=dev)
    top = torch.tensor(-2 ** 63, dtype=torch.int64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())                                                       # noqa: E731
    with torch.cuda.device(dev):
        for _level in range(80):
            if not bool(((klo ^ top) < (khi ^ top)).any()):            # unsigned comparison of the keys
                break
            cols.lib.check(lib.pem_range_hist_f64_dev(cols.n, m, p(cols.flat), m, nr, p(klo), p(khi), bins, p(hist), cols.stream))
            cols._reduce(hist, dist.ReduceOp.SUM, group)
            cols.lib.check(lib.pem_range_narrow_dev(m * nr, bins, p(hist), p(klo), p(khi), p(resid), cols.stream))
        else:
            raise RuntimeError('percentile refinement did not converge')
    return klo.cpu().numpy().view(U64)


def column_percentiles_sharded(a, percentiles, group=None, method: str = 'select'):
    """`np.percentile(concatenation of every rank's `a`, percentiles, axis=0)`, bit for bit, on every rank: `a` is this rank's
    (n_local, ...) CUDA tensor (n_local may be 0 on some ranks).  Returns a numpy array (len(percentiles), ...).
    method 'select' (default): four streaming passes per three percentiles; 'levels': the level loop of rounds 1-2."""
    cols = DeviceColumns(a)          # (more than 256 columns: the kernels go through them 256 at a time)
    res = sharded_percentiles(cols, percentiles, group=group, method=method)
    return res.reshape(res.shape[:-1] + cols.trailing) if cols.trailing else res[..., 0]


def column_percentiles_numpy(a, percentiles, group=None, method: str = 'select'):
    """The same on this rank's rows as a numpy array, every stage in numpy (CPU rehearsals under gloo)."""
    cols = NumpyColumns(a)
    res = sharded_percentiles(cols, percentiles, group=group, method=method)
    return res.reshape(res.shape[:-1] + cols.trailing) if cols.trailing else res[..., 0]
