"""Sampling-loop drivers: forward UQ, data generation with NaN/IQR filtering, Sobol' sensitivity.

These follow the SHAPE of the reference's drivers:
  * `generate_data`, `process_compression`   scripts/gen_data.py:218-294 on a `system.PemV0System`
  * `generate_data_on_device`                the same data set, batched, without the system object
  * `filter_outputs`  scripts/gen_data.py:125-174  (pinned: tests/golden/filter_outputs.npz holds the reference's
                      own `_filter_outputs` results, extracted and run by tests/golden/make_golden.py)
  * `forward_uq`      scripts/pem_v0/monte_carlo.py:63-300  (Ns samples -> predict -> statistics)
  * `sobol_indices`   scripts/pem_v0/sobol.py:46-118  first-order + total indices (compute_s2=False)
The sampler / Sobol' estimator implementations of the reference are amisc / uqtils (third-party, absent): PARITY
UNPINNED for those two; the estimators used here are stated in the docstrings.

Everything between `Design.fill` and the statistics stays in HBM: inputs are generated on the device
(csrc/pem_sampler.hip), evaluated by one `pem_coupled_f64_dev` launch per batch, and reduced with torch.
"""
import functools

import numpy as np

from . import sampling
from .batch import QOI_NAMES, CoupledBatch
from .distributed import shard_bounds

COORDS_STR_ID = '_coords'      # amisc.typing.COORDS_STR_ID as the reference uses it (gen_data.py:143, plume.py:157)


# ------------------------------------------------------------------------------------------- percentiles over the samples
MAX_Q, MAX_Q_WIDE = 6, 3     # include/pem_hip.h PEM_QUANTILE_MAX_Q, PEM_QUANTILE_MAX_Q_WIDE: quantiles per selection (<= 128 / <= 256 columns)


def _linear_ranks(n: int, percentiles):
    """The two order statistics numpy's method 'linear' reads per percentile of n values, and its interpolation weight
    (numpy/lib/_function_base_impl.py: percentile -> _quantile): (rank_prev, rank_next) uint64, gamma float64 (read-only:
    a campaign loop asks for the same ones every call, so they are remembered)."""
    return _linear_ranks_cached(int(n), tuple(float(x) for x in np.atleast_1d(np.asarray(percentiles, dtype=np.float64))))


@functools.lru_cache(maxsize=64)
def _linear_ranks_cached(n: int, percentiles: tuple):
    q = np.true_divide(np.asarray(percentiles, dtype=np.float64), np.float64(100))
    if not np.all((q >= 0) & (q <= 1)):
        raise ValueError('Percentiles must be in the range [0, 100]')
    if n == 0:
        raise ValueError('no samples')
    virtual = (n - 1) * q
    prev = np.floor(virtual)
    nxt = prev + 1
    above = virtual >= n - 1
    prev[above], nxt[above] = -1, -1                       # "take the max value of the array": index -1
    gamma = virtual - prev                                  # (numpy takes the weight from the clipped index as well)
    rank_prev = np.where(prev < 0, n - 1, prev).astype(np.uint64)
    rank_next = np.where(nxt < 0, n - 1, nxt).astype(np.uint64)
    for a in (rank_prev, rank_next, gamma):
        a.setflags(write=False)
    return rank_prev, rank_next, gamma


def column_percentiles(a, percentiles):
    """`np.percentile(a, percentiles, axis=0)` (method 'linear') of a CUDA tensor `a` of shape (n, ...), bit for bit, by exact
    selection on the device (`pem_quantiles_f64_dev`, csrc/pem_quantile.hip) -- the percentiles of gen_data.py:125-174 and
    monte_carlo.py:363-658 at sizes where a sort-based quantile gives up (torch.quantile: 2^24 values per column).
    Returns a CUDA tensor of shape (len(percentiles), ...) (or (...) for a scalar percentile); a column that holds a NaN
    gives NaN, as numpy does."""
    import ctypes as C
    import torch
    from . import _lib
    scalar = np.ndim(percentiles) == 0
    n = int(a.shape[0])
    rank_prev, rank_next, gamma = _linear_ranks(n, percentiles)
    q = gamma
    a = a.double()
    # a (n, m) view whose columns are contiguous arrays -- e.g. `batch.qoi.T`, the [3][n] reduced-QoI tensor seen as (n, 3) --
    # is read in place (`pem_quantiles_strided_f64_dev`): the three scalar QoIs of a campaign in ONE selection, without a copy
    transposed = a.dim() == 2 and a.shape[1] > 1 and a.stride(0) == 1 and a.stride(1) >= n
    flat = a if transposed else a.reshape(n, -1)
    if not transposed and not flat.is_contiguous():
        flat = flat.contiguous()
    m = flat.shape[1]
    ld, cs = (1, flat.stride(1)) if transposed else (m, 1)
    out = torch.empty((q.size, m), dtype=torch.float64, device=flat.device)
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
    with torch.cuda.device(flat.device):
        for c0 in range(0, m, 256):                         # 256 columns and six quantiles (three for more than 128 columns) per call
            mc = min(256, m - c0)
            step = MAX_Q if mc <= 128 else MAX_Q_WIDE
            for i0 in range(0, q.size, step):
                rp, rn, gm = (np.ascontiguousarray(v[i0:i0 + step]) for v in (rank_prev, rank_next, gamma))
                # the call writes rows of mc values: straight into `out` when that is all of its columns
                part = out[i0:i0 + rp.size] if mc == m else torch.empty((rp.size, mc), dtype=torch.float64, device=flat.device)
                _lib.check(lib.pem_quantiles_strided_f64_dev(n, mc, C.c_void_p(flat.data_ptr() + 8 * c0 * cs), ld, cs, rp.size, C.c_void_p(rp.ctypes.data),
                                                             C.c_void_p(rn.ctypes.data), C.c_void_p(gm.ctypes.data), C.c_void_p(part.data_ptr()), stream))
                if mc != m:
                    out[i0:i0 + rp.size, c0:c0 + mc] = part
    out = out.reshape((q.size,) + tuple(a.shape[1:]))
    return out[0] if scalar else out


# ------------------------------------------------------------------------------------------- NaN / outlier masks
ROW_MASKS_MAX_M = 512       # include/pem_hip.h PEM_ROW_MASKS_MAX_M


def _row_masks(a, lo, hi, per_sample: int):
    """`np.any(np.isnan(a), axis=rest)` and `np.sum((a < lo) | (a > hi), axis=rest)` of a CUDA tensor (n, ...) against per-entry
    bounds, in one pass (`pem_row_masks_f64_dev`): (bool tensor (n,), int32 tensor (n,))."""
    import ctypes as C
    import torch
    from . import _lib
    n = int(a.shape[0])
    flat = a.reshape(n, per_sample)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    lo = lo.reshape(per_sample).contiguous()
    hi = hi.reshape(per_sample).contiguous()
    nan = torch.empty(n, dtype=torch.uint8, device=flat.device)
    count = torch.empty(n, dtype=torch.int32, device=flat.device)
    with torch.cuda.device(flat.device):
        stream = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
        _lib.check(_lib.load().pem_row_masks_f64_dev(n, per_sample, C.c_void_p(flat.data_ptr()), per_sample, C.c_void_p(lo.data_ptr()),
                                                     C.c_void_p(hi.data_ptr()), C.c_void_p(nan.data_ptr()), C.c_void_p(count.data_ptr()), stream))
    return nan.bool(), count


def _check_sharded_call(variables: dict, group=None):
    """Every rank of a sharded (collective) statistics call must bring the same variables, in the same order, with the same
    trailing shapes: all-gather a small header and raise on EVERY rank if they differ -- the alternative is a job that hangs in
    the first all-reduce whose sizes do not match."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
        return
    header = [(str(k), tuple(int(d) for d in v.shape[1:])) for k, v in variables.items()]
    seen = [None] * dist.get_world_size(group)
    dist.all_gather_object(seen, header, group=group)
    if any(h != seen[0] for h in seen):
        raise ValueError(f'sharded statistics: the ranks do not bring the same variables / trailing shapes: {seen}')


def filter_outputs(outputs: dict, iqr_factor: float = 1.5, group=None, sharded: bool = False):
    """NaN and interquartile-range outlier masks per output variable; mirrors gen_data.py:125-174.

    sharded (opt-in; the multi-rank drivers pass True): `outputs` is THIS rank's shard of the samples; p25 / p75 are then those
    of ALL ranks' samples (`percentiles.column_percentiles_sharded`, or its numpy restatement for numpy arrays) and the masks
    returned are those of the local samples -- together the masks the reference computes on the whole data set.  The call is
    then a COLLECTIVE: every rank of `group` must make it, with the same variables in the same order and the same trailing
    shapes (checked before the first pass: `_check_sharded_call`; a mismatch raises on every rank instead of hanging).
    The default treats `outputs` as the whole data set, as the reference does, whatever process group exists.

    `outputs`: {name: array (num_samples, ...)} of numeric numpy arrays or torch tensors (device tensors stay on
    the device).  Names containing '_coords' and the name 'errors' are skipped.  A sample is an outlier of a
    variable if MORE than int(0.75 * entries_per_sample) of its entries lie outside [p25 - q*iqr, p75 + q*iqr],
    with p25/p75 taken per entry over the samples -- always true for a scalar variable with one entry outside.
    Returns (nan_idx, outlier_idx): dicts of boolean arrays of shape (num_samples,)."""
    nan_idx, outlier_idx = {}, {}
    cnt_thresh = 0.75
    if sharded:
        _check_sharded_call({k: v for k, v in outputs.items() if COORDS_STR_ID not in str(k) and str(k) != 'errors'}, group)
    for var, arr in outputs.items():
        if COORDS_STR_ID in str(var) or str(var) == 'errors':
            continue
        if type(arr).__module__.split('.')[0] == 'torch':
            import torch
            a = arr.double()
            rest = tuple(range(1, a.dim()))
            per_sample = int(np.prod(a.shape[1:])) if a.dim() > 1 else 1
            if sharded:
                from .percentiles import column_percentiles_numpy, column_percentiles_sharded
                q = column_percentiles_sharded(a, [25.0, 75.0], group=group) if a.is_cuda else \
                    column_percentiles_numpy(a.numpy(), [25.0, 75.0], group=group)
                q = torch.from_numpy(np.ascontiguousarray(q)).to(a.device)
            else:
                q = column_percentiles(a, [25.0, 75.0]) if a.is_cuda else torch.quantile(a, torch.tensor([0.25, 0.75], dtype=a.dtype), dim=0)
            iqr = q[1] - q[0]
            lo, hi = q[0] - iqr_factor * iqr, q[1] + iqr_factor * iqr
            if a.is_cuda and a.shape[0] > 0 and per_sample <= ROW_MASKS_MAX_M:
                nan_idx[var], count = _row_masks(a, lo, hi, per_sample)      # one pass over the array (csrc/pem_masks.hip)
            else:
                nan_idx[var] = torch.isnan(a).any(dim=rest) if rest else torch.isnan(a)
                outside = (a < lo) | (a > hi)
                count = outside.sum(dim=rest) if rest else outside.long()
            outlier_idx[var] = count > int(cnt_thresh * per_sample)
        else:
            a = np.asarray(arr, dtype=np.float64)
            rest = tuple(range(1, a.ndim))
            nan_idx[var] = np.any(np.isnan(a), axis=rest)
            if sharded:
                from .percentiles import column_percentiles_numpy
                p25, p75 = column_percentiles_numpy(a, [25.0, 75.0], group=group)
            else:
                p25, p75 = np.percentile(a, 25, axis=0), np.percentile(a, 75, axis=0)
            iqr = p75 - p25
            outside = (a < p25 - iqr_factor * iqr) | (a > p75 + iqr_factor * iqr)
            outlier_idx[var] = np.sum(outside, axis=rest) > int(cnt_thresh * np.prod(a.shape[1:]))
    return nan_idx, outlier_idx


def discard_mask(nan_idx: dict, outlier_idx: dict, discard_outliers: bool = False):
    """Samples to drop: any NaN always, outliers on request (gen_data.py:177-215)."""
    masks = list(nan_idx.values()) + (list(outlier_idx.values()) if discard_outliers else [])
    out = masks[0].clone() if hasattr(masks[0], 'clone') else masks[0].copy()
    for m in masks[1:]:
        out |= m
    return out


# ------------------------------------------------------------------------------------------- forward evaluation loops
def forward_uq(n: int, seed: int = 0, method: str = 'mc', profile: bool = False, batch_size: int | None = None,
               priors=None, device=None, keep_profile: bool = False, rank: int = 0, world: int = 1, streams: int = 1,
               keep_inputs: bool = True):
    """Forward propagation of the PEM-v0 priors through cathode -> thruster (test double) -> plume.

    Draws global samples [0, n) of the counter-based design (this rank evaluates its contiguous shard), evaluates
    them and returns per-sample QoIs of the shard as CUDA tensors:
    `V_cc, div_angle, T_c, I_B0, T, invalid`, the inputs `x` ([15][n_local]) and, if `keep_profile`, `j_ion`.
    `batch_size` (default: the whole shard in ONE launch; 1e7 samples with the profile: 1.51 ms against 1.61 in launches of 2^21
    and 1.80 of 2^20, without it 0.66 / 0.70 / 0.77 -- every launch pays its own ramp and tail, tools/forward_uq_batch_probe.py)
    cuts the shard into launches over ranges of the resident batch, e.g. to interleave other work on the stream.
    `keep_inputs=False` (method 'mc' only): the generated inputs are not written out and `x` is left out of the result -- a
    campaign that only looks at the outputs saves the 120 bytes per sample (1e7 samples: 1.54 -> 1.44 ms with the profile, 0.64 -> 0.58 without).
    `streams` > 1 deals the launches of a shard of several batches onto that many side streams (they write disjoint ranges; the side
    streams begin after, and the caller's stream continues after, everything enqueued here).  It is NOT the default: what gains
    3-7 % for the evaluate-only launches of bench.py loses 8-20 % here (1e7 samples: 1.77 -> 1.92 ms with the profile, 0.71 -> 0.86 ms
    without) -- the fused Monte-Carlo launches are persistent grids sized to fill the chip, and two of them dispatched side by side
    each run as two passes (tools/forward_uq_streams_probe.py, profiles/launch_amortisation_r03.txt)."""
    import torch
    design = sampling.Design(priors=priors, seed=seed)
    lo, hi = shard_bounds(n, world, rank)
    n_local = hi - lo
    # ONE resident batch holds the shard's inputs and outputs; the launches write ranges of it in place (a scratch batch
    # plus per-batch copies into the result arrays cost more HBM traffic than the reduced-QoI kernel itself)
    batch = CoupledBatch(n_local, device=device, profile=profile or keep_profile)
    bs = n_local if batch_size is None else max(64, min(int(batch_size), n_local)) & ~1   # ranges start at even samples (16-byte aligned profile rows)
    bs = max(bs, 1)
    caller = torch.cuda.current_stream(batch.device)
    side = None
    if streams > 1 and n_local > bs:
        side = [torch.cuda.Stream(device=batch.device) for _ in range(int(streams))]
        begin = caller.record_event()
        for st in side:
            st.wait_event(begin)
    for i, off in enumerate(range(0, n_local, bs)):
        m = min(bs, n_local - off)
        st = side[i % len(side)] if side else caller
        if method == 'mc':      # fused: the inputs are generated inside the evaluation kernel (and stored for `x`)
            batch.run_mc(design, first_index=lo + off, write_inputs=keep_inputs, first=off, count=m, stream=st)
        else:
            design.fill(batch.inputs[:, off:off + m], first_index=lo + off, method=method, n_total=n, stream=st)
            batch.run(first=off, count=m, stream=st)
    if side:
        for st in side:
            caller.wait_event(st.record_event())
    out = {k: batch.qoi[i] for i, k in enumerate(QOI_NAMES)}
    out.update(I_B0=batch.I_B0, T=batch.T, invalid=batch.invalid.bool())
    if keep_inputs or method != 'mc':
        out['x'] = batch.inputs
    if keep_profile:
        out['j_ion'] = batch.j_ion
    return out


def percentile_bands(outputs: dict, percentiles=(5.0, 50.0, 95.0), names=('V_cc', 'div_angle', 'T_c', 'j_ion'), group=None,
                     sharded: bool = False):
    """The 5 / 50 / 95 % bands monte_carlo.py:363-658 draws from its prior / posterior predictive samples
    (`np.percentile(ys, x, axis=0)`), for the device-resident outputs of `forward_uq`: {name: (len(percentiles), ...) CUDA
    tensor}, equal to numpy's values bit for bit (`column_percentiles`).  Samples flagged `invalid` are kept, as there.

    sharded (opt-in; a collective -- every rank of `group` must call with the same names and shapes, see `filter_outputs`):
    `outputs` is THIS rank's shard of the campaign (`forward_uq(..., rank=, world=)`) and the bands are those of ALL ranks'
    samples, the same on every rank -- `percentiles.column_percentiles_sharded`: min / max and two histograms all-reduced, a few
    hundred candidates per wanted rank all-gathered, nothing else crosses xGMI."""
    import torch
    if not sharded:
        return {k: column_percentiles(outputs[k], list(percentiles)) for k in names if k in outputs}
    _check_sharded_call({k: outputs[k] for k in names if k in outputs}, group)
    from .percentiles import column_percentiles_sharded
    return {k: torch.from_numpy(np.ascontiguousarray(column_percentiles_sharded(outputs[k], list(percentiles), group=group))).to(outputs[k].device)
            for k in names if k in outputs}


def _stacked_rows(tensors):
    """1-D CUDA tensors that are equally spaced rows of ONE allocation -- `forward_uq`'s V_cc, div_angle, T_c are rows of the
    batch's [3][n] reduced-QoI tensor -- as an (n, k) strided view (no copy), or None."""
    import torch
    t0 = tensors[0]
    if len(tensors) < 2 or any(t.dim() != 1 or not t.is_cuda or t.dtype != torch.float64 or t.numel() != t0.numel() or t.stride(0) != 1
                               or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr() for t in tensors):
        return None
    n, step = t0.numel(), tensors[1].storage_offset() - t0.storage_offset()
    if step < n or any(t.storage_offset() != t0.storage_offset() + i * step for i, t in enumerate(tensors)):
        return None
    return torch.as_strided(t0, (n, len(tensors)), (1, step))


def campaign_statistics(outputs: dict, iqr_factor: float = 1.5, percentiles=(5.0, 50.0, 95.0), names=None):
    """`filter_outputs` and `percentile_bands` of one campaign's device-resident outputs from ONE selection per variable
    (gen_data.py:125-174 and monte_carlo.py:363-658 want p25 / p75 and 5 / 50 / 95 % of the same arrays): the five quantiles
    share the pilot, the counting pass and the copy pass (`pem_quantiles_f64_dev`, up to six per call), and scalar outputs that
    are rows of one tensor go through one call together.  Returns (nan_idx, outlier_idx, bands), each equal to what the two
    separate calls return (numpy's values bit for bit)."""
    import torch
    names = [k for k in (outputs if names is None else names) if k in outputs and COORDS_STR_ID not in str(k) and str(k) != 'errors']
    pct = [25.0, 75.0] + [float(x) for x in percentiles]
    q = {}
    scalars = [k for k in names if outputs[k].dim() == 1]
    stacked = _stacked_rows([outputs[k] for k in scalars]) if len(scalars) > 1 else None
    if stacked is not None:
        qs = column_percentiles(stacked, pct)                               # (len(pct), k)
        q.update({k: qs[:, i] for i, k in enumerate(scalars)})
    for k in names:
        if k not in q:
            q[k] = column_percentiles(outputs[k], pct)
    nan_idx, outlier_idx, bands = {}, {}, {}
    for k in names:
        a, qk = outputs[k].double(), q[k]
        per_sample = int(np.prod(a.shape[1:])) if a.dim() > 1 else 1
        iqr = qk[1] - qk[0]
        lo, hi = qk[0] - iqr_factor * iqr, qk[1] + iqr_factor * iqr
        if a.shape[0] > 0 and per_sample <= ROW_MASKS_MAX_M:
            nan_idx[k], count = _row_masks(a, lo, hi, per_sample)
        else:
            rest = tuple(range(1, a.dim()))
            nan_idx[k] = torch.isnan(a).any(dim=rest) if rest else torch.isnan(a)
            outside = (a < lo) | (a > hi)
            count = outside.sum(dim=rest) if rest else outside.long()
        outlier_idx[k] = count > int(0.75 * per_sample)
        bands[k] = qk[2:]
    return nan_idx, outlier_idx, bands


FUSED_STATS_MIN_N = 4096      # include/pem_hip.h PEM_MC_STATS_MIN_N


def forward_uq_statistics(n: int, seed: int = 0, keep_profile: bool = True, percentiles=(5.0, 50.0, 95.0), iqr_factor: float = 1.5,
                          priors=None, device=None, keep_inputs: bool = False, fused: bool = True):
    """One forward-UQ campaign of scripts/pem_v0/monte_carlo.py:63-300 / gen_data.py:218-258 on one GPU, statistics included:
    `forward_uq(n, seed, method='mc')`, the NaN / IQR masks of `filter_outputs` and the bands of `percentile_bands` -- with the
    percentiles of the 91-point profile COUNTED WHERE THE PROFILE IS PRODUCED (`pem_coupled_mc_stats_f64_dev`, csrc/pem_qfused.h):
    the samples of the first 3 % are evaluated and bracket the wanted ranks, one launch evaluates everything and counts every
    profile value against the brackets on chip, the order statistics come from the 4 % of values inside them.  The profile is
    not read back for its percentiles (five reads of 7.3 GB per 1e7 samples in round 3; one, for the masks, now) and with
    `keep_profile=False` it is never written at all.

    Returns the dictionary of `forward_uq` plus 'nan_idx', 'outlier_idx' (per output variable, as `filter_outputs`, the profile's
    included whether it is kept or not) and 'bands' ({name: (len(percentiles), ...)} as `percentile_bands`), every number equal to
    numpy's on the same samples bit for bit.  'fused': whether the on-chip selection answered (heavy ties or a non-finite
    profile value make it decline: the percentiles then come from passes over the stored profile -- evaluated again with the
    profile kept when `keep_profile=False`); 'premasked': whether the profile's outlier mask was counted by the evaluation launch
    too (against intervals for the IQR bounds, the few undecided samples settled afterwards) instead of by a pass over the profile."""
    import ctypes as C
    import torch
    from . import _lib, constants
    design = sampling.Design(priors=priors, seed=seed)
    batch = CoupledBatch(n, device=device, profile=keep_profile)
    pct = [25.0, 75.0] + [float(x) for x in percentiles]
    if len(pct) > MAX_Q:
        raise ValueError(f'at most {MAX_Q - 2} percentiles besides the quartiles')
    answered, premasked, qj, qs, certain, uncertain = False, False, None, None, None, None
    thresh = int(0.75 * _lib.NANGLE)
    nv, cap = len(QOI_NAMES), 65536
    dev = batch.device
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        # (what the masks pass below writes is allocated before the evaluation is enqueued: nothing between the two calls but the call)
        n_pad = (n + 3) // 4 * 4                                            # (rows on 4-byte boundaries: the pass writes words)
        nan_s = torch.empty((nv + 1, n_pad), dtype=torch.bool, device=dev)[:, :n]      # rows: the scalar outputs, then the profile
        out_s = torch.empty((nv + 1, n_pad), dtype=torch.bool, device=dev)[:, :n]
        open_rows = torch.empty(cap, dtype=torch.int64, device=dev)
        open_count = torch.zeros(1, dtype=torch.int32, device=dev)
        if fused and n >= FUSED_STATS_MIN_N:
            rp, rn, gm = _linear_ranks(n, pct)
            qj = torch.empty((len(pct), _lib.NANGLE), dtype=torch.float64, device=dev)
            # (the scalar QoIs' percentiles are selected by the same call, on a second stream, while the profile's records are sorted)
            qs = torch.empty((len(pct), nv), dtype=torch.float64, device=dev)
            pilot = None if keep_profile else torch.empty(((n + 31) // 32, _lib.NANGLE), dtype=torch.float64, device=dev)
            certain = torch.empty(n, dtype=torch.uint8, device=dev)
            uncertain = torch.empty(n, dtype=torch.uint8, device=dev)
            ok, pm_ok = C.c_int(0), C.c_int(0)
            ptr = lambda arr: C.c_void_p(arr.ctypes.data)                                       # noqa: E731
            outs = batch._out_ptrs
            _lib.check(_lib.load().pem_coupled_mc_stats_f64_dev(
                n, 0, design.seed, design.stream, ptr(design.kind), ptr(design.a), ptr(design.b), constants.TORR_2_PA, batch.radius,
                C.c_void_p(batch.inputs.data_ptr()) if keep_inputs else None, batch.inputs.stride(0),
                outs[0], outs[1], outs[2], outs[3], None if pilot is None else C.c_void_p(pilot.data_ptr()), outs[4], outs[5], outs[6],
                len(pct), ptr(rp), ptr(rn), ptr(gm), C.c_void_p(qj.data_ptr()), C.c_void_p(qs.data_ptr()), C.byref(ok),
                0, 1, float(iqr_factor), C.c_void_p(certain.data_ptr()), C.c_void_p(uncertain.data_ptr()), C.byref(pm_ok),
                C.c_void_p(stream.cuda_stream)))
            answered, premasked = bool(ok.value), bool(pm_ok.value)
            del pilot
        else:
            batch.run_mc(design, first_index=0, write_inputs=keep_inputs)
    out = {k: batch.qoi[i] for i, k in enumerate(QOI_NAMES)}
    out.update(I_B0=batch.I_B0, T=batch.T, invalid=batch.invalid.view(torch.bool))            # (bytes 0 / 1: the same storage)
    if keep_inputs:
        out['x'] = batch.inputs
    if keep_profile:
        out['j_ion'] = batch.j_ion
    full = None
    if not answered:                         # the passes over the stored profile
        if keep_profile:
            qj = column_percentiles(batch.j_ion, pct)
        else:
            full = CoupledBatch(n, device=device, profile=True, thruster_qoi=False)
            full.run_mc(design, first_index=0)
            qj = column_percentiles(full.j_ion, pct)
    if qs is None:
        qs = column_percentiles(batch.qoi.T, pct)                  # the three scalar QoIs in one selection
    q = {k: qs[:, i] for i, k in enumerate(QOI_NAMES)}
    q['j_ion'] = qj
    nan_idx, outlier_idx, bands = {}, {}, {}
    # the scalar outputs' masks (a variable of one entry per sample is an outlier when that entry lies outside p25 - f iqr .. p75 + f iqr)
    # and the verdict of the profile's premask counts: one pass, one thread per sample (`pem_campaign_masks_f64_dev`)
    dp = lambda t: C.c_void_p(t.data_ptr())                                                       # noqa: E731
    with torch.cuda.device(dev):
        vars_ = (C.c_void_p * nv)(*[batch.qoi[i].data_ptr() for i in range(nv)])
        _lib.check(_lib.load().pem_campaign_masks_f64_dev(
            n, nv, vars_, dp(qs), qs.stride(0), 0, 1, float(iqr_factor), dp(nan_s), dp(out_s), nan_s.stride(0), dp(certain) if premasked else None,
            dp(uncertain) if premasked else None, thresh, dp(open_rows), dp(open_count), cap, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    for i, k in enumerate(QOI_NAMES):
        nan_idx[k], outlier_idx[k] = nan_s[i], out_s[i]
    # the profile's masks (gen_data.py:160-168): NaN per sample, more than int(0.75 * 91) values outside [p25 - f iqr, p75 + f iqr]

    def profile_bounds():
        iqr = qj[1] - qj[0]
        return qj[0] - iqr_factor * iqr, qj[1] + iqr_factor * iqr
    if premasked:
        # counted by the evaluation launch against INTERVALS for the bounds: a sample is settled unless its uncertain values could
        # change the verdict -- those few (none, as a rule) are looked at again with the exact bounds
        n_open = int(open_count.item())
        if n_open > cap:                                                  # (a list that did not fit: the same test in torch)
            c, u = certain.to(torch.int32), uncertain.to(torch.int32)
            rows_open = torch.nonzero((c <= thresh) & (c + u > thresh)).flatten()
        else:
            rows_open = open_rows[:n_open]
        outl = out_s[nv]
        if rows_open.numel():
            lo, hi = profile_bounds()
            rows = batch.j_ion[rows_open] if keep_profile else _profile_rows(design, rows_open, device=dev)
            outl[rows_open] = ((rows < lo) | (rows > hi)).sum(dim=1) > thresh
        nan_idx['j_ion'] = nan_s[nv]                                                 # (zeros: a non-finite value makes the selection decline)
        outlier_idx['j_ion'] = outl
    else:
        lo, hi = profile_bounds()
        prof = batch.j_ion if keep_profile else (full.j_ion if full is not None else None)
        if prof is None:                     # answered on chip but no premask (bounds it cannot take): the profile once, for the masks
            full = CoupledBatch(n, device=device, profile=True, thruster_qoi=False)
            full.run_mc(design, first_index=0)
            prof = full.j_ion
        nan_idx['j_ion'], count = _row_masks(prof, lo, hi, _lib.NANGLE)
        outlier_idx['j_ion'] = count > thresh
    del full
    for k in q:
        bands[k] = q[k][2:]
    out.update(nan_idx=nan_idx, outlier_idx=outlier_idx, bands=bands, fused=answered, premasked=premasked)
    return out


def _profile_rows(design, indices, device=None):
    """The profile rows of global samples `indices` (a CUDA index tensor) of the Monte-Carlo design, evaluated one by one -- the few
    samples a campaign without a stored profile has to look at again."""
    import torch
    rows = []
    one = CoupledBatch(2, device=device, profile=True, thruster_qoi=False)
    for i in indices.tolist():
        one.run_mc(design, first_index=int(i), count=1)
        rows.append(one.j_ion[0].clone())
    return torch.stack(rows)


def generate_data(system, description: str, num_samples: int = 500, executor=None, verbose: bool = False,
                  iqr_factor: float = 1.5, device_resident: bool = False):
    """gen_data.py:218-258, same signature, on a `system.PemV0System`: sample the input space (calibration and
    nuisance variables from their pdfs), evaluate the true models, normalise the outputs as their variables declare,
    compute the NaN / IQR masks and pickle `{description: (samples, outputs), 'nan_idx', 'outlier_idx', 'iqr_factor'}`
    to `system.root_dir/description/description.pkl` (when the system has a root_dir).

    device_resident: keep samples and outputs as CUDA tensors (no host copy, no pickle) -- for sizes where the
    0.75 KB per sample of profile should not cross PCIe; `j_ion_coords` is then omitted."""
    import os
    import pickle
    from .system import COORDS_STR_ID, to_model_dataset
    if getattr(system, 'logger', None) is not None:
        system.logger.info(f'Generating {description} data for {system.name} -- {num_samples} samples...')
    if system.root_dir is not None and not device_resident:
        os.mkdir(system.root_dir / description)                      # raises if it exists, as the reference does
    samples = system.sample_inputs(num_samples, normalize=True, use_pdf=['calibration', 'nuisance'],
                                   as_tensor=device_resident)
    outputs = system.predict(samples, use_model='best', model_dir=None if system.root_dir is None else
                             system.root_dir / description, executor=executor, verbose=verbose)
    if device_resident:
        outputs = {k: v for k, v in outputs.items() if not k.endswith(COORDS_STR_ID)}
    samples, coords = to_model_dataset(samples, system.inputs())
    samples.update(coords)
    norm_outputs = {var.name: var.normalize(outputs[var.name]) for var in system.outputs() if var in outputs}
    nan_idx, outlier_idx = filter_outputs(norm_outputs, iqr_factor=iqr_factor)
    dump = {description: (samples, outputs), 'nan_idx': nan_idx, 'outlier_idx': outlier_idx, 'iqr_factor': iqr_factor}
    if system.root_dir is not None and not device_resident:
        with open(system.root_dir / description / f'{description}.pkl', 'wb') as fd:
            pickle.dump(dump, fd)
    return dump


def process_compression(system, data: dict, discard_outliers: bool = False):
    """gen_data.py:261-294: compute the SVD maps of the field outputs from the 'compression' data set and save the
    system.  NaN samples are always dropped, IQR outliers only with `discard_outliers`."""
    from .system import COORDS_STR_ID
    outputs = data['compression'][1]
    discard = discard_mask(data['nan_idx'], data['outlier_idx'], discard_outliers=discard_outliers)
    keep = ~discard
    for var in system.outputs():
        if var.compression is None:
            continue
        coords = outputs.get(f'{var}{COORDS_STR_ID}')
        if coords is not None:
            var.compression.coords = coords[0]                        # gen_data.py:281: all coords are the same
        if var.compression.method.lower() != 'svd':
            raise ValueError(f"Compression method '{var.compression.method}' not supported.")
        var.compression.compute_map({f: var.normalize(outputs[f][keep]) for f in var.compression.fields})
    if system.root_dir is not None:
        (system.root_dir / 'compression').mkdir(exist_ok=True)
        system.save_to_file(f'{system.name}_compression.pkl', system.root_dir / 'compression')
    return system


def generate_data_on_device(n: int, seed: int = 0, description: str = 'test_set', method: str = 'mc',
                            iqr_factor: float = 1.5, batch_size: int | None = None, device=None):
    """`generate_data` without the system object, batched (fused sampling + evaluation kernel, any n): returns the
    same dictionary layout with CUDA tensors."""
    import torch
    res = forward_uq(n, seed=seed, method=method, profile=True, keep_profile=True, batch_size=batch_size, device=device)
    design = sampling.Design(seed=seed)
    samples = design.as_dict(res['x'])
    outputs = {k: res[k] for k in ('V_cc', 'I_B0', 'T', 'j_ion', 'div_angle', 'T_c')}
    norm = dict(outputs)
    norm['j_ion'] = torch.log10(outputs['j_ion'])
    nan_idx, outlier_idx = filter_outputs(norm, iqr_factor=iqr_factor)
    return {description: (samples, outputs), 'nan_idx': nan_idx, 'outlier_idx': outlier_idx, 'iqr_factor': iqr_factor}


# ------------------------------------------------------------------------------------------- surrogate training
def train_surrogate(system, fidelity: str = 'multi', **fit_kwargs):
    """scripts/fit_surr.py:101-200 on a `system.PemV0System`: train the surrogate with `system.fit(num_refine=1000, ...)`, read the
    allocation (`get_allocation()`'s 4-tuple, per (component, alpha)) and the test-error history, for the multi-fidelity run, the
    single-fidelity run (components' `model_fidelity` emptied, `system.clear()` first), or both (the second under
    `root_dir / 'amisc_single_fidelity'`).  The coupled PEM-v0 graph has ONE fidelity here -- the reference's only multi-fidelity
    component is the Julia thruster, out of scope -- so the two runs train the same surrogate; the shape of the call and of what
    comes back is the reference's.  Returns {'multi' | 'single': {'cost_alloc', 'model_cost', 'overhead_cost', 'model_evals',
    'train_history', 'test_error' ([iterations][targets]), 'highest_cost'}}; plotting is left out (fit_surr.py:162-200)."""
    import copy
    if fidelity not in ('multi', 'single', 'both'):
        raise ValueError("fidelity must be 'multi', 'single' or 'both'")
    fit_kwargs = dict(dict(num_refine=1000), **fit_kwargs)
    for k in ('estimate_bounds', 'update_bounds', 'plot_interval'):      # amisc options without a counterpart (accepted, unused)
        fit_kwargs.pop(k, None)
    targets = fit_kwargs.get('targets', None)
    results = {}

    def one_run(tag):
        system.fit(**fit_kwargs)
        system.plot_allocation()
        cost_alloc, model_cost, overhead, evals = system.get_allocation()
        history = copy.deepcopy(system.train_history)
        live = evals[np.nonzero(evals)]
        if getattr(system, 'logger', None) is not None and live.size:
            system.logger.info(f'Minimum model evaluations per iteration: {np.min(live):.2f}')
            system.logger.info(f'Average model evaluations per iteration: {np.mean(live):.2f}')
            system.logger.info(f'Maximum model evaluations per iteration: {np.max(live):.2f}')
        names = list(targets or (history[-1].get('test_error') or {}).keys())
        test = np.full((len(history), len(names)), np.nan)
        for j, res in enumerate(history):
            for i, var in enumerate(names):
                if res.get('test_error') is not None and var in res['test_error']:
                    test[j, i] = res['test_error'][var]
        # the cost of one evaluation at the highest fidelity, per component, summed (fit_surr.py:134-139)
        highest = sum(max(system[c.name].model_costs.values()) for c in system.components if c.name in cost_alloc)
        results[tag] = {'cost_alloc': cost_alloc, 'model_cost': model_cost, 'overhead_cost': overhead, 'model_evals': evals,
                        'train_history': history, 'test_error': test, 'targets': names, 'highest_cost': highest}

    if fidelity in ('multi', 'both'):
        one_run('multi')
    if fidelity in ('single', 'both'):
        system.clear()
        for comp in system.components:
            comp.model_fidelity = ()
        if fidelity == 'both' and system.root_dir is not None:
            system.root_dir = system.root_dir / 'amisc_single_fidelity'
        one_run('single')
    return results


# ------------------------------------------------------------------------------------------- Sobol' indices
def sobol_indices(n_base: int, seed: int = 0, qois=QOI_NAMES, priors=None, fixed: dict | None = None,
                  batch_size: int = 1 << 20, device=None, group=None, precision: str = 'fp64', fused: bool | None = None):
    """First-order and total Sobol' indices of scalar QoIs by the Saltelli design: N (d + 2) evaluations for the
    d non-constant inputs (matrices A, B and A with column i from B), `compute_s2=False` as sobol.py:113 asks.

    Estimators (Saltelli et al. 2010, Table 2; Jansen 1999), with f0/Var from the pooled A and B evaluations:
        S1_i = mean( f(B) * (f(AB_i) - f(A)) ) / Var          ST_i = mean( (f(A) - f(AB_i))^2 ) / (2 Var)
    `fixed` pins inputs (e.g. operating conditions at nominal, as sobol.py:104 does) -- they are not varied.
    With a torch.distributed `group` every rank evaluates its shard of the N base samples and the sums are
    all-reduced (O(d * n_qoi) doubles).  Returns {'S1': {qoi: [d]}, 'ST': ..., 'inputs': names}.

    fused (default True): the whole shard in ONE launch that keeps design rows, QoIs and estimator terms on chip and
    accumulates the sums in fp64 (`pem_saltelli_f64_dev` / `pem_saltelli_f32_dev`, csrc/pem_saltelli.hip); the result then
    also carries `non_physical` / `invalid` counts (the thruster filter of thruster.py:490-493 and plume.py:105 over all
    evaluations).  fused=False (fp64 only) is the block-by-block driver: d + 2 fused sample+evaluate launches and as many
    partial-sum launches per batch.
    precision='fp32' (BASELINE configs[4]: "fp64 -> fp32 mixed with tolerance check"): the SAME design -- the fp64
    counter-based rows, rounded to float -- through the fp32-arithmetic model.  fp32.compare_with_fp64 is the per-QoI
    tolerance report of that model."""
    import torch
    import torch.distributed as dist
    pri = dict(sampling.PEM_V0_PRIORS if priors is None else priors)
    for k, v in (fixed or {}).items():
        pri[k] = sampling.Prior(sampling.UNIFORM, float(v), float(v), 'fixed')
    design = sampling.Design(priors=pri, seed=seed)
    varied = [i for i, k in enumerate(design.names) if k not in (fixed or {})]
    world = dist.get_world_size(group) if (group is not None or dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_bounds(n_base, world, rank)
    bs = max(64, min(batch_size, max(hi - lo, 1)))
    nq, nd = len(qois), len(varied)
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    acc = torch.zeros((3 + 2 * nd, nq), dtype=torch.float64, device=dev)   # sum f, sum f^2, count | S1 sums | ST sums

    rows = [{'V_cc': 0, 'div_angle': 1, 'T_c': 2}.get(k) for k in qois]
    if any(r is None for r in rows):
        raise ValueError(f'sobol_indices handles the scalar QoIs V_cc, div_angle, T_c; got {qois}')
    if precision not in ('fp64', 'fp32'):
        raise ValueError("precision must be 'fp64' or 'fp32'")
    fused = True if fused is None else bool(fused)
    if precision == 'fp32' and not fused:
        raise ValueError('the fp32 model exists in the fused launch only')
    counts = None
    if fused:
        from .fp32 import saltelli_sums
        counts = torch.zeros(2, dtype=torch.int64, device=dev)
        if hi > lo:
            sums, counts = saltelli_sums(design, varied, hi - lo, first_index=lo, device=dev, precision=precision)
            sums = sums[:, rows]
            acc[0], acc[1], acc[2] = sums[0], sums[1], 2 * (hi - lo)
            acc[3:3 + nd], acc[3 + nd:] = sums[2::2], sums[3::2]
        lo = hi                                                        # nothing left for the block loop below
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    NB = 1024                                                          # workgroups (= deterministic partial sums) per pass
    ptr = lambda x: C.c_void_p(x.data_ptr())                           # noqa: E731
    all_rows = rows == [0, 1, 2]
    # Per block one fused sample+evaluate launch and one partial-sum launch, nothing else: at 1e6-sample batches the
    # GPU needs ~0.1 ms per block, and every extra torch op in this loop (copies, per-block reductions) costs the
    # host about as much as a block costs the device.  The three Saltelli blocks of a step live in three batches, the
    # kernels write their QoIs where the estimator reads them, and the partial sums of all blocks of a batch are reduced
    # together.
    batches, partial, mlast = None, None, -1

    def block(b, first, swap):
        b.run_mc(design, first_index=first, swap_dim=swap)             # Saltelli block generated inside the kernel
        return b.qoi if all_rows else b.qoi[rows].contiguous()         # [nq][m]

    def sums(slot, fA, fB, fAB, m):
        _lib.check(lib.pem_sobol_partial_f64_dev(m, nq, fA.stride(0), ptr(fA), ptr(fB), ptr(fAB) if fAB is not None else None,
                                                 ptr(partial[slot]), NB, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))

    for off in range(lo, hi, bs):
        m = min(bs, hi - off)
        if m != mlast:
            batches = [CoupledBatch(m, device=dev, profile=False, thruster_qoi=False) for _ in range(3)]
            partial = torch.empty((nd + 1, NB, nq, 2), dtype=torch.float64, device=dev)
            mlast = m
        fA = block(batches[0], off, -1)
        fB = block(batches[1], off, -2)
        sums(0, fA, fB, None, m)
        for j, d in enumerate(varied):
            fAB = block(batches[2], off, d)
            sums(1 + j, fA, fB, fAB, m)
        s = partial.sum(dim=1)                                         # [nd + 1][nq][2], one reduction per batch
        acc[0] += s[0, :, 0]
        acc[1] += s[0, :, 1]
        acc[2] += 2 * m
        acc[3:3 + nd] += s[1:, :, 0]
        acc[3 + nd:] += s[1:, :, 1]
    if world > 1:
        dist.all_reduce(acc, group=group)
        if counts is not None:
            dist.all_reduce(counts, group=group)
    cnt = acc[2]
    mean = acc[0] / cnt
    var = acc[1] / cnt - mean * mean
    n_tot = cnt / 2
    S1 = acc[3:3 + nd] / n_tot / var
    ST = acc[3 + nd:] / n_tot / (2 * var)
    names = [design.names[d] for d in varied]
    res = {'S1': {q: S1[:, i] for i, q in enumerate(qois)}, 'ST': {q: ST[:, i] for i, q in enumerate(qois)},
           'inputs': names, 'mean': {q: mean[i] for i, q in enumerate(qois)}, 'var': {q: var[i] for i, q in enumerate(qois)},
           'evaluations': int(n_base) * (nd + 2), 'precision': precision, 'fused': fused}
    if counts is not None:
        res['non_physical'], res['invalid'] = int(counts[0]), int(counts[1])
    return res


# ------------------------------------------------------------------------------------------- rejection of plume spikes
PLUME_INPUTS = ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')


def sample_plume_without_spikes(n: int, seed: int = 0, threshold: float = 200.0, I_B0: float = 4.0, radius: float = 1.0,
                                priors=None, max_rounds: int = 64, device=None):
    """Plume inputs whose profile stays below `threshold` A/m^2 everywhere: the rejection loop of the reference's Sobol'
    sampler (scripts/pem_v0/sobol.py:50-66: evaluate the plume at r = 1 m, I_B0 = 4 A, redraw every sample with any
    j >= 200 until none is left).

    Deterministic: attempt k of global sample i is draw i of stream 2k of the counter-based design, so the accepted
    set does not depend on batching.  Returns ([8][n] CUDA tensor in PLUME_INPUTS order, number of rounds)."""
    import torch
    from .models.plume import current_density
    pri = sampling.PEM_V0_PRIORS if priors is None else priors
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    x = sampling.Design(priors=pri, names=PLUME_INPUTS, seed=seed, stream=0).sample(n, device=dev)
    pending = torch.arange(n, device=dev)
    ib0 = torch.full((1,), float(I_B0), dtype=torch.float64, device=dev)
    for attempt in range(max_rounds):
        cur = x[:, pending]
        ins = {k: cur[i] for i, k in enumerate(PLUME_INPUTS)}
        ins['I_B0'] = ib0.expand(pending.numel())
        j = current_density(ins, sweep_radius=radius)['j_ion']
        bad = (j >= threshold).any(dim=-1)
        if not bool(bad.any()):
            return x, attempt + 1
        pending = pending[bad]
        redraw = sampling.Design(priors=pri, names=PLUME_INPUTS, seed=seed, stream=2 * (attempt + 1)).sample(n, device=dev)
        x[:, pending] = redraw[:, pending]
    raise RuntimeError(f'{pending.numel()} samples still exceed {threshold} A/m^2 after {max_rounds} rounds')
