"""pem_quantiles_f64_dev / drivers.column_percentiles (csrc/pem_quantile.hip): exact per-column percentiles over the sample axis
-- the p25 / p75 of the IQR masks (gen_data.py:125-174) and the 5 / 50 / 95 % bands of monte_carlo.py:363-658 -- held to
np.percentile BIT FOR BIT (order statistics are selected, not estimated, and the interpolation is numpy's own _lerp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(a, pcts):
    import torch
    from hallthrusterpem_amd.drivers import column_percentiles
    with np.errstate(invalid='ignore'):                       # (inf - inf inside numpy's interpolation of a column of infinities)
        want = np.percentile(a, pcts, axis=0)
    got = column_percentiles(torch.from_numpy(np.ascontiguousarray(a)).cuda(), pcts).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True), (np.abs(got - want).max(), a.shape, pcts)


@pytest.mark.parametrize('shape', [(1, 1), (2, 3), (3, 91), (777, 91), (100_003, 91), (50_000, 1), (4_000, 64), (4_000, 65), (3_000, 129),
                                   (2_000, 256), (1_500, 300), (20_000, 7, 13)])
def test_random_columns_equal_numpy(shape):
    rng = np.random.default_rng(sum(shape))
    a = rng.lognormal(0.0, 3.0, shape) * np.where(rng.random(shape) < 0.3, -1.0, 1.0)      # 30 decades, both signs
    _check(a, [25.0, 75.0])
    _check(a, [5.0, 50.0, 95.0])
    _check(a, 50.0)
    _check(a, [100.0, 0.0, 33.3, 99.999, 1e-3])             # unsorted, the ends, more than three per call


def test_ties_constants_and_special_values():
    rng = np.random.default_rng(5)
    n = 300_000
    a = rng.lognormal(0.0, 1.0, (n, 6))
    a[:, 0] = 7.25                                           # a constant column
    a[rng.random(n) < 0.4, 1] = 1e-20                       # 40 % ties at the minimum (the profile of invalid samples): lists longer than LDS
    a[:, 2] = rng.integers(0, 5, n)                         # five distinct values
    a[::7, 3] = np.inf
    a[1::7, 3] = -np.inf
    a[:, 4] = np.where(rng.random(n) < 0.5, 0.0, -0.0)      # signed zeros only
    a[12345, 5] = np.nan                                     # one NaN: the column's percentiles are NaN
    for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0], [0.0, 10.0, 40.0, 100.0]):
        _check(a, pcts)
    b = np.full((1000, 3), np.nan)                           # nothing but NaN
    b[:, 1] = 1.0
    _check(b, [25.0, 75.0])
    d = rng.standard_normal((4097, 2))                       # denormals and huge values in one column
    d[:, 0] *= 1e-310
    d[::2, 1] *= 1e300
    _check(d, [25.0, 50.0, 75.0])


def test_filter_outputs_on_the_device_equals_the_host_path_at_a_size_torch_quantile_refuses():
    """drivers.filter_outputs (gen_data.py:125-174) on CUDA tensors goes through the selection kernel: identical masks to the
    numpy branch, also where one column is longer than torch.quantile's 2^24 limit."""
    import torch
    from hallthrusterpem_amd.drivers import filter_outputs
    rng = np.random.default_rng(11)
    n = 200_000
    out = {'T_c': rng.normal(0.08, 0.01, n), 'j_ion': rng.lognormal(0.0, 1.0, (n, 91)), 'j_ion_coords': np.zeros((n, 91))}
    out['T_c'][::1000] = 5.0                                 # outliers
    out['j_ion'][5::5000] *= 1e4
    out['j_ion'][7, 3] = np.nan
    nan_h, outl_h = filter_outputs(out)
    dev = {k: torch.from_numpy(v).cuda() for k, v in out.items()}
    nan_d, outl_d = filter_outputs(dev)
    for k in nan_h:
        assert np.array_equal(nan_h[k], nan_d[k].cpu().numpy()) and np.array_equal(outl_h[k], outl_d[k].cpu().numpy())
    assert outl_h['T_c'].sum() >= n // 1000 and nan_h['j_ion'].sum() == 1
    big = rng.standard_normal(17_000_000)                    # > 2^24 samples of a scalar QoI
    from hallthrusterpem_amd.drivers import column_percentiles
    got = column_percentiles(torch.from_numpy(big).cuda(), [25.0, 75.0]).cpu().numpy()
    assert np.array_equal(got, np.percentile(big, [25.0, 75.0]))


def test_percentile_bands_of_a_forward_uq_campaign():
    from hallthrusterpem_amd import drivers
    out = drivers.forward_uq(300_000, seed=4, keep_profile=True)
    bands = drivers.percentile_bands(out)
    assert set(bands) == {'V_cc', 'div_angle', 'T_c', 'j_ion'} and bands['j_ion'].shape == (3, 91) and bands['T_c'].shape == (3,)
    for k, v in bands.items():
        assert np.array_equal(v.cpu().numpy(), np.percentile(out[k].cpu().numpy(), [5.0, 50.0, 95.0], axis=0), equal_nan=True)
    assert bool((bands['j_ion'][0] <= bands['j_ion'][1]).all() and (bands['j_ion'][1] <= bands['j_ion'][2]).all())


def test_range_histogram_and_key_minmax_equal_their_numpy_restatement():
    """pem_key_minmax_f64_dev / pem_range_hist_f64_dev (the local operations of the multi-rank percentiles) against
    percentiles.local_minmax_numpy / local_hist_numpy -- the restatement the gloo tests run the level logic on."""
    import torch
    from hallthrusterpem_amd import percentiles as P
    rng = np.random.default_rng(8)
    for n, m in ((5000, 1), (4000, 7), (3000, 91), (2000, 200), (0, 5)):
        a = rng.lognormal(0.0, 3.0, (n, m)) * rng.choice([-1.0, 1.0], (n, m))
        if n:
            a[::9, 0] = np.nan
            a[1::9, m - 1] = -np.inf
        cols = P.DeviceColumns(torch.from_numpy(a).cuda())
        kmin, kmax, nan = cols.minmax()
        wmin, wmax, wnan = P.local_minmax_numpy(a)
        assert np.array_equal(kmin, wmin) and np.array_equal(kmax, wmax) and np.array_equal(nan, wnan)
        if not n:
            continue
        keys = np.sort(P.key_of(a[~np.isnan(a)]))
        for nr, bins in ((1, 64), (2, 32), (4, 16), (6, 8)):
            lo = keys[rng.integers(0, keys.size, (m, nr))]
            hi = np.maximum(lo, keys[rng.integers(0, keys.size, (m, nr))])
            hi[0, 0] = lo[0, 0]                                  # a single-key range
            if nr > 1:
                hi[0, 1] = lo[0, 1] + np.uint64(bins - 1)        # fewer keys than bins: every key its own bin
            got = cols.hist(lo, hi, bins)
            assert np.array_equal(got, P.local_hist_numpy(a, lo, hi, bins)), (n, m, nr)


def test_sharded_percentiles_on_one_rank_equal_numpy():
    import torch
    from hallthrusterpem_amd.percentiles import column_percentiles_sharded
    rng = np.random.default_rng(9)
    a = rng.lognormal(0.0, 3.0, (200_000, 91)) * rng.choice([-1.0, 1.0], (200_000, 91))
    a[rng.random(200_000) < 0.2, 3] = 1e-20
    a[5, 7] = np.nan
    d = torch.from_numpy(a).cuda()
    for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0], 50.0):
        for on_device in (True, False):      # the levels decided on the device (pem_range_narrow_dev) and by the numpy restatement
            assert np.array_equal(column_percentiles_sharded(d, pcts, on_device=on_device), np.percentile(a, pcts, axis=0), equal_nan=True)
    s = torch.from_numpy(a[:, 0].copy()).cuda()
    assert column_percentiles_sharded(s, [10.0, 90.0]).shape == (2,) and np.array_equal(column_percentiles_sharded(s, [10.0, 90.0]), np.percentile(a[:, 0], [10.0, 90.0]))
    wide = torch.from_numpy(rng.standard_normal((3000, 300))).cuda()
    assert np.array_equal(column_percentiles_sharded(wide, [50.0]), np.percentile(wide.cpu().numpy(), [50.0], axis=0))


def _two_rank_worker(rank, world, port, out_dir):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)            # (both ranks share this box's one GPU: gloo collectives)
    try:
        from hallthrusterpem_amd import drivers
        from hallthrusterpem_amd.percentiles import column_percentiles_sharded
        n = 400_001
        lo, hi = (0, 150_000) if rank == 0 else (150_000, n)
        out = drivers.forward_uq(n, seed=6, keep_profile=True, rank=0, world=1)          # the whole campaign, for the expected values
        mine = {k: out[k][lo:hi].contiguous() for k in ('j_ion', 'T_c')}
        for k, pcts in (('j_ion', [5.0, 50.0, 95.0]), ('T_c', [25.0, 75.0])):
            got = column_percentiles_sharded(mine[k], pcts)
            assert np.array_equal(got, np.percentile(out[k].cpu().numpy(), pcts, axis=0), equal_nan=True), (rank, k)
        open(os.path.join(out_dir, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_two_ranks_sharing_the_gpu_get_the_percentiles_of_the_whole_campaign(tmp_path):
    """The multi-rank path end to end with the DEVICE histograms: two processes hold 150 000 and 250 001 samples of one
    forward-UQ campaign, all-reduce their counts level by level and both arrive at np.percentile of all 400 001 samples."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()
