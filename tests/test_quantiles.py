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


def _last_path():
    from hallthrusterpem_amd import _lib
    return _lib.load().pem_quantiles_last_path()


@pytest.fixture
def pilot_from(monkeypatch):
    """The pilot form of the selection (every 32nd row brackets the ranks, two passes over the data) is used from 2^19 rows on;
    the fixture lowers that threshold so that small cases take it too."""
    def set_min(n, stride=None):
        monkeypatch.setenv('PEM_QUANTILE_PILOT_MIN', str(n))
        if stride is not None:
            monkeypatch.setenv('PEM_QUANTILE_PILOT', str(stride))
    return set_min


@pytest.mark.parametrize('shape', [(129, 1), (4_000, 64), (4_000, 65), (40_000, 91), (3_000, 129), (2_000, 256), (1_500, 300), (20_000, 7, 13),
                                   (600_000, 91), (2_000_000, 3)])
def test_pilot_form_equals_numpy(shape, pilot_from):
    pilot_from(128)
    rng = np.random.default_rng(sum(shape) + 1)
    a = rng.lognormal(0.0, 3.0, shape) * np.where(rng.random(shape) < 0.3, -1.0, 1.0)
    for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0], 50.0, [100.0, 0.0, 33.3, 99.999, 1e-3]):
        _check(a, pcts)
        assert _last_path() == 1, (shape, pcts)             # exchangeable rows: the brackets hold


def test_pilot_form_ties_constants_and_special_values(pilot_from):
    pilot_from(128)
    rng = np.random.default_rng(6)
    n = 300_000
    a = rng.lognormal(0.0, 1.0, (n, 7))
    a[:, 0] = 7.25                                           # a constant column: a bracket of one key
    a[rng.random(n) < 0.4, 1] = 1e-20                       # 40 % ties at the minimum
    a[:, 2] = rng.integers(0, 5, n)                         # five distinct values
    a[::7, 3] = np.inf
    a[1::7, 3] = -np.inf
    a[:, 4] = np.where(rng.random(n) < 0.5, 0.0, -0.0)
    a[12345, 5] = np.nan                                     # a NaN in a row the pilot does not see
    a[64, 6] = np.nan                                        # ... and one in a row it does
    for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0], [0.0, 10.0, 40.0, 100.0]):
        _check(a, pcts)
        assert _last_path() == 1
    b = np.full((1000, 3), np.nan)
    b[:, 1] = 1.0
    _check(b, [25.0, 75.0])
    d = rng.standard_normal((4097, 2))
    d[:, 0] *= 1e-310
    d[::2, 1] *= 1e300
    _check(d, [25.0, 50.0, 75.0])


def test_pilot_form_with_many_hits_per_wave_instruction(pilot_from):
    """A tiny array under a forced pilot: 126 pilot rows leave brackets that are open at one end, their sub-bins hold thousands
    of the 4001 values, and most wave instructions of the copy pass hold several hits.  The case in which a version of that
    pass that parked its hits in LDS lost values (the library now checks every list's length against its count)."""
    pilot_from(128)
    rng = np.random.default_rng(3)
    a = np.stack([rng.permutation(4001).astype(np.float64) for _ in range(64)], axis=1)
    for _ in range(2):
        for pcts in ([25.0, 75.0], 50.0, [12.5, 37.5, 62.5]):
            _check(a, pcts)
            assert _last_path() == 1


def test_pilot_form_on_ordered_data_and_its_fallback(pilot_from):
    """Sorted and periodic data: a strided subsample represents a sorted column exactly, so the brackets hold; rows whose
    every 32nd member is an outlier defeat them -- the counts of pass A say so and the call repeats with the four passes.
    Either way the result is np.percentile's."""
    pilot_from(128)
    rng = np.random.default_rng(8)
    n = 200_000
    a = np.sort(rng.standard_normal((n, 5)), axis=0)
    a[:, 1] = a[::-1, 1]
    _check(a, [5.0, 50.0, 95.0])
    assert _last_path() == 1
    b = rng.standard_normal((n, 4))
    b[::32, 2] = 1e6 + rng.random(b[::32, 2].shape)          # the pilot sees nothing but these in column 2
    for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0]):
        _check(b, pcts)
        assert _last_path() == 2
    _check(b, 0.0)                                           # the minimum: its bracket is open below and holds it whatever the pilot saw
    assert _last_path() == 1
    pilot_from(128, stride=0)                                # switched off
    _check(b, [25.0, 75.0])
    assert _last_path() == 0
    pilot_from(128, stride=7)                                # another stride: column 2 is ordinary again
    _check(b, [25.0, 75.0])
    assert _last_path() == 1


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


@pytest.mark.parametrize('shape', [(1, 1), (1000, 1), (50_001, 3), (4_000, 64), (4_001, 65), (30_000, 91), (3_000, 129), (2_000, 256),
                                   (1_500, 300), (700, 512), (300, 600), (5_000, 7, 13)])
def test_filter_outputs_masks_equal_the_numpy_branch(shape):
    """The one-pass NaN / outside-count kernel (csrc/pem_masks.hip) behind drivers.filter_outputs on CUDA tensors against the
    numpy branch (= the reference's arithmetic, tests/test_reference_suite.py pins that one to the reference's own function):
    every row layout of the kernel (1..64 entries: several rows per wave instruction; 65..512: 2, 4, 8 chunks per row), more
    than 512 entries (the torch expressions), NaNs, infinities, a column whose bounds are NaN."""
    import torch
    from hallthrusterpem_amd.drivers import filter_outputs
    rng = np.random.default_rng(sum(shape) + 17)
    a = rng.lognormal(0.0, 1.0, shape)
    n = shape[0]
    a[rng.random(n) < 0.02] *= 1e3                           # whole samples far outside: outliers of the variable
    a[rng.random(shape) < 0.01] = 1e4                        # single entries outside
    if n > 10:
        a[3].flat[0] = np.nan
        a[n // 2].flat[-1] = np.nan
        a[7].flat[0] = np.inf
        a[8].flat[0] = -np.inf
    for f in (1.5, 0.0):
        nan_h, out_h = filter_outputs({'v': a, 'v_coords': a, 'errors': a}, iqr_factor=f)
        nan_d, out_d = filter_outputs({'v': torch.from_numpy(a).cuda(), 'v_coords': a, 'errors': a}, iqr_factor=f)
        assert set(nan_d) == {'v'} and nan_d['v'].dtype == torch.bool and out_d['v'].dtype == torch.bool
        assert np.array_equal(nan_h['v'], nan_d['v'].cpu().numpy()) and np.array_equal(out_h['v'], out_d['v'].cpu().numpy())
    if n > 10:
        assert nan_h['v'].sum() == 2                         # (the entries that hold a NaN have NaN bounds: nothing is outside them)


def test_row_masks_entry_point_counts_and_errors():
    import ctypes as C
    import torch
    from hallthrusterpem_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(23)
    for n, m, ld in ((777, 5, 8), (1000, 91, 96), (64, 200, 200)):
        a = rng.standard_normal((n, ld))
        a[::50, 0] = np.nan
        lo, hi = rng.normal(-1.0, 0.1, m), rng.normal(1.0, 0.1, m)
        lo[m // 2] = np.nan                                  # a comparison with NaN is false
        d, dl, dh = (torch.from_numpy(v).cuda() for v in (a, lo, hi))
        nan = torch.full((n,), 7, dtype=torch.uint8, device='cuda')
        cnt = torch.full((n,), -1, dtype=torch.int32, device='cuda')
        _lib.check(lib.pem_row_masks_f64_dev(n, m, C.c_void_p(d.data_ptr()), ld, C.c_void_p(dl.data_ptr()), C.c_void_p(dh.data_ptr()),
                                             C.c_void_p(nan.data_ptr()), C.c_void_p(cnt.data_ptr()), None))
        torch.cuda.synchronize()
        with np.errstate(invalid='ignore'):
            want = ((a[:, :m] < lo) | (a[:, :m] > hi)).sum(axis=1)
        assert np.array_equal(cnt.cpu().numpy(), want) and np.array_equal(nan.cpu().numpy(), np.isnan(a[:, :m]).any(axis=1))
    assert lib.pem_row_masks_f64_dev(10, 0, None, 0, None, None, None, None, None) == 1            # PEM_ERR_INVALID_ARG
    assert lib.pem_row_masks_f64_dev(10, 513, C.c_void_p(d.data_ptr()), 513, None, None, None, None, None) == 1
    assert lib.pem_row_masks_f64_dev(10, 4, C.c_void_p(d.data_ptr()), 3, C.c_void_p(dl.data_ptr()), C.c_void_p(dh.data_ptr()),
                                     C.c_void_p(nan.data_ptr()), C.c_void_p(cnt.data_ptr()), None) == 1
    assert lib.pem_row_masks_f64_dev(0, 4, None, 4, None, None, None, None, None) == 0               # nothing to do


def test_five_and_six_quantiles_share_one_selection(pilot_from):
    """Round 4: up to six quantiles per selection (PEM_QUANTILE_MAX_Q) -- a campaign's p25 / p75 and its 5 / 50 / 95 % bands go
    through ONE pilot, one counting pass and one copy pass -- in the four-pass form and in the pilot form, ties and NaN included,
    and three for more than 128 columns (PEM_QUANTILE_MAX_Q_WIDE; the driver splits)."""
    import ctypes as C
    import torch
    from hallthrusterpem_amd import _lib
    rng = np.random.default_rng(41)
    five, six = [25.0, 75.0, 5.0, 50.0, 95.0], [0.0, 5.0, 25.0, 50.0, 99.9, 100.0]
    for shape in ((40_000, 91), (9_000, 3), (5_000, 128), (3_000, 130), (2_000, 256), (6_000, 7, 13)):
        a = rng.lognormal(0.0, 2.0, shape)
        a[rng.random(shape[0]) < 0.3] = 1e-20                # 30 % of the rows tied at the minimum: the 5 % and 25 % brackets coincide
        _check(a, five)
        _check(a, six)
    pilot_from(256)                                          # the pilot form from 2^8 values on
    for shape in ((300_000, 91), (200_000, 3), (100_000, 128)):
        a = rng.lognormal(0.0, 2.0, shape) * np.where(rng.random(shape) < 0.2, -1.0, 1.0)
        a[rng.random(shape[0]) < 0.1] = 1e-20
        _check(a, five)
        assert _last_path() == 1
        _check(a, six)
        assert _last_path() == 1
    a[777, 1] = np.nan
    _check(a, five)
    # the C entry point refuses more than it is instantiated for
    lib = _lib.load()
    d = torch.zeros((100, 200), dtype=torch.float64, device='cuda')
    out = torch.zeros((6, 200), dtype=torch.float64, device='cuda')
    r = (C.c_uint64 * 6)(*[10] * 6)
    g = (C.c_double * 6)(*[0.0] * 6)
    assert lib.pem_quantiles_f64_dev(100, 200, C.c_void_p(d.data_ptr()), 200, 4, r, r, g, C.c_void_p(out.data_ptr()), None) == 1
    assert lib.pem_quantiles_f64_dev(100, 128, C.c_void_p(d.data_ptr()), 200, 7, r, r, g, C.c_void_p(out.data_ptr()), None) == 1
    assert lib.pem_quantiles_f64_dev(100, 128, C.c_void_p(d.data_ptr()), 200, 6, r, r, g, C.c_void_p(out.data_ptr()), None) == 0


def test_columns_read_in_place_from_a_transposed_tensor(pilot_from):
    """`pem_quantiles_strided_f64_dev`: the [3][n] reduced-QoI tensor of a batch seen as (n, 3) -- V_cc, div_angle, T_c in ONE
    selection, without a copy -- equals numpy on the transposed copy; four-pass and pilot form, 1 to 70 columns."""
    import torch
    from hallthrusterpem_amd.drivers import column_percentiles
    rng = np.random.default_rng(43)
    for pilot in (False, True):
        if pilot:
            pilot_from(256)
        for k, n in ((3, 250_001), (1, 5_000), (2, 70_000), (64, 9_000), (70, 8_000)):
            rows = rng.lognormal(0.0, 1.5, (k, n + 5))      # rows longer than n: a column stride larger than the row count
            t = torch.from_numpy(rows).cuda()
            view = t[:, :n].T                                # (n, k), strides (1, n + 5)
            assert k == 1 or (view.stride(0) == 1 and not view.is_contiguous())
            for pcts in ([25.0, 75.0, 5.0, 50.0, 95.0], 50.0):
                got = column_percentiles(view, pcts).cpu().numpy()
                assert np.array_equal(got, np.percentile(rows[:, :n].T, pcts, axis=0)), (k, n, pilot)


def test_campaign_statistics_equal_the_two_separate_calls():
    """drivers.campaign_statistics: masks and bands of a campaign from one selection of five quantiles per variable, the three
    scalar QoIs stacked in one call -- the same masks and the same bands as filter_outputs + percentile_bands, bit for bit."""
    import torch
    from hallthrusterpem_amd import drivers
    out = drivers.forward_uq(400_000, seed=14, keep_profile=True)
    out['T_c'][::997] = 9.0                                  # outliers of a scalar
    out['j_ion'][5::4001] *= 1e3                             # whole profiles far outside
    keep = {k: out[k] for k in ('V_cc', 'div_angle', 'T_c', 'j_ion')}
    assert drivers._stacked_rows([out[k] for k in ('V_cc', 'div_angle', 'T_c')]) is not None
    nan_a, outl_a = drivers.filter_outputs(keep)
    bands_a = drivers.percentile_bands(out)
    nan_b, outl_b, bands_b = drivers.campaign_statistics(keep)
    assert set(nan_b) == set(nan_a) == set(keep) and set(bands_b) == set(bands_a)
    for k in keep:
        assert torch.equal(nan_a[k], nan_b[k]) and torch.equal(outl_a[k], outl_b[k]), k
        assert torch.equal(bands_a[k], bands_b[k]), k
        assert np.array_equal(bands_b[k].cpu().numpy(), np.percentile(out[k].cpu().numpy(), [5.0, 50.0, 95.0], axis=0))
    assert int(outl_b['T_c'].sum()) >= 400_000 // 997 and int(outl_b['j_ion'].sum()) >= 90
    # tensors that are not rows of one allocation go one by one, with the same result
    sep = {k: v.clone() for k, v in keep.items()}
    assert drivers._stacked_rows([sep[k] for k in ('V_cc', 'div_angle', 'T_c')]) is None
    nan_c, outl_c, bands_c = drivers.campaign_statistics(sep, names=('T_c', 'j_ion', 'V_cc'))
    for k in ('T_c', 'j_ion', 'V_cc'):
        assert torch.equal(nan_c[k], nan_a[k]) and torch.equal(outl_c[k], outl_a[k]) and torch.equal(bands_c[k], bands_a[k])


@pytest.mark.parametrize('n,keep', [(300_000, True), (300_000, False), (70_001, True), (4_096, False)])
def test_fused_campaign_statistics_equal_numpy(n, keep):
    """drivers.forward_uq_statistics (pem_coupled_mc_stats_f64_dev): the percentiles of the profile counted inside the evaluation
    kernel -- brackets from the first 3 % of the samples, one evaluate-and-count launch, selection from the records -- against
    np.percentile of the profile the plain campaign writes, bit for bit; the other outputs, masks and bands against the separate
    calls; with and without a stored profile; a ragged last tile."""
    import torch
    from hallthrusterpem_amd import drivers
    ref = drivers.forward_uq(n, seed=21, keep_profile=True, keep_inputs=False)
    got = drivers.forward_uq_statistics(n, seed=21, keep_profile=keep)
    assert got['fused'] is (n >= 50_000)                     # (4096 samples: the brackets of 128 pilot rows overlap -- declined, still right)
    for k in ('V_cc', 'div_angle', 'T_c', 'I_B0', 'T', 'invalid'):
        if got['fused'] or keep:
            assert torch.equal(got[k], ref[k]), k            # the counting launch integrates the profile as the profile mode does
        else:                                                # (declined without a profile: the reduced-QoI launch, equal within its tolerance)
            assert torch.allclose(got[k].double(), ref[k].double(), rtol=1e-10, atol=0.0), k
    assert ('j_ion' in got) == keep
    if keep:
        assert torch.equal(got['j_ion'], ref['j_ion'])
    want = np.percentile(ref['j_ion'].cpu().numpy(), [5.0, 50.0, 95.0], axis=0)
    assert np.array_equal(got['bands']['j_ion'].cpu().numpy(), want)
    own = {k: got[k] for k in ('V_cc', 'div_angle', 'T_c')}
    bands = drivers.percentile_bands(own)
    for k in own:
        assert torch.equal(got['bands'][k], bands[k]), k
    nan_a, outl_a = drivers.filter_outputs(dict(own, j_ion=ref['j_ion']))
    for k in got['nan_idx']:
        assert torch.equal(got['nan_idx'][k], nan_a[k]) and torch.equal(got['outlier_idx'][k], outl_a[k]), k
    assert set(got['nan_idx']) == {'V_cc', 'div_angle', 'T_c', 'j_ion'} and got['premasked'] is got['fused']
    if n == 70_001:                                          # the sampled inputs, when asked for, are the plain campaign's
        refx = drivers.forward_uq(n, seed=21, keep_profile=False, keep_inputs=True)
        gotx = drivers.forward_uq_statistics(n, seed=21, keep_profile=keep, keep_inputs=True)
        assert gotx['fused'] and torch.equal(gotx['x'], refx['x']) and torch.equal(gotx['bands']['j_ion'], got['bands']['j_ion'])


def test_fused_campaign_statistics_decline_and_fall_back(monkeypatch):
    """What makes the on-chip selection decline leaves the results right: priors under which many samples are invalid (their
    profile is the constant 1e-20: brackets of one key), record regions that are too small, other percentile sets."""
    import torch
    from hallthrusterpem_amd import drivers, sampling
    n = 200_000
    pri = dict(sampling.PEM_V0_PRIORS)
    pri['c3'] = sampling.Prior(sampling.UNIFORM, -0.6, 1.1, 'test')          # alpha1 <= 0 for a third of the samples (plume.py:105)
    got = drivers.forward_uq_statistics(n, seed=5, keep_profile=True, priors=pri, percentiles=(5.0, 50.0, 95.0))
    assert got['fused'] is False and float(got['invalid'].float().mean()) > 0.2
    j = got['j_ion'].cpu().numpy()
    assert np.array_equal(got['bands']['j_ion'].cpu().numpy(), np.percentile(j, [5.0, 50.0, 95.0], axis=0))
    nof = drivers.forward_uq_statistics(n, seed=5, keep_profile=False, priors=pri)
    assert nof['fused'] is False and torch.equal(nof['bands']['j_ion'], got['bands']['j_ion'])
    monkeypatch.setenv('PEM_QUANTILE_RECORD_CAP', '8')                          # every wave overflows its records
    small = drivers.forward_uq_statistics(n, seed=6, keep_profile=True)
    monkeypatch.delenv('PEM_QUANTILE_RECORD_CAP')
    assert small['fused'] is False
    ext = drivers.forward_uq_statistics(n, seed=6, keep_profile=True, percentiles=(0.0, 0.2, 99.8, 100.0))
    assert ext['fused'] is False                                                # (the brackets of 0 % and 0.2 % overlap)
    assert np.array_equal(ext['bands']['j_ion'].cpu().numpy(), np.percentile(ext['j_ion'].cpu().numpy(), [0.0, 0.2, 99.8, 100.0], axis=0))
    ends = drivers.forward_uq_statistics(n, seed=6, keep_profile=True, percentiles=(0.0, 1.0, 99.0, 100.0))
    assert ends['fused'] is True                                                # (minimum and maximum: brackets open at one end, counted on chip)
    assert np.array_equal(ends['bands']['j_ion'].cpu().numpy(), np.percentile(ends['j_ion'].cpu().numpy(), [0.0, 1.0, 99.0, 100.0], axis=0))
    ok = drivers.forward_uq_statistics(n, seed=6, keep_profile=True, percentiles=(2.0, 40.0, 60.0, 98.0))
    assert ok['fused'] is True and ok['premasked'] is False and torch.equal(ok['j_ion'], small['j_ion'])      # six brackets: no premask
    assert np.array_equal(ok['bands']['j_ion'].cpu().numpy(), np.percentile(ok['j_ion'].cpu().numpy(), [2.0, 40.0, 60.0, 98.0], axis=0))
    assert torch.equal(ok['outlier_idx']['j_ion'], small['outlier_idx']['j_ion'])
    assert np.array_equal(small['bands']['j_ion'].cpu().numpy(), np.percentile(ok['j_ion'].cpu().numpy(), [5.0, 50.0, 95.0], axis=0))


def test_percentile_bands_of_a_forward_uq_campaign():
    from hallthrusterpem_amd import drivers
    out = drivers.forward_uq(300_000, seed=4, keep_profile=True)
    bands = drivers.percentile_bands(out)
    assert set(bands) == {'V_cc', 'div_angle', 'T_c', 'j_ion'} and bands['j_ion'].shape == (3, 91) and bands['T_c'].shape == (3,)
    for k, v in bands.items():
        assert np.array_equal(v.cpu().numpy(), np.percentile(out[k].cpu().numpy(), [5.0, 50.0, 95.0], axis=0), equal_nan=True)
    assert bool((bands['j_ion'][0] <= bands['j_ion'][1]).all() and (bands['j_ion'][1] <= bands['j_ion'][2]).all())


def test_range_histogram_and_key_minmax_equal_their_numpy_restatement():
    """pem_qsel_minmax_f64_dev / pem_range_hist_f64_dev (the local operations of the level loop the sharded percentiles fall back
    to) against percentiles.local_minmax_numpy / local_hist_numpy -- the restatement the gloo tests run the level logic on."""
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


def test_selection_stages_equal_their_numpy_restatement():
    """Every stage of the sharded selection (pem_qsel_hist1 / decide1 / hist2 / decide2 / compact / select) against the numpy
    restatement the gloo rehearsal runs (percentiles.NumpyColumns, decide1, decide2, select_lists), on data with ties, infinities,
    a constant column, a NaN, for 2 / 4 / 6 targets and 1 ... 200 columns -- and with the lists of two "ranks" put side by side."""
    import ctypes as C
    import torch
    from hallthrusterpem_amd import _lib, percentiles as P
    lib = _lib.load()
    rng = np.random.default_rng(21)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())                                   # noqa: E731
    for n, m, pcts in ((60_000, 91, [5.0, 50.0, 95.0]), (20_000, 7, [25.0, 75.0]), (9000, 1, [50.0]), (5000, 200, [1.0, 99.0]), (300, 64, [0.0, 100.0, 50.0])):
        a = rng.lognormal(0.0, 3.0, (n, m)) * rng.choice([-1.0, 1.0], (n, m))
        a[rng.random(n) < 0.25, 0] = 1e-20
        if m > 3:
            a[:, 2] = 4.5
            a[::11, 3] = np.inf
            a[5, 1] = np.nan
        ref, dev = P.NumpyColumns(a), P.DeviceColumns(torch.from_numpy(a).cuda())
        kmin, kmax, nan = ref.minmax()
        dmin, dmax, dnan = dev.minmax()
        assert np.array_equal(kmin, dmin) and np.array_equal(kmax, dmax) and np.array_equal(nan, dnan)
        rp, rn, _ = P.linear_ranks(n, pcts)
        ranks = np.stack([rp, rn], axis=1).reshape(-1)
        nt = ranks.size
        bins1, bins2 = P.qsel_bins(m, nt)
        b1, b2 = C.c_int(0), C.c_int(0)
        _lib.check(lib.pem_qsel_bins(m, nt, C.byref(b1), C.byref(b2)))
        assert (b1.value, b2.value) == (bins1, bins2)
        i32, i64 = dict(dtype=torch.int32, device='cuda'), dict(dtype=torch.int64, device='cuda')
        kmn, kmx = (torch.from_numpy(k.view(np.int64).copy()).cuda() for k in (kmin, kmax))
        # stage 1
        h1 = torch.empty((m, bins1), **i32)
        _lib.check(lib.pem_qsel_hist1_f64_dev(n, m, p(dev.flat), m, p(kmn), p(kmx), bins1, p(h1), st))
        want_h1 = ref.hist1(kmin, kmax, bins1)
        assert np.array_equal(h1.cpu().numpy(), want_h1)
        resid = torch.from_numpy(np.broadcast_to(ranks, (m, nt)).astype(np.int64).copy()).cuda()
        bin1, done, ans = torch.empty((m, nt), **i32), torch.empty((m, nt), **i32), torch.zeros((m, nt), **i64)
        _lib.check(lib.pem_qsel_decide1_dev(m, nt, p(kmn), p(kmx), p(h1), bins1, p(resid), p(bin1), p(done), p(ans), st))
        w_bin1, w_done, w_ans, w_resid = P.decide1(kmin, kmax, want_h1, np.broadcast_to(ranks, (m, nt)).astype(np.int64))
        live = ~w_done
        assert np.array_equal(done.cpu().numpy().astype(bool), w_done) and np.array_equal(bin1.cpu().numpy(), w_bin1)
        assert np.array_equal(resid.cpu().numpy()[live], w_resid[live]) and np.array_equal(ans.cpu().numpy().view(np.uint64)[w_done], w_ans[w_done])
        # stage 2
        h2 = torch.empty((m, nt, bins2), **i32)
        _lib.check(lib.pem_qsel_hist2_f64_dev(n, m, p(dev.flat), m, p(kmn), p(kmx), nt, p(bin1), bins1, bins2, p(h2), st))
        want_h2 = ref.hist2(kmin, kmax, w_bin1, bins1, bins2)
        assert np.array_equal(h2.cpu().numpy(), want_h2)
        bin2, cnt, cntl = torch.empty((m, nt), **i32), torch.empty((m, nt), **i64), torch.empty((m, nt), **i32)
        _lib.check(lib.pem_qsel_decide2_dev(m, nt, p(h2), p(h2), bins2, p(bin1), p(done), p(resid), p(bin2), p(cnt), p(cntl), st))
        w_bin2, w_cnt, w_cntl, w_resid2 = P.decide2(want_h2, want_h2, w_bin1, w_done, w_resid)
        assert np.array_equal(bin2.cpu().numpy(), w_bin2) and np.array_equal(cnt.cpu().numpy(), w_cnt)
        assert np.array_equal(cntl.cpu().numpy(), w_cntl) and np.array_equal(resid.cpu().numpy()[live], w_resid2[live])
        # stage 3: the padded lists (order inside a list is the kernel's own: compared as sets), then the selection
        L = P._list_len(int(w_cntl.max(initial=0)))
        cand, cur = torch.empty((m * nt, L), **i64), torch.empty(m * nt, **i32)
        _lib.check(lib.pem_qsel_compact_f64_dev(n, m, p(dev.flat), m, p(kmn), p(kmx), nt, p(bin1), p(bin2), p(done), bins1, bins2, L, p(cand), p(cur), st))
        w_cand, w_cur = ref.compact(kmin, kmax, w_bin1, w_bin2, w_done, bins1, bins2, L)
        assert np.array_equal(cur.cpu().numpy().reshape(m, nt), w_cur)
        assert np.array_equal(np.sort(cand.cpu().numpy().view(np.uint64).reshape(m, nt, L), axis=2), np.sort(w_cand, axis=2))
        out = torch.zeros((m, nt), **i64)
        for world in (1, 2):                   # two "ranks": the same lists twice -> the union holds every key twice
            gathered = torch.stack([cand] * world).contiguous()
            r2 = resid * world if world == 2 else resid
            _lib.check(lib.pem_qsel_select_dev(m, nt, world, L, p(gathered), p(bin1), p(bin2), p(done), p(r2), p(out), st))
            want = P.select_lists(np.stack([w_cand] * world), w_bin1, w_bin2, w_done, w_resid2 * world, w_ans)
            ok = live & (nan == 0)[:, None]        # (a rank past a NaN column's last value has an empty list: its result is NaN anyway)
            assert np.array_equal(out.cpu().numpy().view(np.uint64)[ok], want[ok]), (n, m, world)
        col_sorted = np.sort(a, axis=0)
        keep = (nan == 0)
        assert np.array_equal(P.value_of(np.where(w_done, w_ans, want))[keep], col_sorted[ranks].T[keep])


def test_sharded_percentiles_on_one_rank_equal_numpy():
    import torch
    from hallthrusterpem_amd.percentiles import column_percentiles_sharded
    rng = np.random.default_rng(9)
    a = rng.lognormal(0.0, 3.0, (200_000, 91)) * rng.choice([-1.0, 1.0], (200_000, 91))
    a[rng.random(200_000) < 0.2, 3] = 1e-20
    a[5, 7] = np.nan
    d = torch.from_numpy(a).cuda()
    for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0], 50.0):
        for method in ('select', 'levels'):  # the four-pass selection, and the level loop it falls back to
            assert np.array_equal(column_percentiles_sharded(d, pcts, method=method), np.percentile(a, pcts, axis=0), equal_nan=True)
    # 20 % ties at one value: with a short cap on the candidate lists the pass is handed to the level loop, same result
    from hallthrusterpem_amd import percentiles as P
    keep, P.LIST_CAP = P.LIST_CAP, 256
    try:
        assert np.array_equal(column_percentiles_sharded(d, [5.0, 50.0, 95.0]), np.percentile(a, [5.0, 50.0, 95.0], axis=0), equal_nan=True)
    finally:
        P.LIST_CAP = keep
    # ... a percentile INSIDE the tie block (40 000 equal keys in one sub-bin, more than any list may hold): the level loop again
    assert np.array_equal(column_percentiles_sharded(d[:, 3].contiguous(), [10.0]), np.percentile(a[:, 3], [10.0]))
    s = torch.from_numpy(a[:, 0].copy()).cuda()
    assert column_percentiles_sharded(s, [10.0, 90.0]).shape == (2,) and np.array_equal(column_percentiles_sharded(s, [10.0, 90.0]), np.percentile(a[:, 0], [10.0, 90.0]))
    wide = torch.from_numpy(rng.standard_normal((3000, 300))).cuda()
    for method in ('select', 'levels'):
        assert np.array_equal(column_percentiles_sharded(wide, [50.0], method=method), np.percentile(wide.cpu().numpy(), [50.0], axis=0))
    # more columns than 3072: round 2 sized the level loop's bins from the whole width and never converged
    wider = torch.from_numpy(rng.standard_normal((50, 3100))).cuda()
    for method in ('select', 'levels'):
        assert np.array_equal(column_percentiles_sharded(wider, [5.0, 50.0, 95.0], method=method), np.percentile(wider.cpu().numpy(), [5.0, 50.0, 95.0], axis=0))


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
        # the driver-level form: the bands of the whole campaign from this rank's shard (sharding is opt-in: ADVICE r3)
        bands = drivers.percentile_bands(mine, names=('T_c', 'j_ion'), sharded=True)
        for k in ('T_c', 'j_ion'):
            assert bands[k].is_cuda and np.array_equal(bands[k].cpu().numpy(), np.percentile(out[k].cpu().numpy(), [5.0, 50.0, 95.0], axis=0), equal_nan=True)
        # ... and the NaN / IQR masks of the whole campaign's data set, for this rank's samples
        whole = {k: out[k].clone() for k in ('j_ion', 'T_c')}
        whole['T_c'][::777] = 9.0
        whole['j_ion'][123_456, 5] = float('nan')
        nan_all, outl_all = drivers.filter_outputs(whole, sharded=False)
        nan_r, outl_r = drivers.filter_outputs({k: v[lo:hi].contiguous() for k, v in whole.items()}, sharded=True)
        for k in nan_all:
            assert torch.equal(nan_r[k], nan_all[k][lo:hi]) and torch.equal(outl_r[k], outl_all[k][lo:hi]), (rank, k)
        assert int(outl_all['T_c'].sum()) >= n // 777 and int(nan_all['j_ion'].sum()) == 1
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


@pytest.mark.gpu
@pytest.mark.parametrize('pad', [0, 1], ids=['rows_n_apart', 'rows_on_4_byte_boundaries'])
def test_campaign_masks_entry_point_equals_numpy(pad):
    """pem_campaign_masks_f64_dev: the masks of scalar outputs (NaN; outside p25 - f iqr .. p75 + f iqr with numpy's roundings) and the
    verdict of the premask counts (outlier for certain, settled, or listed as open), against numpy on crafted arrays."""
    import ctypes as C
    import torch
    from hallthrusterpem_amd import _lib
    rng = np.random.default_rng(11)
    n, nv, f, thresh = 100_003, 3, 1.5, 68
    x = rng.standard_normal((nv, n)) * np.array([[1.0], [1e-3], [250.0]]) + np.array([[30.0], [0.2], [-4.0]])
    x[1, ::977] = np.nan
    x[2, 5] = np.inf
    q = np.stack([np.nanpercentile(x, p, axis=1) for p in (25.0, 75.0, 5.0, 50.0, 95.0)])          # [5][3]
    certain = rng.integers(0, 92, n).astype(np.uint8)
    uncertain = np.minimum(91 - certain, rng.integers(0, 4, n)).astype(np.uint8)
    xd, qd = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    cd, ud = torch.from_numpy(certain).cuda(), torch.from_numpy(uncertain).cuda()
    pitch = n + pad                                           # (100 004: the pass writes 4-byte words; 100 003: bytes)
    nan_o = torch.full((nv + 1, pitch), 7, dtype=torch.uint8, device='cuda')[:, :n]
    out_o = torch.full((nv + 1, pitch), 7, dtype=torch.uint8, device='cuda')[:, :n]
    cap = 64
    rows = torch.full((cap,), -1, dtype=torch.int64, device='cuda')
    count = torch.zeros(1, dtype=torch.int32, device='cuda')
    p = lambda t: C.c_void_p(t.data_ptr())                                                   # noqa: E731
    vars_ = (C.c_void_p * nv)(*[xd[i].data_ptr() for i in range(nv)])
    lib = _lib.load()
    _lib.check(lib.pem_campaign_masks_f64_dev(n, nv, vars_, p(qd), qd.stride(0), 0, 1, f, p(nan_o), p(out_o), nan_o.stride(0), p(cd), p(ud), thresh, p(rows), p(count),
                                              cap, None))
    torch.cuda.synchronize()
    iqr = q[1] - q[0]
    lo, hi = q[0] - f * iqr, q[1] + f * iqr
    with np.errstate(invalid='ignore'):
        want_out = (x < lo[:, None]) | (x > hi[:, None])
    assert np.array_equal(nan_o[:nv].cpu().numpy().astype(bool), np.isnan(x))
    assert np.array_equal(out_o[:nv].cpu().numpy().astype(bool), want_out)
    assert not nan_o[nv].any() and np.array_equal(out_o[nv].cpu().numpy().astype(bool), certain > thresh)
    open_want = np.nonzero((certain <= thresh) & (certain.astype(int) + uncertain > thresh))[0]
    n_open = int(count.item())
    assert n_open == open_want.size > cap                    # more than the list holds: counted all the same, the first `cap` listed
    listed = rows.cpu().numpy()
    assert np.all(np.isin(listed, open_want)) and np.unique(listed).size == cap
    # a list that fits; without the premask counts the profile's row is left alone
    rows2 = torch.full((open_want.size + 8,), -1, dtype=torch.int64, device='cuda')
    count.zero_()
    _lib.check(lib.pem_campaign_masks_f64_dev(n, nv, vars_, p(qd), qd.stride(0), 0, 1, f, p(nan_o), p(out_o), nan_o.stride(0), p(cd), p(ud), thresh, p(rows2), p(count),
                                              rows2.numel(), None))
    assert np.array_equal(np.sort(rows2[:int(count.item())].cpu().numpy()), open_want)
    nan_o.fill_(7)
    _lib.check(lib.pem_campaign_masks_f64_dev(n, nv, vars_, p(qd), qd.stride(0), 0, 1, f, p(nan_o), p(out_o), nan_o.stride(0), None, None, thresh, None, None, 0, None))
    torch.cuda.synchronize()
    assert bool((nan_o[nv] == 7).all()) and np.array_equal(nan_o[:nv].cpu().numpy().astype(bool), np.isnan(x))
    assert lib.pem_campaign_masks_f64_dev(n, 9, vars_, p(qd), qd.stride(0), 0, 1, f, p(nan_o), p(out_o), nan_o.stride(0), None, None, thresh, None, None, 0, None) == 1
    assert lib.pem_campaign_masks_f64_dev(n, nv, vars_, p(qd), 2, 0, 1, f, p(nan_o), p(out_o), nan_o.stride(0), None, None, thresh, None, None, 0, None) == 1
    assert lib.pem_campaign_masks_f64_dev(n, nv, vars_, p(qd), qd.stride(0), 0, 1, f, p(nan_o), p(out_o), nan_o.stride(0), p(cd), None, thresh, None, None, 0, None) == 1


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_host_threads_run_campaigns_at_once():
    """The fused statistics keep process-wide buffers (two sets, each behind a mutex) and start a worker thread of their own per
    call: two campaigns issued from two host threads at the same time serialise on those and both come out right."""
    import threading
    import torch
    from hallthrusterpem_amd import drivers
    n = 150_000
    want = {}
    for seed in (31, 32):
        ref = drivers.forward_uq(n, seed=seed, keep_profile=True, keep_inputs=False)
        want[seed] = (np.percentile(ref['j_ion'].cpu().numpy(), [5.0, 50.0, 95.0], axis=0),
                      np.percentile(ref['T_c'].cpu().numpy(), [5.0, 50.0, 95.0], axis=0))
    got, errors = {}, []

    def run(seed, keep):
        try:
            for _ in range(6):
                r = drivers.forward_uq_statistics(n, seed=seed, keep_profile=keep)
                torch.cuda.synchronize()
                got[seed] = (r['fused'], r['bands']['j_ion'].cpu().numpy(), r['bands']['T_c'].cpu().numpy())
        except Exception as exc:                                       # noqa: BLE001 (reported by the main thread)
            errors.append(exc)
    threads = [threading.Thread(target=run, args=(31, True)), threading.Thread(target=run, args=(32, False))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in threads), 'a campaign did not come back'
    assert not errors, errors
    for seed in (31, 32):
        assert got[seed][0] is True
        assert np.array_equal(got[seed][1], want[seed][0]) and np.array_equal(got[seed][2], want[seed][1]), seed
