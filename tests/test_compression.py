"""SVD compression of field QoIs on the fp64 MFMA path.  amisc's Compression is
third-party (parity unpinned): the kernels are checked against numpy matmul of the formulas stated in
hallthrusterpem_amd/compression.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _field(n, dof, seed):
    rng = np.random.default_rng(seed)
    x = np.linspace(0, 1, dof)
    modes = np.stack([np.exp(-((x - c) / w) ** 2) for c, w in ((0.1, 0.2), (0.5, 0.3), (0.8, 0.1), (0.3, 0.6))])
    coef = rng.lognormal(0, 0.5, (n, 4)) * np.array([5.0, 1.0, 0.3, 0.05])
    return coef @ modes + 0.02


@pytest.mark.parametrize('n', [1, 15, 16, 17, 1000, 65_537])
@pytest.mark.parametrize('dof,rank,norm,scale', [(91, 5, 'log10', 1.0), (91, 16, 'none', 1.0), (202, 7, 'linear', 1e-3),
                                                 (7, 3, 'none', 1.0)])
def test_compress_and_reconstruct_match_numpy(n, dof, rank, norm, scale):
    import torch
    from hallthrusterpem_amd.compression import SVDCompression
    y = _field(n, dof, seed=n + dof)
    rng = np.random.default_rng(rank)
    basis = np.linalg.qr(rng.standard_normal((dof, rank)))[0]
    c = SVDCompression(norm=norm, scale=scale, rank=rank)
    c.basis = torch.from_numpy(np.ascontiguousarray(basis)).cuda()
    ynorm = np.log10(y) if norm == 'log10' else y * scale
    want_z = ynorm @ basis
    got_z = c.compress(torch.from_numpy(y).cuda()).cpu().numpy()
    assert got_z.shape == (n, rank)
    assert np.max(np.abs(got_z - want_z)) <= 2e-13 * max(1.0, np.abs(ynorm).max() * np.sqrt(dof))
    z = rng.standard_normal((n, rank)) * 0.3
    want_y = z @ basis.T
    want_y = 10.0 ** want_y if norm == 'log10' else want_y / scale
    got_y = c.reconstruct(torch.from_numpy(z).cuda()).cpu().numpy()
    assert got_y.shape == (n, dof)
    assert np.max(np.abs(got_y - want_y) / np.abs(want_y).max()) <= 1e-13


def test_fit_rank_from_reconstruction_tol_on_plume_profiles():
    """yml:273-280: j_ion, norm log10, reconstruction_tol 0.01 -- on real profiles from the coupled kernel."""
    import torch
    from hallthrusterpem_amd import drivers
    from hallthrusterpem_amd.compression import SVDCompression
    res = drivers.forward_uq(20_000, seed=4, keep_profile=True)
    j = res['j_ion'][~res['invalid']]
    c = SVDCompression(norm='log10', reconstruction_tol=0.01).fit(j)
    assert 1 <= c.rank <= 16 and c.relative_error <= 0.01
    a = torch.log10(j)
    back = torch.log10(c.reconstruct(c.compress(j)))
    rel = float(torch.linalg.norm(back - a) / torch.linalg.norm(a))
    assert rel <= 0.01 and rel == pytest.approx(c.relative_error, rel=1e-6)       # Eckart-Young: exactly the tail energy
    worse = SVDCompression(norm='log10', rank=max(1, c.rank - 1)).fit(j)
    assert worse.relative_error > c.relative_error
    # the basis is orthonormal and compress(reconstruct(z)) is the identity on latents
    eye = c.basis.T @ c.basis
    assert float((eye - torch.eye(c.rank, device=eye.device, dtype=eye.dtype)).abs().max()) < 1e-12
    z = c.compress(j[:1000])
    assert float((c.compress(c.reconstruct(z)) - z).abs().max()) < 1e-9


def test_fused_log10_and_exp10_norms_elementwise():
    """csrc/pem_math.h through the C ABI: with unit vectors as the basis, latent[:, r] = log10(field[:, r]) and
    field[:, r] = 10**latent[:, r] -- every element against numpy, including the arguments that take the library path."""
    import torch
    from hallthrusterpem_amd.compression import SVDCompression
    rng = np.random.default_rng(12)
    dof, rank, n = 91, 16, 40_000
    basis = np.zeros((dof, rank))
    basis[np.arange(rank), np.arange(rank)] = 1.0
    c = SVDCompression(norm='log10', rank=rank)
    c.basis = torch.from_numpy(basis).cuda()
    x = np.ones((n, dof))
    x[:, :rank] = 10.0 ** rng.uniform(-300, 300, (n, rank))
    x[0, :rank] = [1.0, 10.0, 100.0, 1e-20, 0.5, 2.0, 1.4142135623730951, 1.4142135623730954, 0.9999999999999999,
                   1.0000000000000002, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 3.0, 7.0, 1e22]
    x[1:6, 0] = [0.0, -0.0, -1.0, np.inf, np.nan]          # a special value makes its row's other columns 0 * inf = NaN
    with np.errstate(divide='ignore', invalid='ignore'):
        want = np.log10(x[:, :rank])
    got = c.compress(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(got[1:6, 0], [-np.inf, -np.inf, np.nan, np.inf, np.nan], equal_nan=True)
    got, want = np.delete(got, slice(1, 6), 0), np.delete(want, slice(1, 6), 0)
    ulp = np.abs(got - want) / np.spacing(np.maximum(np.abs(want), 1e-300))
    assert ulp.max() <= 2.0
    # exact at 1 and at the profile fill 1e-20 (an invalid sample's latents are -20 x column sums in the fused mode);
    # the table version is within 1.3 ulp elsewhere, not correctly rounded: log10(10) is 1 - 2^-53
    assert got[0, 0] == 0.0 and got[0, 3] == -20.0 and np.all(np.abs(got[0, 1:3] - [1.0, 2.0]) <= 2.3e-16)

    z = rng.uniform(-299.0, 299.0, (n, rank))
    z[0] = [0.0, 1.0, 2.0, -20.0, 0.5, -0.5, 22.0, -300.0, 300.0, 308.0, -308.0, -320.0, 309.0, -330.0, 3.0, -3.0]
    z[1:4, 0] = [np.nan, np.inf, -np.inf]                  # (a non-finite latent makes the rest of its row 0 * inf = NaN)
    with np.errstate(over='ignore'):
        want = np.power(10.0, z)
    got = c.reconstruct(torch.from_numpy(z).cuda()).cpu().numpy()
    assert np.isnan(got[1, 0]) and got[2, 0] == np.inf and got[3, 0] == 0.0
    got, want = np.delete(got, slice(1, 4), 0), np.delete(want, slice(1, 4), 0)
    assert np.array_equal(got[:, rank:], np.ones((n - 3, dof - rank)))
    got = got[:, :rank]
    fin = np.isfinite(want) & (want > 1e-300)
    assert np.array_equal(np.isinf(got), np.isinf(want)) and not np.isnan(got).any()
    assert np.max(np.abs(got[fin] - want[fin]) / want[fin]) <= 4.5e-16
    assert got[0, 0] == 1.0 and got[0, 1] == 10.0 and got[0, 2] == 100.0 and got[0, 12] == np.inf
    assert got[0, 11] == pytest.approx(1e-320, rel=1e-3) and got[0, 13] == 0.0      # denormal result, underflow


@pytest.mark.parametrize('norm', ['log10', 'none'])
@pytest.mark.parametrize('n', [1, 777, 100_003])
def test_fused_coupled_compression_matches_two_launch_pipeline(n, norm):
    """pem_coupled_latent_f64_dev == pem_coupled_f64_dev followed by pem_svd_compress_f64_dev, invalid samples included."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.compression import SVDCompression
    from hallthrusterpem_amd.sampling import Design
    rng = np.random.default_rng(3)
    rank = 6
    basis = np.linalg.qr(rng.standard_normal((91, rank)))[0]
    c = SVDCompression(norm=norm, rank=rank)
    c.basis = torch.from_numpy(np.ascontiguousarray(basis)).cuda()
    ref = CoupledBatch(n, profile=True, thruster_qoi=False)
    Design(seed=31).fill(ref.inputs)
    if n > 10:
        ref.inputs[10, 5:9], ref.inputs[11, 5:9] = 0.0, -1.0          # alpha1 = c3 <= 0: invalid, profile 1e-20
    ref.run()
    want = c.compress(ref.j_ion)
    fused = CoupledBatch(n, profile=False, thruster_qoi=False)
    fused.inputs.copy_(ref.inputs)
    got = fused.run_latent(c)
    torch.cuda.synchronize()
    assert got.shape == (n, rank)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-12 * max(1.0, scale)
    # same formulas; the rolled angle loop of the fused mode lets hipcc contract mul+add pairs differently (ulp level)
    assert float(((ref.qoi - fused.qoi).abs() / ref.qoi.abs().clamp_min(1e-300)).max()) < 1e-13
    assert torch.equal(ref.invalid, fused.invalid)
    if n > 10:
        assert ref.invalid[5:9].all()
    with pytest.raises(Exception):
        big = SVDCompression(norm=norm, rank=9)
        big.basis = torch.zeros((91, 9), dtype=torch.float64, device='cuda')
        fused.run_latent(big)


@pytest.mark.gpu
@pytest.mark.parametrize('rank', [1, 2, 3, 5, 7, 8])
def test_fused_coupled_compression_every_rank(rank):
    """coupled_latent_kernel<RANK, LOGN> (csrc/pem_latent.hip) is instantiated per rank: every one against the two-launch
    pipeline, on a batch with whole waves (coalesced 16-byte pieces through LDS) and a ragged tail (per-lane stores)."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.compression import SVDCompression
    from hallthrusterpem_amd.sampling import Design
    n = 64 * 5 + 37
    basis = np.linalg.qr(np.random.default_rng(rank).standard_normal((91, rank)))[0]
    for norm in ('log10', 'none'):
        c = SVDCompression(norm=norm, rank=rank)
        c.basis = torch.from_numpy(np.ascontiguousarray(basis)).cuda()
        ref = CoupledBatch(n, profile=True, thruster_qoi=False)
        Design(seed=40 + rank).fill(ref.inputs)
        ref.inputs[10, 70:73], ref.inputs[11, 70:73] = 0.0, -1.0       # alpha1 = c3 <= 0: invalid, profile 1e-20
        ref.run()
        want = c.compress(ref.j_ion)
        fused = CoupledBatch(n, profile=False, thruster_qoi=False)
        fused.inputs.copy_(ref.inputs)
        out = torch.full((n + 3, rank), float('nan'), dtype=torch.float64, device='cuda')
        got = fused.run_latent(c, out=out[:n])
        torch.cuda.synchronize()
        assert torch.isnan(out[n:]).all()                              # nothing written behind the batch
        assert float((got - want).abs().max()) <= 1e-12 * max(1.0, float(want.abs().max()))
        assert torch.equal(ref.invalid, fused.invalid) and bool(ref.invalid[70:73].all())
