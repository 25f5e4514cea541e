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
