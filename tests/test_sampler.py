"""Sampler: Philox known-answer vectors and design properties (CPU), HIP sampler vs the numpy restatement (GPU).
The driver layer of the reference is third-party (amisc/uqtils): parity unpinned, these tests check the library
against its own stated formulas."""
import numpy as np
import pytest

from oracle import sampler_np as snp


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    cases = [((0, 0, 0, 0), (0, 0), '6627e8d5 e169c58d bc57ac4c 9b00dbd8'),
             ((0xffffffff,) * 4, (0xffffffff,) * 2, '408f276d 41c83b0e a20bc7c6 6d5451fd'),
             ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), 'd16cfe09 94fdcceb 5001e420 24126ea1')]
    for c, k, want in cases:
        got = snp.philox4x32_10(*c, *k)
        assert ' '.join(f'{int(x):08x}' for x in got) == want


def test_numpy_design_properties():
    kind = [snp.UNIFORM, snp.LOGUNIFORM, snp.NORMAL, snp.UNIFORM, snp.UNIFORM]
    a = [2.0, -8.0, 1.0, 0.0, -1.0]
    b = [5.0, -4.0, 0.5, 1.0, 1.0]
    x = snp.sample(4000, 0, 42, 0, kind, a, b)
    assert x.shape == (5, 4000)
    assert x[0].min() >= 2 and x[0].max() < 5 and abs(x[0].mean() - 3.5) < 0.05
    assert x[1].min() >= 1e-8 and x[1].max() <= 1e-4 and abs(np.log10(x[1]).mean() + 6) < 0.06
    assert abs(x[2].mean() - 1.0) < 0.03 and abs(x[2].std() - 0.5) < 0.02
    # sharding invariance: any slice of the design is the same numbers
    y = np.concatenate([snp.sample(1500, 0, 42, 0, kind, a, b), snp.sample(2500, 1500, 42, 0, kind, a, b)], axis=1)
    assert np.array_equal(x, y)
    assert not np.array_equal(x, snp.sample(4000, 0, 43, 0, kind, a, b))
    # Saltelli blocks: A and B independent, AB_d = A with column d from B
    A, B = snp.sample(100, 0, 7, 0, kind, a, b, swap_dim=-1), snp.sample(100, 0, 7, 0, kind, a, b, swap_dim=-2)
    AB3 = snp.sample(100, 0, 7, 0, kind, a, b, swap_dim=3)
    assert np.array_equal(AB3[3], B[3]) and all(np.array_equal(AB3[d], A[d]) for d in (0, 1, 2, 4))
    assert not np.array_equal(A[0], B[0])
    # Latin hypercube: exactly one sample per stratum in every dimension, however the design is sliced
    for n in (10, 1000, 1025):
        L = np.concatenate([snp.sample(n // 3, 0, 5, 0, [0] * 4, [0.0] * 4, [1.0] * 4, mode='lhs', n_total=n),
                            snp.sample(n - n // 3, n // 3, 5, 0, [0] * 4, [0.0] * 4, [1.0] * 4, mode='lhs', n_total=n)], axis=1)
        for d in range(4):
            assert np.array_equal(np.sort(np.floor(L[d] * n).astype(int)), np.arange(n))
        assert not np.array_equal(np.floor(L[0] * n), np.floor(L[1] * n))      # dimensions permuted independently


def test_prior_table_matches_appendix_a():
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    from hallthrusterpem_amd.sampling import LOGUNIFORM, PEM_V0_PRIORS, UNIFORM
    assert tuple(PEM_V0_PRIORS) == COUPLED_INPUTS
    assert PEM_V0_PRIORS['P_b'].kind == LOGUNIFORM and (PEM_V0_PRIORS['P_b'].a, PEM_V0_PRIORS['P_b'].b) == (-8.0, -4.0)
    assert PEM_V0_PRIORS['c3'].kind == UNIFORM and PEM_V0_PRIORS['c3'].b == 1.570796
    assert PEM_V0_PRIORS['a_1'].kind == LOGUNIFORM and 10 ** PEM_V0_PRIORS['a_1'].a == pytest.approx(0.00316)


@pytest.mark.gpu
def test_hip_sampler_matches_numpy_restatement():
    import torch
    from hallthrusterpem_amd.sampling import Design, Prior, NORMAL, PEM_V0_PRIORS
    d = Design(seed=2026, stream=3)
    n, first = 10_007, 123_456_789_012
    got = d.sample(n, first_index=first).cpu().numpy()
    want = snp.sample(n, first, 2026, 3, d.kind, d.a, d.b)
    lin = d.kind == 0
    assert np.array_equal(got[lin], want[lin])                       # uniform dims: same IEEE operations
    assert np.max(np.abs(got[~lin] / want[~lin] - 1)) < 4e-15        # log-uniform: exp() differs by an ulp at most
    # Saltelli blocks and sharding invariance on the device
    A, B, AB = d.sample(500), d.sample(500, swap_dim=-2), d.sample(500, swap_dim=8)
    assert torch.equal(AB[8], B[8]) and torch.equal(AB[7], A[7]) and not torch.equal(A[8], B[8])
    whole = d.sample(3000)
    parts = torch.cat([d.sample(1000), d.sample(2000, first_index=1000)], dim=1)
    assert torch.equal(whole, parts)
    # the same numbers written tile-interleaved ([tiles][15][64], the layout of pem_coupled_tiled_f64_dev), ragged last tile
    for m in (3000, 64, 1):
        tiled = torch.zeros(((m + 63) // 64, d.ndim, 64), dtype=torch.float64, device='cuda')
        d.fill_tiled(tiled, m, first_index=0)
        assert torch.equal(tiled.permute(1, 0, 2).reshape(d.ndim, -1)[:, :m], whole[:, :m])
    tiled = torch.zeros((8, d.ndim, 64), dtype=torch.float64, device='cuda')
    d.fill_tiled(tiled, 500, swap_dim=8)
    assert torch.equal(tiled.permute(1, 0, 2).reshape(d.ndim, -1)[:, :500], AB)
    # Latin hypercube
    L = d.sample(4097, method='lhs', n_total=4097).cpu().numpy()
    Lw = snp.sample(4097, 0, 2026, 3, d.kind, d.a, d.b, mode='lhs', n_total=4097)
    assert np.array_equal(L[lin], Lw[lin])
    u = (L[1] - 200.0) / 200.0                                       # V_a is U(200, 400)
    assert np.array_equal(np.sort(np.floor(u * 4097).astype(int)), np.arange(4097))
    # a normal prior
    pri = dict(PEM_V0_PRIORS)
    pri['V_vac'] = Prior(NORMAL, 30.0, 2.0, 'test')
    g = Design(priors=pri, seed=1).sample(200_000)[3]
    assert abs(float(g.mean()) - 30.0) < 0.02 and abs(float(g.std()) - 2.0) < 0.02
    wn = snp.sample(1000, 0, 1, 0, [NORMAL], [30.0], [2.0])
    gn = Design(priors={'x': Prior(NORMAL, 30.0, 2.0, 't')}, names=('x',), seed=1).sample(1000).cpu().numpy()
    assert np.max(np.abs(gn - wn)) < 1e-12
