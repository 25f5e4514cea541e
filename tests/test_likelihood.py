"""jion likelihood kernel against a numpy restatement of the stated formula (np.interp on the mirrored grid)."""
import numpy as np
import pytest
from scipy.special import logsumexp

pytestmark = pytest.mark.gpu


def test_loglik_matches_numpy():
    import torch
    from hallthrusterpem_amd import drivers
    from hallthrusterpem_amd.likelihood import JionLikelihood
    rng = np.random.default_rng(3)
    T, M, Ne, Na = 5, 37, 8, 43
    n = T * M * Ne
    res = drivers.forward_uq(n, seed=8, keep_profile=True)
    j = res['j_ion'].reshape(T, M, Ne, 91)
    alpha = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, (Ne, Na)), axis=1)
    alpha[0, 0], alpha[0, -1], alpha[1, 3] = -np.pi / 2, np.pi / 2, 0.0            # grid end points and the centreline
    grid = np.linspace(0, np.pi / 2, 91)
    jh = j.cpu().numpy()
    y = np.stack([np.interp(np.abs(alpha[e]), grid, jh[0, 0, e]) for e in range(Ne)]) * rng.lognormal(0, 0.2, (Ne, Na))
    std = 0.2 * y + 0.05
    lk = JionLikelihood(alpha, y, std)
    got = lk.per_sample(j).cpu().numpy()
    want = np.empty((T, M, Ne))
    for t in range(T):
        for m in range(M):
            for e in range(Ne):
                model = np.interp(np.abs(alpha[e]), grid, jh[t, m, e])
                want[t, m, e] = np.sum(-0.5 * ((y[e] - model) / std[e]) ** 2)
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) < 1e-11
    L = lk.log_likelihood(j).cpu().numpy()
    assert np.allclose(L, logsumexp(want.sum(-1), axis=-1), rtol=1e-12, atol=1e-9)
    with pytest.raises(ValueError):
        JionLikelihood(np.array([[2.0]]), np.array([[1.0]]), np.array([[1.0]]))
    assert lk.per_sample(torch.empty((0, 91), dtype=torch.float64, device='cuda')).shape == (0,)


@pytest.mark.parametrize('n', [8 * 37 * 5, 100003])
def test_fused_coupled_loglik_matches_two_launch_pipeline(n):
    """pem_coupled_loglik_f64_dev (profile reduced in LDS) == pem_coupled_f64_dev followed by pem_jion_loglik_f64_dev."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.likelihood import JionLikelihood
    from hallthrusterpem_amd.sampling import Design
    rng = np.random.default_rng(11)
    Ne, Na = 8, 43
    alpha = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, (Ne, Na)), axis=1)
    alpha[0, 0], alpha[0, -1], alpha[1, 3] = -np.pi / 2, np.pi / 2, 0.0
    y = rng.lognormal(1.0, 1.0, (Ne, Na))
    lk = JionLikelihood(alpha, y, 0.2 * y + 0.05)
    ref = CoupledBatch(n, profile=True, thruster_qoi=False)
    Design(seed=21).fill(ref.inputs, method='mc')
    ref.inputs[10, 5:9], ref.inputs[11, 5:9] = 0.0, -1.0     # alpha1 = c3 <= 0: invalid, profile 1e-20 in both paths
    ref.run()
    want = lk.per_sample(ref.j_ion)
    fused = CoupledBatch(n, profile=False, thruster_qoi=False)
    fused.inputs.copy_(ref.inputs)
    got = fused.run_loglik(lk)
    torch.cuda.synchronize()
    w, g = want.cpu().numpy(), got.cpu().numpy()
    assert np.array_equal(np.isnan(w), np.isnan(g))
    ok = ~np.isnan(w)
    assert np.max(np.abs(g[ok] - w[ok]) / np.maximum(1.0, np.abs(w[ok]))) < 1e-12
    assert torch.equal(ref.qoi, fused.qoi) and torch.equal(ref.invalid, fused.invalid)
    assert ref.invalid[5:9].all()
