"""Posterior of calibration parameters given measured ion current density (hallthrusterpem_amd/calibration.py):
the prior against scipy, the likelihood against the oracle + numpy, graph replay against eager evaluation, and a
Metropolis run on synthetic data.  The reference flow (scripts/pem_v0/mcmc.py:57-130) is stale/untested: unpinned."""
import numpy as np
import pytest
from scipy import stats
from scipy.special import logsumexp

from hallthrusterpem_amd.calibration import OPERATING, Q_OVER_M, log_prior
from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
from hallthrusterpem_amd.sampling import NORMAL, PEM_V0_PRIORS, Prior

NAMES = ('T_e', 'c4', 'V_vac')


def test_log_prior_matches_scipy_and_is_minus_inf_outside_the_support():
    rng = np.random.default_rng(0)
    priors = dict(PEM_V0_PRIORS, V_vac=Prior(NORMAL, 30.0, 5.0, 'test'))
    theta = np.stack([rng.uniform(0.5, 5.5, 200), 10.0 ** rng.uniform(17.5, 22.5, 200), rng.normal(30, 10, 200)], axis=-1)
    got = log_prior(theta, NAMES, priors)
    want = (stats.uniform(1.0, 4.0).logpdf(theta[:, 0]) + stats.loguniform(1e18, 1e22).logpdf(theta[:, 1])
            + stats.norm(30.0, 5.0).logpdf(theta[:, 2]))
    assert np.array_equal(np.isneginf(got), np.isneginf(want)) and np.isneginf(got).any() and np.isfinite(got).any()
    ok = np.isfinite(want)
    assert np.allclose(got[ok], want[ok], rtol=1e-13, atol=0)
    import torch
    assert np.array_equal(log_prior(torch.as_tensor(theta), NAMES, priors).numpy(), got)
    assert log_prior(theta.reshape(4, 50, 3), NAMES, priors).shape == (4, 50)


def _problem(K, M, seed=0, Ne=6, Na=31):
    rng = np.random.default_rng(seed)
    operating = np.stack([10.0 ** rng.uniform(-6, -4.5, Ne), rng.uniform(250, 350, Ne), rng.uniform(4e-6, 6e-6, Ne)], axis=1)
    alpha = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, (Ne, Na)), axis=1)
    return operating, alpha


@pytest.mark.gpu
def test_log_likelihood_matches_oracle_and_graph_replay_matches_eager():
    import torch
    from oracle import oracle_ctypes as oc
    from hallthrusterpem_amd import constants
    from hallthrusterpem_amd.calibration import JionPosterior
    K, M = 5, 13
    operating, alpha = _problem(K, M)
    Ne, Na = alpha.shape
    rng = np.random.default_rng(1)
    y = rng.lognormal(0.0, 1.0, (Ne, Na))
    std = 0.3 * y + 0.1
    names = ('c0', 'c2', 'c4', 'T_e')
    post = JionPosterior(names, operating, alpha, y, std, n_chains=K, n_nuisance=M, seed=4, fresh_nuisance=False)
    theta = torch.tensor([[0.3, 2.0, 1e20, 2.5], [0.6, -5.0, 3e19, 4.0], [0.1, 10.0, 1e21, 1.5],
                          [1.5, 0.0, 1e20, 3.0],                       # c0 outside U(0, 1): prior -inf
                          [0.5, 0.0, 1e20, 3.0]], dtype=torch.float64, device='cuda')
    got = post.log_likelihood(theta).cpu().numpy()
    x = post.batch.inputs.cpu().numpy()                                   # the samples the evaluation used
    xv = x.reshape(len(COUPLED_INPUTS), K, M, Ne)
    for j, k in enumerate(OPERATING):
        assert np.array_equal(xv[COUPLED_INPUTS.index(k)], np.broadcast_to(operating[:, j], (K, M, Ne)))
    for j, k in enumerate(names):
        assert np.array_equal(xv[COUPLED_INPUTS.index(k)], np.broadcast_to(theta.cpu().numpy()[:, j, None, None], (K, M, Ne)))
    assert len(np.unique(xv[COUPLED_INPUTS.index('c5')])) == K * M * Ne                 # nuisance: all different
    ref = oc.coupled(dict(zip(COUPLED_INPUTS, x)), torr2pa=constants.TORR_2_PA)
    grid = np.linspace(0, np.pi / 2, 91)
    j = ref['j_ion'].reshape(K, M, Ne, 91)
    ll = np.zeros((K, M))
    for k in range(K):
        for m in range(M):
            for e in range(Ne):
                model = np.interp(np.abs(alpha[e]), grid, j[k, m, e])
                ll[k, m] += np.sum(-0.5 * ((y[e] - model) / std[e]) ** 2)
    i_d = Q_OVER_M * xv[COUPLED_INPUTS.index('mdot_a')] / (1.0 - 2.0 * xv[COUPLED_INPUTS.index('a_1')])
    ll += np.sum(-0.5 * ((4.5 - i_d) / 0.2) ** 2, axis=-1)
    want = logsumexp(ll, axis=-1)
    assert np.allclose(got, want, rtol=1e-10, atol=1e-8)

    from hallthrusterpem_amd.calibration import log_prior as log_prior_table
    lp_dev, lp_host = post.log_prior(theta).cpu().numpy(), log_prior_table(theta.cpu().numpy(), names)
    assert np.array_equal(np.isneginf(lp_dev), np.isneginf(lp_host)) and np.isneginf(lp_dev[3])
    assert np.allclose(lp_dev[[0, 1, 2, 4]], lp_host[[0, 1, 2, 4]], rtol=1e-14)
    eager = post.log_posterior(theta).clone()
    assert torch.isneginf(eager[3]) and torch.isfinite(eager[[0, 1, 2, 4]]).all()
    assert np.allclose(eager.cpu().numpy()[[0, 1, 2, 4]], (want + post.log_prior(theta).cpu().numpy())[[0, 1, 2, 4]], rtol=1e-10)
    replay = post.capture()
    assert torch.equal(replay(theta), eager)                               # same launches, same bits
    theta2 = theta.flip(0).contiguous()
    assert torch.equal(replay(theta2), post.log_posterior(theta2))


@pytest.mark.gpu
def test_fresh_nuisance_draws_advance_the_design_counter():
    import torch
    from hallthrusterpem_amd.calibration import JionPosterior
    operating, alpha = _problem(2, 8)
    y = np.ones_like(alpha)
    post = JionPosterior(('c0',), operating, alpha, y, 0.5 * y, n_chains=2, n_nuisance=8, seed=1)
    theta = torch.full((2, 1), 0.4, dtype=torch.float64, device='cuda')
    a = post.log_likelihood(theta).clone()
    x1 = post.batch.inputs.clone()
    b = post.log_likelihood(theta).clone()
    assert post.first_index == 2 * post.n and not torch.equal(x1, post.batch.inputs) and not torch.equal(a, b)
    with pytest.raises(KeyError):
        JionPosterior(('V_a',), operating, alpha, y, y, n_chains=1)


@pytest.mark.gpu
def test_metropolis_chains_recover_a_synthetic_truth():
    import torch
    from hallthrusterpem_amd.calibration import JionPosterior, Metropolis
    from hallthrusterpem_amd.models.coupled import pem_v0_coupled
    K, M, Ne, Na = 32, 8, 6, 25
    operating, alpha = _problem(K, M, seed=3, Ne=Ne, Na=Na)
    truth = {'c0': 0.35, 'c3': 0.6}
    # every non-calibrated, non-operating input gets a (nearly) degenerate prior so that the data pin theta
    nominal = {'T_e': 3.0, 'V_vac': 30.0, 'Pstar': 5e-5, 'P_T': 5e-5, 'a_1': 0.02, 'c1': 0.3, 'c2': 5.0, 'c4': 1e20,
               'c5': 1e16, 'sigma_cex': 55e-20}
    priors = dict(PEM_V0_PRIORS)
    for k, v in nominal.items():
        priors[k] = Prior(NORMAL, v, 1e-9 * abs(v), 'test: pinned')
    inputs = {k: np.full(Ne, v) for k, v in {**nominal, **truth}.items()}
    for j, k in enumerate(OPERATING):
        inputs[k] = operating[:, j]
    out = pem_v0_coupled(inputs)
    grid = np.linspace(0, np.pi / 2, 91)
    y = np.stack([np.interp(np.abs(alpha[e]), grid, out['j_ion'][e]) for e in range(Ne)])
    std = 0.05 * y + 1e-3
    post = JionPosterior(tuple(truth), operating, alpha, y, std, n_chains=K, n_nuisance=M, priors=priors, seed=2,
                         discharge=None)
    mh = {g: Metropolis(post, [0.5, 0.9], scale=[0.02, 0.02], seed=5, use_graph=g) for g in (False, True)}
    trace = mh[True].run(600)
    assert trace.shape == (600, K, 2)
    tail = trace[300:].reshape(-1, 2).mean(0).cpu().numpy()
    assert abs(tail[0] - truth['c0']) < 0.03 and abs(tail[1] - truth['c3']) < 0.03
    acc = mh[True].acceptance.mean().item()
    assert 0.02 < acc < 0.95
    # the eager stepper advances the same chain law (same posterior; its own random numbers)
    t2 = mh[False].run(150)
    assert torch.isfinite(t2).all() and torch.isfinite(mh[False].logp).all()
    # the delayed-rejection adaptive sampler on the same posterior, one hipGraph replay per step (both stages inside it)
    from hallthrusterpem_amd.calibration import DRAM
    post.fresh = False                                   # recorded nuisance draws: the target does not change between replays
    dr = DRAM(post.log_posterior, np.broadcast_to([0.5, 0.9], (K, 2)), cov0=[4e-4, 4e-4], seed=7, adapt_after=150, adapt_interval=50,
              gamma=0.1, device=post.device, use_graph=True)
    t3 = dr.run(600)
    tail3 = t3[300:].reshape(-1, 2).mean(0).cpu().numpy()
    assert abs(tail3[0] - truth['c0']) < 0.03 and abs(tail3[1] - truth['c3']) < 0.03
    assert float(dr.acceptance[1].mean()) > 0.0 and torch.isfinite(dr.logp).all()


@pytest.mark.gpu
def test_marginal_kernel_edge_cases_against_scipy():
    """pem_loglik_marginal_f64_dev: log-sum-exp over draws with NaN, -inf and widely spread values; prior masking."""
    import ctypes as C
    import torch
    from hallthrusterpem_amd import _lib
    rng = np.random.default_rng(5)
    K, M, Ne = 7, 333, 5
    ll = rng.normal(-50.0, 30.0, (K, M, Ne))
    ll[1] *= 1e3                                   # spread far beyond exp's range: needs the max shift
    ll[2, 17, 3] = np.nan                          # a NaN draw poisons its chain
    ll[3] = -np.inf                                # all draws impossible
    ll[4, ::2] = -np.inf                           # some draws impossible
    mdot, a1 = rng.uniform(4e-6, 6e-6, (K, M, Ne)), rng.uniform(0.01, 0.1, (K, M, Ne))
    lp = np.array([0.5, -1.0, 2.0, 3.0, -np.inf, 1.0, np.nan])
    dev = lambda a: torch.as_tensor(a, device='cuda')                                                    # noqa: E731
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None                                    # noqa: E731
    lib = _lib.load()
    s = ll.sum(-1)
    with np.errstate(invalid='ignore'):
        want = {'plain': logsumexp(s, axis=-1)}
        i_d = Q_OVER_M * mdot / (1.0 - 2.0 * a1)
        want['discharge'] = logsumexp(s + np.sum(-0.5 * ((4.5 - i_d) / 0.2) ** 2, axis=-1), axis=-1)
    want['plain'][2] = want['discharge'][2] = np.nan
    t_ll, t_m, t_a, t_lp = dev(ll), dev(mdot), dev(a1), dev(lp)
    out = torch.empty(K, dtype=torch.float64, device='cuda')
    for key, (m_, a_) in {'plain': (None, None), 'discharge': (t_m, t_a)}.items():
        _lib.check(lib.pem_loglik_marginal_f64_dev(K, M, Ne, p(t_ll), p(m_), p(a_), 4.5, 0.2, None, p(out), None))
        got = out.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want[key])) and np.isnan(got[2]) and np.isneginf(got[3])
        ok = np.isfinite(want[key])
        assert np.allclose(got[ok], want[key][ok], rtol=1e-12, atol=0)
        _lib.check(lib.pem_loglik_marginal_f64_dev(K, M, Ne, p(t_ll), p(m_), p(a_), 4.5, 0.2, p(t_lp), p(out), None))
        post = out.cpu().numpy()
        assert np.isneginf(post[[2, 3, 4, 6]]).all()            # NaN likelihood, -inf likelihood, -inf prior, NaN prior
        assert np.allclose(post[[0, 1, 5]], (want[key] + lp)[[0, 1, 5]], rtol=1e-12)
    assert lib.pem_loglik_marginal_f64_dev(K, 0, Ne, p(t_ll), None, None, 0.0, 1.0, None, p(out), None) == _lib.PEM_ERR_INVALID_ARG
    assert lib.pem_loglik_marginal_f64_dev(K, M, Ne, p(t_ll), p(t_m), p(t_a), 4.5, 0.0, None, p(out), None) == _lib.PEM_ERR_INVALID_ARG


def test_dram_recovers_a_correlated_gaussian_and_its_delayed_stage_accepts():
    """calibration.DRAM on a closed-form target (CPU tensors): the adapted chains reproduce mean and covariance of a correlated
    3-d Gaussian started far from it with a poor proposal; the delayed stage contributes acceptances; flat-prior bounds
    (-inf outside) are never crossed."""
    import torch
    from hallthrusterpem_amd.calibration import DRAM
    mu = torch.tensor([1.0, -2.0, 0.5], dtype=torch.float64)
    A = torch.tensor([[1.0, 0.0, 0.0], [0.8, 0.6, 0.0], [-0.3, 0.2, 0.4]], dtype=torch.float64)
    sigma = A @ A.T
    prec = torch.linalg.inv(sigma)

    def logp(t):
        r = t - mu
        lp = -0.5 * torch.einsum('ki,ij,kj->k', r, prec, r)
        return torch.where(t[:, 2] > -1.0, lp, torch.full_like(lp, -float('inf')))      # a hard bound on one coordinate
    s = DRAM(logp, [0.0, 0.0, 0.0], cov0=np.diag([4.0, 4.0, 4.0]), n_chains=48, seed=3, adapt_after=400, adapt_interval=100, gamma=0.1)
    s.run(800, keep=False)
    trace = s.run(3000)
    flat = trace.reshape(-1, 3)
    assert float(flat[:, 2].min()) > -1.0
    assert torch.allclose(flat.mean(dim=0), mu, atol=0.08)
    assert torch.allclose(torch.cov(flat.T), sigma, atol=0.12)
    acc = s.acceptance
    assert 0.15 < float(acc[0].mean()) < 0.6 and float(acc[1].mean()) > 0.02
    # without adaptation the 2-sigma-per-axis start proposal accepts far less at stage 1; its delayed stage (gamma = 0.1) does the work
    fixed = DRAM(logp, mu.numpy(), cov0=np.diag([4.0, 4.0, 4.0]), n_chains=48, seed=4, adapt_after=10 ** 9)
    fixed.run(1500, keep=False)
    assert float(fixed.acceptance[0].mean()) < 0.5 * float(acc[0].mean()) and float(fixed.acceptance[1].mean()) > 2.0 * float(fixed.acceptance[0].mean())
