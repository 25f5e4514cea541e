"""Posterior of PEM-v0 calibration parameters given measured ion current density, evaluated for many chains at once.

Restates the flow of `spt100_log_likelihood` / `spt100_log_prior` / `spt100_log_posterior`
(scripts/pem_v0/mcmc.py:57-130) for the `jion` quantity with the TRUE coupled model in place of the surrogate:

    theta (K, n_theta)  ->  inputs of shape (K, M, Ne): operating conditions fixed per experiment e, theta broadcast,
                            every other variable drawn from its prior (M nuisance draws)          mcmc.py:60-63
                        ->  model + Gaussian log-likelihood summed over the Ne x Na measurements   mcmc.py:76-98
                        ->  + discharge-current weight, summed over Ne                              mcmc.py:102-104
                        ->  log-sum-exp over the M nuisance draws (constants dropped)               mcmc.py:106-107

One posterior evaluation is five launches: `pem_log_prior_f64_dev`, `pem_sample_f64_dev` (nuisance draws), two row
scatters, ONE `pem_coupled_loglik_f64_dev` (the profile never leaves LDS) and `pem_loglik_marginal_f64_dev`.  At MCMC batch
sizes (K M Ne ~ 1e4..1e5 samples) every launch is latency-bound, so `capture()` records the whole evaluation --
and `Metropolis` a whole accept/reject step -- into a hipGraph (torch.cuda.CUDAGraph) and replays it.

The reference scripts are stale and untested, its sampler (`mcmciterators` DRAM) and surrogate are third-party and
absent: parity UNPINNED.  The likelihood is checked against the oracle + numpy (tests/test_calibration.py); the
discharge current of the analytic thruster test double is I_d = I_B0 / (1 - 2 a_1) (tests/sim_hallthruster.jl:35-48;
its `c1` is the anomalous-transport coefficient, PEM variable `a_1`).
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .batch import CoupledBatch
from .likelihood import JionLikelihood
from .models.coupled import COUPLED_INPUTS
from .sampling import LOGUNIFORM, NORMAL, PEM_V0_PRIORS, UNIFORM, Design

Q_OVER_M = 1.6e-19 / 2.18e-25           # tests/sim_hallthruster.jl:36-37
OPERATING = ('P_b', 'V_a', 'mdot_a')    # XE_ARRAY columns, mcmc.py:46


def log_prior(theta, names, priors=None):
    """Sum of the log prior densities of `names` at theta (..., len(names)); -inf outside the bounds
    (mcmc.py:110-121).  Works on numpy arrays and on torch tensors (any device)."""
    priors = PEM_V0_PRIORS if priors is None else priors
    is_np = isinstance(theta, np.ndarray)
    if is_np:
        xp, total = np, np.zeros(theta.shape[:-1])
    else:
        import torch
        xp, total = torch, torch.zeros(theta.shape[:-1], dtype=theta.dtype, device=theta.device)
    ninf = -math.inf
    for i, k in enumerate(names):
        p, x = priors[k], theta[..., i]
        if p.kind == UNIFORM:
            lp = xp.where((x >= p.a) & (x <= p.b), xp.zeros_like(x) - math.log(p.b - p.a), ninf)
        elif p.kind == LOGUNIFORM:      # density of 10^U(a, b): 1 / (x ln10 (b - a))
            inside = (x >= 10.0 ** p.a) & (x <= 10.0 ** p.b)
            safe = xp.where(inside, x, xp.ones_like(x))
            lp = xp.where(inside, -xp.log(safe) - math.log(math.log(10.0) * (p.b - p.a)), ninf)
        elif p.kind == NORMAL:
            lp = -0.5 * ((x - p.a) / p.b) ** 2 - math.log(p.b * math.sqrt(2.0 * math.pi))
        else:
            raise ValueError(f'unknown prior kind {p.kind}')
        total = total + lp
    return total


class JionPosterior:
    def __init__(self, theta_names, operating, alpha, y, std, n_chains: int, n_nuisance: int = 100, priors=None,
                 seed: int = 0, discharge=(4.5, 0.2), sweep_radius: float = 1.0, fresh_nuisance: bool = True,
                 device=None):
        """theta_names: calibrated inputs (subset of the 15 coupled inputs, not operating ones);
        operating: (Ne, 3) array of `P_b [Torr], V_a [V], mdot_a [kg/s]` per experiment;
        alpha, y, std: (Ne, Na) measurement angles [rad], current densities and standard deviations at `sweep_radius`;
        discharge: (I_d, sigma) of the extra discharge-current weight (mcmc.py:48-49,102-104) or None;
        fresh_nuisance: new nuisance draws on every evaluation (as the reference) -- inside a captured graph the
        draws are whatever was recorded (common random numbers)."""
        import torch
        self.names = tuple(theta_names)
        for k in self.names:
            if k not in COUPLED_INPUTS or k in OPERATING:
                raise KeyError(f"'{k}' is not a calibratable input of the coupled model")
        op = np.atleast_2d(np.asarray(operating, dtype=np.float64))
        self.K, self.M, self.Ne = int(n_chains), int(n_nuisance), op.shape[0]
        if op.shape[1] != len(OPERATING):
            raise ValueError('operating conditions are rows of (P_b, V_a, mdot_a)')
        self.priors = PEM_V0_PRIORS if priors is None else priors
        self.lik = JionLikelihood(alpha, y, std, device=device)
        if self.lik.n_cond != self.Ne:
            raise ValueError('one row of measurements per operating condition')
        self.n = self.K * self.M * self.Ne
        self.batch = CoupledBatch(self.n, device=self.lik.device, profile=False, sweep_radius=sweep_radius,
                                  thruster_qoi=False)
        self.device = self.batch.device
        self.design = Design(priors=self.priors, seed=seed)
        self.operating = torch.as_tensor(op, device=self.device)                       # (Ne, 3)
        self.theta_rows = [COUPLED_INPUTS.index(k) for k in self.names]
        self.op_rows = [COUPLED_INPUTS.index(k) for k in OPERATING]
        self.discharge = None if discharge is None else (float(discharge[0]), float(discharge[1]))
        self.fresh = bool(fresh_nuisance)
        self.first_index = 0
        self.loglik = torch.empty(self.n, dtype=torch.float64, device=self.device)
        self._view = lambda t: t.view(self.K, self.M, self.Ne)
        # few launches per evaluation (every one is latency-bound): row scatter indices and the prior table on device
        dev_i = lambda rows: torch.as_tensor(rows, dtype=torch.int64, device=self.device)                   # noqa: E731
        self._op_idx, self._theta_idx = dev_i(self.op_rows), dev_i(self.theta_rows)
        self._op_vals = self.operating.T.contiguous()[:, None, None, :]                 # (3, 1, 1, Ne)
        pr = [self.priors[k] for k in self.names]
        self._kind = np.ascontiguousarray([q.kind for q in pr], dtype=np.int32)
        self._a = np.ascontiguousarray([q.a for q in pr], dtype=np.float64)
        self._b = np.ascontiguousarray([q.b for q in pr], dtype=np.float64)
        self._lp = torch.empty(self.K, dtype=torch.float64, device=self.device)
        self._out = torch.empty(self.K, dtype=torch.float64, device=self.device)

    # ------------------------------------------------------------------------------------------------ evaluation
    def assemble_inputs(self, theta):
        """Fill the batch: nuisance draws for everything, then the operating columns and theta broadcast over them."""
        self.design.fill(self.batch.inputs, first_index=self.first_index)
        if self.fresh:
            self.first_index += self.n
        x = self.batch.inputs.view(len(COUPLED_INPUTS), self.K, self.M, self.Ne)
        x.index_copy_(0, self._op_idx, self._op_vals.expand(-1, self.K, self.M, -1))
        x.index_copy_(0, self._theta_idx, theta.T[:, :, None, None].expand(-1, -1, self.M, self.Ne))

    def _marginal(self, log_prior, out):
        import torch
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None                                    # noqa: E731
        x = self.batch.inputs
        d = self.discharge
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pem_loglik_marginal_f64_dev(
                self.K, self.M, self.Ne, p(self.loglik), p(x[COUPLED_INPUTS.index('mdot_a')]) if d else None,
                p(x[COUPLED_INPUTS.index('a_1')]) if d else None, d[0] if d else 0.0, d[1] if d else 1.0, p(log_prior),
                p(out), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def log_likelihood(self, theta, out=None):
        """theta: (K, n_theta) float64 tensor on the device -> (K,) marginal log-likelihoods."""
        import torch
        assert theta.shape == (self.K, len(self.names)) and theta.dtype == torch.float64 and theta.device == self.device
        self.assemble_inputs(theta)
        self.batch.run_loglik(self.lik, out=self.loglik)
        return self._marginal(None, torch.empty_like(self._out) if out is None else out)

    def log_prior(self, theta, out=None):
        """`log_prior(theta, names, priors)` of this module for a (K, n_theta) device tensor (`pem_log_prior_f64_dev`)."""
        import torch
        theta = theta.contiguous()
        out = torch.empty(theta.shape[0], dtype=torch.float64, device=self.device) if out is None else out
        ptr = lambda arr: C.c_void_p(arr.ctypes.data)                                                        # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pem_log_prior_f64_dev(
                theta.shape[0], len(self.names), ptr(self._kind), ptr(self._a), ptr(self._b), C.c_void_p(theta.data_ptr()),
                C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def log_posterior(self, theta, out=None):
        """Prior + likelihood; -inf outside the prior support and where the model produced NaN (mcmc.py:124-130 --
        there only in-support rows are evaluated; here all K rows run, shapes stay static for graph capture).
        Five launches: nuisance draws, two row scatters, model + likelihood, marginalisation + prior."""
        import torch
        assert theta.shape == (self.K, len(self.names)) and theta.dtype == torch.float64 and theta.device == self.device
        self.log_prior(theta, out=self._lp)
        self.assemble_inputs(theta)
        self.batch.run_loglik(self.lik, out=self.loglik)
        return self._marginal(self._lp, torch.empty_like(self._out) if out is None else out)

    # ---------------------------------------------------------------------------------------------- graph capture
    def capture(self):
        """Record `log_posterior` into a hipGraph.  Returns `replay(theta) -> (K,) tensor` (a static output buffer)."""
        import torch
        static_theta = torch.zeros((self.K, len(self.names)), dtype=torch.float64, device=self.device)
        for j, k in enumerate(self.names):                     # a point inside the support for the warm-up runs
            p = self.priors[k]
            static_theta[:, j] = 10.0 ** (0.5 * (p.a + p.b)) if p.kind == LOGUNIFORM else (p.a if p.kind == NORMAL else 0.5 * (p.a + p.b))
        fresh, self.fresh = self.fresh, False                  # a recorded launch carries its sample offset by value
        graph, out = capture_graph(lambda: self.log_posterior(static_theta), self.device)
        self.fresh = fresh

        def replay(theta):
            static_theta.copy_(theta)
            graph.replay()
            return out
        replay.graph, replay.theta, replay.out = graph, static_theta, out
        return replay


def capture_graph(body, device, warmup: int = 2, generator=None):
    """Run `body` (libpem_hip launches and torch ops on the current stream) `warmup` times on a side stream, then
    record it once into a torch.cuda.CUDAGraph (a hipGraph).  A non-default torch `generator` that `body` draws from is
    registered with the graph so that every replay advances its Philox offset.  Returns (graph, body's return value =
    static output)."""
    import torch
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        for _ in range(warmup):
            body()
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize(device)
    graph = torch.cuda.CUDAGraph()
    if generator is not None:
        graph.register_generator_state(generator)
    with torch.cuda.graph(graph):
        out = body()
    return graph, out


class Metropolis:
    """K independent random-walk Metropolis chains advanced together, one hipGraph replay per step.

    A stand-in for the delayed-rejection adaptive Metropolis of the reference (`mcmciterators`, third-party, absent;
    call site mcmc.py:283-300): same role -- draw from `log_posterior` -- simplest valid kernel.  Proposal:
    theta' = theta + scale * N(0, I) with a per-parameter `scale`; the nuisance draws inside the posterior are the
    recorded ones (common random numbers), i.e. the chain targets the M-sample marginal likelihood estimate."""

    def __init__(self, posterior: JionPosterior, theta0, scale, seed: int = 0, use_graph: bool = True):
        import torch
        self.post = posterior
        dev = posterior.device
        self.theta = torch.as_tensor(np.asarray(theta0, dtype=np.float64), device=dev).expand(
            posterior.K, len(posterior.names)).contiguous()
        self.scale = torch.as_tensor(np.asarray(scale, dtype=np.float64), device=dev)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.accepted = torch.zeros(posterior.K, dtype=torch.int64, device=dev)
        fresh, posterior.fresh = posterior.fresh, False
        self.logp = posterior.log_posterior(self.theta).clone()
        posterior.fresh = fresh
        self.steps = 0
        self._graph = None
        if use_graph:           # the warm-up steps before the recording are undone: the chain starts at theta0
            theta_start, logp_start = self.theta.clone(), self.logp.clone()
            posterior.fresh = False
            self._graph, _ = capture_graph(self._step, dev, generator=self.gen)
            posterior.fresh = fresh
            self.theta.copy_(theta_start)
            self.logp.copy_(logp_start)
            self.accepted.zero_()

    def _step(self):
        import torch
        prop = self.theta + self.scale * torch.randn(self.theta.shape, dtype=torch.float64, device=self.theta.device,
                                                     generator=self.gen)
        logp = self.post.log_posterior(prop)
        u = torch.rand(self.post.K, dtype=torch.float64, device=self.theta.device, generator=self.gen)
        accept = torch.log(u) < (logp - self.logp)                 # -inf proposals are never accepted; NaN compares false
        self.theta.copy_(torch.where(accept[:, None], prop, self.theta))
        self.logp.copy_(torch.where(accept, logp, self.logp))
        self.accepted.add_(accept.to(torch.int64))

    def run(self, n_steps: int, keep: bool = True):
        """Advance every chain n_steps; returns the (n_steps, K, n_theta) trace (device tensor) if `keep`."""
        import torch
        trace = (torch.empty((n_steps,) + tuple(self.theta.shape), dtype=torch.float64, device=self.theta.device)
                 if keep else None)
        for i in range(n_steps):
            if self._graph is not None:
                self._graph.replay()
            else:
                self._step()
            if keep:
                trace[i].copy_(self.theta)
        self.steps += n_steps
        return trace

    @property
    def acceptance(self):
        return self.accepted.double() / max(1, self.steps)


class DRAM:
    """K chains of delayed-rejection adaptive Metropolis (Haario, Laine, Mira & Saksman 2006) advanced together -- the
    algorithm behind the reference's `uq.dram(fun, p0, niter, cov0=None, adapt_after=5000, adapt_interval=1000, eps=1e-12,
    gamma=0.1)` (scripts/pem_v0/mcmc.py:297-298; `uqtils` is third-party and absent: parity UNPINNED, the keyword names and
    their meaning are that call's).  Per step and chain:

      stage 1   y1 = x + L z1,  L L^T = C;  accepted with a1 = min(1, pi(y1) / pi(x))
      stage 2   (on rejection) y2 = x + sqrt(gamma) L z2, accepted with
                a2 = min(1, pi(y2) q(y1 | y2) [1 - a1(y2 -> y1)] / (pi(x) q(y1 | x) [1 - a1(x -> y1)]))
      adaptation  after `adapt_after` steps, every `adapt_interval` steps: C = (2.4^2 / d) (cov(chain so far) + eps I), from a
                running mean / scatter matrix per chain (Welford)

    `log_posterior(theta[K, d]) -> logp[K]` is any callable on torch tensors (a `JionPosterior.log_posterior`, or a closed
    form on the CPU in the tests).  Both stages are evaluated for all chains every step (a fixed launch sequence: with
    `use_graph` the step is one hipGraph replay, as `Metropolis`); the adaptation runs between replays and updates `L` in place."""

    def __init__(self, log_posterior, theta0, cov0=None, n_chains: int | None = None, seed: int = 0, adapt_after: int = 5000,
                 adapt_interval: int = 1000, eps: float = 1e-12, gamma: float = 0.1, device=None, use_graph: bool = False):
        import torch
        self.f = log_posterior
        t0 = np.atleast_1d(np.asarray(theta0, dtype=np.float64))
        if t0.ndim == 1:
            t0 = np.broadcast_to(t0, (int(n_chains or 1), t0.size))
        self.K, self.d = t0.shape
        dev = torch.device(device) if device is not None else torch.device('cpu')
        self.theta = torch.as_tensor(np.ascontiguousarray(t0), device=dev).clone()
        c0 = np.eye(self.d) if cov0 is None else np.asarray(cov0, dtype=np.float64)
        if c0.ndim == 1:
            c0 = np.diag(c0)
        self.L = torch.linalg.cholesky(torch.as_tensor(c0, device=dev)).expand(self.K, self.d, self.d).contiguous()
        self.adapt_after, self.adapt_interval, self.eps, self.gamma = int(adapt_after), int(adapt_interval), float(eps), float(gamma)
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.logp = self.f(self.theta).clone()
        self.accepted = torch.zeros((2, self.K), dtype=torch.int64, device=dev)          # per stage
        self.mean = self.theta.clone()                                                    # running moments of the chain
        self.scatter = torch.zeros((self.K, self.d, self.d), dtype=torch.float64, device=dev)
        self.count = 1
        self.steps = 0
        self._graph = None
        if use_graph:
            state = [t.clone() for t in (self.theta, self.logp, self.accepted)]
            self._graph, _ = capture_graph(self._step, dev, generator=self.gen)
            for t, s in zip((self.theta, self.logp, self.accepted), state):
                t.copy_(s)

    def _step(self):
        import torch
        K, d, dev = self.K, self.d, self.theta.device
        x, lp0, L = self.theta, self.logp, self.L
        z = torch.randn((2, K, d), dtype=torch.float64, device=dev, generator=self.gen)
        u = torch.rand((2, K), dtype=torch.float64, device=dev, generator=self.gen)
        y1 = x + torch.einsum('kij,kj->ki', L, z[0])
        lp1 = self.f(y1)
        a1 = torch.exp((lp1 - lp0).clamp(max=0.0))                        # NaN (e.g. -inf - -inf) compares false below: rejected
        acc1 = u[0] < a1
        y2 = x + math.sqrt(self.gamma) * torch.einsum('kij,kj->ki', L, z[1])
        lp2 = self.f(y2)
        a1_rev = torch.exp((lp1 - lp2).clamp(max=0.0))                    # first-stage acceptance of y1 seen from y2
        w = torch.linalg.solve_triangular(L, (y1 - y2).unsqueeze(-1), upper=False).squeeze(-1)
        log_q = -0.5 * ((w * w).sum(dim=1) - (z[0] * z[0]).sum(dim=1))    # log q(y1 | y2) - log q(y1 | x)
        log_a2 = (lp2 - lp0) + log_q + torch.log1p(-a1_rev) - torch.log1p(-a1)
        acc2 = (~acc1) & (torch.log(u[1]) < log_a2)                        # a1 = 1 is accepted at stage 1; a1_rev = 1 gives -inf
        self.theta.copy_(torch.where(acc1[:, None], y1, torch.where(acc2[:, None], y2, x)))
        self.logp.copy_(torch.where(acc1, lp1, torch.where(acc2, lp2, lp0)))
        self.accepted[0].add_(acc1.to(torch.int64))
        self.accepted[1].add_(acc2.to(torch.int64))

    def _adapt(self):
        import torch
        n = self.count
        if n < 2:
            return
        cov = self.scatter / (n - 1) + self.eps * torch.eye(self.d, dtype=torch.float64, device=self.theta.device)
        self.L.copy_(torch.linalg.cholesky((2.4 ** 2 / self.d) * cov))

    def run(self, n_steps: int, keep: bool = True):
        """Advance every chain n_steps; returns the (n_steps, K, d) trace if `keep`."""
        import torch
        trace = torch.empty((n_steps, self.K, self.d), dtype=torch.float64, device=self.theta.device) if keep else None
        for i in range(n_steps):
            if self._graph is not None:
                self._graph.replay()
            else:
                self._step()
            self.steps += 1
            self.count += 1                                               # Welford update of the chain's mean and scatter
            delta = self.theta - self.mean
            self.mean += delta / self.count
            self.scatter += delta[:, :, None] * (self.theta - self.mean)[:, None, :]
            if self.steps >= self.adapt_after and (self.steps - self.adapt_after) % self.adapt_interval == 0:
                self._adapt()
            if keep:
                trace[i].copy_(self.theta)
        return trace

    @property
    def acceptance(self):
        """(2, K): fraction of steps accepted at the first and at the delayed stage"""
        return self.accepted.double() / max(1, self.steps)
