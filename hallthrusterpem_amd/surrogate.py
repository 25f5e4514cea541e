"""Sparse-grid Lagrange surrogate of the coupled PEM-v0 QoIs with adaptive index-set refinement.

Follows the SHAPE of the reference's surrogate training (scripts/fit_surr.py:101-116: `System.fit(num_refine=1000,
max_iter=..., targets=...)`; index-set bookkeeping as restated in scripts/pem_v0/monte_carlo.py:708-767: a
downward-closed set of activated multi-indices plus the candidate set one unit vector away).  The surrogate itself is
amisc's (third-party, absent): PARITY UNPINNED.  What is built here, in its own words:

  * coordinates: every varied input is mapped to t in [-1, 1] over its prior support (linear, or linear in log10 for
    log-uniform priors);
  * level l in a dimension: 1 node (t = 0) for l = 0, 2^l + 1 Chebyshev-Lobatto nodes for l >= 1 (nested);
  * a multi-index beta = tensor-product Lagrange interpolant on the nodes of its levels (inactive dimensions at t = 0);
  * the surrogate: the combination technique  f = sum_beta c_beta I_beta,  c_beta = sum_{e in {0,1}^D, beta+e in I} (-1)^|e|
    over a downward-closed index set I;
  * adaptive refinement: each iteration activates the candidate with the largest error indicator
    mean |f_{I + beta} - f_I| over `num_refine` random points (evaluated with the batched HIP predict kernel), relative
    to the QoI range, and adds its forward neighbours that keep the set downward closed.

True-model evaluations at grid nodes go through one `pem_coupled_f64_dev` launch per new index; predictions through
`pem_sparse_predict_f64_dev` (csrc/pem_surrogate.hip).
"""
import ctypes as C
import itertools

import numpy as np

from . import _lib, sampling
from .batch import CoupledBatch
from .models.coupled import COUPLED_INPUTS

MAX_ACTIVE = 3
MAX_LEVEL = 3


def nodes(level: int) -> np.ndarray:
    if level == 0:
        return np.zeros(1)
    m = 2 ** level + 1
    x = -np.cos(np.pi * np.arange(m) / (m - 1))
    x[(m - 1) // 2] = 0.0                      # exact centre node (cos(pi/2) is 6e-17 in floating point)
    return x


class SparseGridSurrogate:
    def __init__(self, varied, fixed: dict | None = None, priors=None, qoi=('V_cc', 'div_angle', 'T_c'), device=None):
        import torch
        self.priors = dict(sampling.PEM_V0_PRIORS if priors is None else priors)
        self.varied = tuple(varied)
        self.fixed = dict(fixed or {})
        missing = [k for k in COUPLED_INPUTS if k not in self.varied and k not in self.fixed]
        if missing:
            raise ValueError(f'inputs neither varied nor fixed: {missing}')
        self.qoi = tuple(qoi)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.D = len(self.varied)
        self.index_set = []            # activated multi-indices (downward closed)
        self.candidates = []
        self.values = {}               # beta -> [prod(m)][n_out] numpy array of true-model QoIs at the grid nodes
        self.model_evals = 0
        self._tables = None
        zero = (0,) * self.D
        self._ensure_values(zero)
        self._activate(zero)

    # ---- coordinates ----------------------------------------------------------------------------------------------
    def to_physical(self, t):
        """[D][n] normalised coordinates -> dict of physical inputs (numpy)."""
        t = np.asarray(t, dtype=np.float64)
        out = {}
        for d, k in enumerate(self.varied):
            p = self.priors[k]
            u = 0.5 * (t[d] + 1.0)
            v = p.a + (p.b - p.a) * u
            out[k] = 10.0 ** v if p.kind == sampling.LOGUNIFORM else v
        return out

    # ---- true model at the nodes of one multi-index ------------------------------------------------------------------
    def _grid(self, beta):
        axes = [nodes(l) for l in beta]
        return np.array(list(itertools.product(*axes)), dtype=np.float64).T.reshape(self.D, -1)     # [D][prod m]

    def _ensure_values(self, *betas):
        """True-model values at the grid nodes of the multi-indices that have none yet: ONE coupled launch for all of them."""
        new = [b for b in dict.fromkeys(betas) if b not in self.values]
        if not new:
            return
        import torch
        grids = [self._grid(b) for b in new]
        t = np.concatenate(grids, axis=1)
        x = self.to_physical(t)
        n = t.shape[1]
        batch = CoupledBatch(n, device=self.device, profile=False)
        full = {k: np.full(n, float(self.fixed[k])) for k in self.fixed}
        full.update(x)
        batch.set_inputs(full)
        batch.run()
        o = batch.outputs()
        y = torch.stack([o[k] for k in self.qoi], dim=1).cpu().numpy()          # (the copy synchronises)
        off = 0
        for b, g in zip(new, grids):
            self.values[b] = y[off:off + g.shape[1]]
            off += g.shape[1]
        self.model_evals += n

    # ---- index-set bookkeeping (shape of monte_carlo.py:714-747) ------------------------------------------------------
    def _admissible(self, beta):
        if sum(1 for l in beta if l > 0) > MAX_ACTIVE or max(beta) > MAX_LEVEL:
            return False
        for d in range(self.D):            # downward closed: every backward neighbour is active
            if beta[d] > 0:
                back = beta[:d] + (beta[d] - 1,) + beta[d + 1:]
                if back not in self.index_set:
                    return False
        return True

    def _activate(self, beta):
        if beta in self.candidates:
            self.candidates.remove(beta)
        self.index_set.append(beta)
        for d in range(self.D):
            new = beta[:d] + (beta[d] + 1,) + beta[d + 1:]
            if new not in self.index_set and new not in self.candidates and self._admissible(new):
                self.candidates.append(new)
        self._tables = None

    @staticmethod
    def combination_coefficients(index_set):
        """c_beta = sum over e in {0,1}^D with beta + e in I of (-1)^|e| (only dimensions where beta + e_d can be in I)."""
        members = set(index_set)
        coefs = {}
        for beta in index_set:
            free = [d for d in range(len(beta)) if beta[:d] + (beta[d] + 1,) + beta[d + 1:] in members]
            c = 0
            for r in range(len(free) + 1):
                for dims in itertools.combinations(free, r):
                    nb = list(beta)
                    for d in dims:
                        nb[d] += 1
                    if tuple(nb) in members:
                        c += (-1) ** r
            coefs[beta] = c
        return coefs

    # ---- device tables + predict ------------------------------------------------------------------------------------
    @staticmethod
    def combination_delta(index_set, cand):
        """Change of the combination coefficients when `cand` joins `index_set`: every term of c_beta's sum that has
        beta + e = cand, i.e. {cand - e: (-1)^|e|} over e in {0,1}^D with cand - e in the set (or e = 0).  Equal to
        combination_coefficients(index_set + [cand]) - combination_coefficients(index_set); 2^(active dims of cand) terms."""
        members = set(index_set)
        active = [d for d in range(len(cand)) if cand[d] > 0]
        delta = {}
        for r in range(len(active) + 1):
            for dims in itertools.combinations(active, r):
                b = list(cand)
                for d in dims:
                    b[d] -= 1
                b = tuple(b)
                if r == 0 or b in members:
                    delta[b] = (-1) ** r
        return delta

    def _build_tables(self, index_set, unit_coefficients: bool = False):
        """Device tables of the kernel for `index_set`: the grids with a non-zero combination coefficient, or -- with
        `unit_coefficients` -- every grid with coefficient 1 (for `grid_values`)."""
        import torch
        coefs = {b: 1 for b in index_set} if unit_coefficients else self.combination_coefficients(index_set)
        used = [b for b in index_set if coefs[b] != 0]
        idx = np.zeros((len(used), 2 + 2 * MAX_ACTIVE), dtype=np.int32)
        vals, off = [], 0
        for i, beta in enumerate(used):
            active = [d for d in range(self.D) if beta[d] > 0]
            idx[i, 0] = len(active)
            idx[i, 1] = off
            for a, d in enumerate(active):
                idx[i, 2 + a] = d
                idx[i, 2 + MAX_ACTIVE + a] = beta[d]
            # node order of the kernel: active dims in increasing dimension order, last one fastest == itertools.product
            v = self.values[beta]
            vals.append(v)
            off += v.shape[0]
        dev = self.device
        return (torch.from_numpy(idx).to(dev), torch.tensor([float(coefs[b]) for b in used], dtype=torch.float64, device=dev),
                torch.from_numpy(np.concatenate(vals)).to(dev), len(used))

    def predict(self, t, index_set=None):
        """t: [D][n] CUDA tensor of normalised coordinates -> [n_out][n] predictions."""
        import torch
        if index_set is None:
            if self._tables is None:
                self._tables = self._build_tables(self.index_set)
            tables = self._tables
        else:
            tables = self._build_tables(index_set)
        idx, coef, vals, nb = tables
        t = t.to(device=self.device, dtype=torch.float64).contiguous()
        n = t.shape[1]
        out = torch.empty((len(self.qoi), n), dtype=torch.float64, device=self.device)
        p = lambda x: C.c_void_p(x.data_ptr())                                                             # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pem_sparse_predict_f64_dev(
                n, self.D, nb, p(idx), p(coef), p(vals), len(self.qoi), p(t), t.stride(0), p(out), out.stride(0),
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def grid_values(self, t, index_set):
        """[len(index_set)][n_out][n]: the interpolant of every grid of `index_set` at the points t ([D][n] CUDA tensor), one
        launch (`pem_sparse_grid_values_f64_dev`).  A prediction with combination coefficients c is c @ grid_values."""
        import torch
        idx, coef, vals, nb = self._build_tables(index_set, unit_coefficients=True)
        t = t.to(device=self.device, dtype=torch.float64).contiguous()
        n = t.shape[1]
        out = torch.empty((nb, len(self.qoi), n), dtype=torch.float64, device=self.device)
        p = lambda x: C.c_void_p(x.data_ptr())                                                             # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pem_sparse_grid_values_f64_dev(
                n, self.D, nb, p(idx), p(coef), p(vals), len(self.qoi), p(t), t.stride(0), p(out), out.stride(1),
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    # ---- adaptive refinement -------------------------------------------------------------------------------------------
    def refine(self, max_iter: int = 10, num_refine: int = 1000, seed: int = 0, max_tol: float = 0.0):
        """Activate, `max_iter` times, the candidate with the largest error indicator.  Returns the training history
        [(beta, indicator, model_evals)]."""
        import torch
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        history = []
        for _ in range(max_iter):
            if not self.candidates:
                break
            t = torch.rand((self.D, num_refine), dtype=torch.float64, device=self.device, generator=g) * 2 - 1
            cands = list(self.candidates)
            self._ensure_values(*cands)
            # A prediction is linear in the combination coefficients: ONE launch gives every grid's interpolant at the
            # points (active grids and candidates alike), and the current surrogate plus every trial index set is a row
            # of a small matrix product -- instead of one table upload and one launch per candidate.
            every = self.index_set + cands
            col = {b: i for i, b in enumerate(every)}
            cmat = np.zeros((1 + len(cands), len(every)))
            for b, c in self.combination_coefficients(self.index_set).items():
                cmat[0, col[b]] = c
            for r, cand in enumerate(cands):
                cmat[1 + r] = cmat[0]
                for b, c in self.combination_delta(self.index_set, cand).items():
                    cmat[1 + r, col[b]] += c
            gv = self.grid_values(t, every)                                                     # [B][n_out][n]
            f = (torch.from_numpy(cmat).to(self.device) @ gv.reshape(len(every), -1)).reshape(1 + len(cands), len(self.qoi), -1)
            base = f[0]
            scale = (base.max(dim=1).values - base.min(dim=1).values).clamp_min(1e-12)
            errs = ((f[1:] - base).abs().mean(dim=2) / scale).max(dim=1).values.cpu().numpy()
            k = int(np.argmax(errs))                                                            # first of equal maxima, as before
            best, best_err = cands[k], float(errs[k])
            self._activate(best)
            history.append((best, best_err, self.model_evals))
            if best_err < max_tol:
                break
        return history
