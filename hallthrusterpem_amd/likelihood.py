"""Gaussian log-likelihood of measured ion current density given plume-model profiles, marginalised over nuisance
samples -- the `jion` branch of `spt100_log_likelihood` (scripts/pem_v0/mcmc.py:57-106).

Layout as in the reference: profiles of shape (..., M, Ne, 91) -- M nuisance draws for each of Ne experimental
conditions -- give `log p(data | theta) = logsumexp_M( sum_{e, a} -0.5 ((y_ea - J(|alpha_ea|)) / std_ea)^2 )`
(the reference also drops the constant terms, mcmc.py:66-69).  The per-sample sums run in one HIP pass over the
profiles (csrc/pem_likelihood.hip); the sum over conditions and the log-sum-exp over M are O(n) torch reductions.
The reference scripts are stale and untested: parity unpinned.
"""
import ctypes as C

import numpy as np

from . import _lib

GRID_STEP = (np.pi / 2) / 90.0


class JionLikelihood:
    def __init__(self, alpha, y, std, device=None):
        """alpha, y, std: (Ne, Na) measurement angles (rad, |alpha| <= pi/2; the plume is mirror-symmetric),
        current densities and standard deviations."""
        import torch
        alpha = np.abs(np.atleast_2d(np.asarray(alpha, dtype=np.float64)))
        if alpha.max() > np.pi / 2 + 1e-12:
            raise ValueError('measurement angles beyond 90 degrees are outside the model sweep (plume.py:53)')
        pos = np.minimum(alpha / GRID_STEP, 90.0)
        k = np.minimum(np.floor(pos).astype(np.int32), 89)
        w = pos - k
        self.n_cond, self.n_ang = alpha.shape
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        f = lambda a: torch.as_tensor(np.array(np.broadcast_to(a, alpha.shape)), device=dev)   # noqa: E731
        self.kidx, self.weight = f(k), f(w)
        self.y = f(np.asarray(y, dtype=np.float64))
        self.inv_std = f(1.0 / np.asarray(std, dtype=np.float64))
        self.device = dev

    def per_sample(self, j_ion):
        """j_ion: (..., 91) CUDA tensor whose flattened sample index i belongs to condition i mod Ne -> (...,) sums."""
        import torch
        flat = j_ion.double().contiguous().reshape(-1, _lib.NANGLE)
        out = torch.empty(flat.shape[0], dtype=torch.float64, device=flat.device)
        p = lambda t: C.c_void_p(t.data_ptr())                                                              # noqa: E731
        with torch.cuda.device(flat.device):
            _lib.check(_lib.load().pem_jion_loglik_f64_dev(
                flat.shape[0], self.n_cond, self.n_ang, p(self.kidx), p(self.weight), p(self.y), p(self.inv_std),
                p(flat), p(out), C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)))
        return out.reshape(j_ion.shape[:-1])

    def log_likelihood(self, j_ion):
        """j_ion: (..., M, Ne, 91) -> (...,) marginal log-likelihood (log-sum-exp over the M nuisance draws)."""
        import torch
        ll = self.per_sample(j_ion).sum(dim=-1)             # (..., M): all conditions of one draw
        return torch.logsumexp(ll, dim=-1)
