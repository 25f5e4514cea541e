"""On-device input designs for the PEM-v0 sampling loops (Monte-Carlo, Latin hypercube, Saltelli blocks).

Mirrors the CALL PATTERN of the reference drivers -- `system.sample_inputs(N, use_pdf=['calibration', 'nuisance'])`
in scripts/gen_data.py:238, `(Ns, Nx)` sampling in scripts/pem_v0/monte_carlo.py and sobol.py:46-66 -- whose
implementation is amisc/uqtils (third-party, absent: parity unpinned).  The prior table below is DATA taken from
scripts/pem_v0/pem_v0_SPT-100.yml: calibration/nuisance variables are drawn from their
distribution, operating variables uniformly over their domain in normalised (log10 where declared) space.

Designs are counter-based (csrc/pem_sampler.hip): sample i of a design is the same numbers however the design is
split over batches or GPUs.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .models.coupled import COUPLED_INPUTS

UNIFORM, LOGUNIFORM, NORMAL = 0, 1, 2


@dataclass(frozen=True)
class Prior:
    kind: int       # UNIFORM: U(a, b);  LOGUNIFORM: 10^U(a, b) with a, b = log10 bounds;  NORMAL: N(a, b)
    a: float
    b: float
    source: str     # yml line(s) the entry restates


def _logu(lo, hi, src):
    return Prior(LOGUNIFORM, float(np.log10(lo)), float(np.log10(hi)), src)


# the 15 inputs of the coupled cathode -> thruster(test double) -> plume graph
PEM_V0_PRIORS = {
    'P_b': _logu(1e-8, 1e-4, 'yml:9-17 operating, domain (1e-8, 1e-4), norm log10'),
    'V_a': Prior(UNIFORM, 200.0, 400.0, 'yml:18-24 operating, domain (200, 400)'),
    'T_e': Prior(UNIFORM, 1.0, 5.0, 'yml:25-31 U(1, 5)'),
    'V_vac': Prior(UNIFORM, 0.0, 60.0, 'yml:32-38 U(0, 60)'),
    'Pstar': Prior(UNIFORM, 10e-6, 100e-6, 'yml:39-46 Uniform(10e-6, 100e-6)'),
    'P_T': Prior(UNIFORM, 10e-6, 100e-6, 'yml:47-54 Uniform(10e-6, 100e-6)'),
    'mdot_a': Prior(UNIFORM, 2e-6, 7e-6, 'yml:113-121 operating, domain (2e-6, 7e-6)'),
    'a_1': _logu(0.00316, 0.1, 'yml:138-144 LogUniform(0.00316, 0.1)'),
    'c0': Prior(UNIFORM, 0.0, 1.0, 'yml:221-226 U(0, 1)'),
    'c1': Prior(UNIFORM, 0.1, 0.9, 'yml:227-232 U(0.1, 0.9)'),
    'c2': Prior(UNIFORM, -15.0, 15.0, 'yml:233-239 U(-15, 15)'),
    'c3': Prior(UNIFORM, 0.2, 1.570796, 'yml:240-246 U(0.2, 1.570796)'),
    'c4': _logu(1e18, 1e22, 'yml:247-254 LogUniform(1e18, 1e22)'),
    'c5': _logu(1e14, 1e18, 'yml:255-262 LogUniform(1e14, 1e18)'),
    'sigma_cex': Prior(UNIFORM, 51e-20, 58e-20, 'yml:263-270 Uniform(51e-20, 58e-20)'),
}
assert tuple(PEM_V0_PRIORS) == COUPLED_INPUTS


class Design:
    """A counter-based design over `names` with the given priors; `fill()` writes any slice of it into HBM."""

    def __init__(self, priors=None, names=COUPLED_INPUTS, seed: int = 0, stream: int = 0):
        priors = PEM_V0_PRIORS if priors is None else priors
        self.names = tuple(names)
        self.seed, self.stream = int(seed), int(stream)
        self.kind = np.ascontiguousarray([priors[k].kind for k in self.names], dtype=np.int32)
        self.a = np.ascontiguousarray([priors[k].a for k in self.names], dtype=np.float64)
        self.b = np.ascontiguousarray([priors[k].b for k in self.names], dtype=np.float64)

    @property
    def ndim(self) -> int:
        return len(self.names)

    def fill(self, out, first_index: int = 0, method: str = 'mc', n_total: int | None = None, swap_dim: int = -1,
             stream=None):
        """Write samples first_index .. first_index + n - 1 into `out` ([ndim][n] float64 CUDA tensor, rows = names).

        method 'mc': independent draws;  'lhs': Latin hypercube over `n_total` strata per dimension.
        swap_dim (mc only): Saltelli blocks -- -1 matrix A, -2 matrix B, d >= 0 A with column d from B."""
        import torch
        lib = _lib.load()
        assert out.is_cuda and out.dtype == torch.float64 and out.dim() == 2 and out.shape[0] == self.ndim
        assert out.stride(1) == 1
        n, ld = out.shape[1], out.stride(0)
        s = torch.cuda.current_stream(out.device) if stream is None else stream
        ptr = lambda arr: C.c_void_p(arr.ctypes.data)                                           # noqa: E731
        with torch.cuda.device(out.device):
            if method == 'mc':
                rc = lib.pem_sample_f64_dev(n, first_index, self.seed, self.stream, self.ndim, ptr(self.kind), ptr(self.a),
                                            ptr(self.b), swap_dim, C.c_void_p(out.data_ptr()), ld, C.c_void_p(s.cuda_stream))
            elif method == 'lhs':
                if swap_dim != -1:
                    raise ValueError('Saltelli blocks are built from Monte-Carlo matrices')
                rc = lib.pem_sample_lhs_f64_dev(n, first_index, int(n_total if n_total is not None else n), self.seed,
                                                self.stream, self.ndim, ptr(self.kind), ptr(self.a), ptr(self.b),
                                                C.c_void_p(out.data_ptr()), ld, C.c_void_p(s.cuda_stream))
            else:
                raise ValueError(f"unknown sampling method '{method}'")
        _lib.check(rc)
        return out

    def fill_tiled(self, out, n: int, first_index: int = 0, swap_dim: int = -1, stream=None):
        """The numbers of `fill(method='mc')` written tile-interleaved: `out` is a [ceil(n / 64)][ndim][64] float64 CUDA
        tensor (`CoupledBatch(layout='tile').inputs`), sample i at out[i // 64, :, i % 64]."""
        import torch
        assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous()
        assert out.dim() == 3 and out.shape[1] == self.ndim and out.shape[2] == 64 and out.shape[0] * 64 >= n
        s = torch.cuda.current_stream(out.device) if stream is None else stream
        ptr = lambda arr: C.c_void_p(arr.ctypes.data)                                           # noqa: E731
        with torch.cuda.device(out.device):
            rc = _lib.load().pem_sample_tiled_f64_dev(int(n), first_index, self.seed, self.stream, self.ndim, ptr(self.kind),
                                                      ptr(self.a), ptr(self.b), swap_dim, C.c_void_p(out.data_ptr()),
                                                      C.c_void_p(s.cuda_stream))
        _lib.check(rc)
        return out

    def sample(self, n: int, device=None, **kw):
        import torch
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        return self.fill(torch.empty((self.ndim, n), dtype=torch.float64, device=dev), **kw)

    def as_dict(self, tensor):
        return {k: tensor[i] for i, k in enumerate(self.names)}
