"""SVD compression of field QoIs (`j_ion`, `u_ion`) -- the `compression: {method: svd, reconstruction_tol: 0.01}`
blocks of scripts/pem_v0/pem_v0_SPT-100.yml:207-214,273-280 and `process_compression` of scripts/gen_data.py:261-294.

The reference delegates this to amisc (`var.compression.compute_map / compress / reconstruct`; third-party, absent:
parity unpinned).  Stated here instead:
  * `fit`: thin SVD of the normalised data matrix A (num_samples x dof) -- no centring -- with torch.linalg.svd
    (factorisation is plumbing, done once per data set); the basis is the first `rank` right singular vectors,
    `rank` = the smallest r whose relative Frobenius reconstruction error sqrt(sum_{i>r} s_i^2 / sum s_i^2) is
    <= `reconstruction_tol` (or given explicitly).
  * `compress` / `reconstruct`: the per-sample hot operations, tall-skinny fp64 GEMMs on the MFMA units with the
    variable's norm fused into the load / store (csrc/pem_svd.hip).
"""
import ctypes as C

from . import _lib

NORM_NONE, NORM_LOG10, NORM_LINEAR = 0, 1, 2


class SVDCompression:
    def __init__(self, norm: str = 'none', scale: float = 1.0, reconstruction_tol: float = 0.01, rank: int | None = None):
        self.norm = {'none': NORM_NONE, 'log10': NORM_LOG10, 'linear': NORM_LINEAR}[norm]
        self.scale = float(scale)
        self.reconstruction_tol = float(reconstruction_tol)
        self.rank = rank
        self.basis = None            # [dof][rank] CUDA float64
        self.singular_values = None

    # -- the variable's norm, as amisc applies it before compressing (yml `norm:`) --
    def normalize(self, x):
        import torch
        return torch.log10(x) if self.norm == NORM_LOG10 else (x * self.scale if self.norm == NORM_LINEAR else x)

    def fit(self, data, normalized: bool = False):
        """data: [num_samples][dof] CUDA tensor of RAW field values (the norm is applied here), or of already
        normalised ones (`normalized=True`, the data matrix of gen_data.py:287-289)."""
        import torch
        a = data.double() if normalized else self.normalize(data.double())
        _, s, vh = torch.linalg.svd(a, full_matrices=False)
        energy = torch.cumsum(s * s, 0) / torch.sum(s * s)
        err = torch.sqrt(torch.clamp(1.0 - energy, min=0.0))          # relative Frobenius error keeping r = i+1 vectors
        if self.rank is None:
            ok = torch.nonzero(err <= self.reconstruction_tol)
            self.rank = int(ok[0].item()) + 1 if ok.numel() else int(s.numel())
        if self.rank > 16:
            raise ValueError(f'rank {self.rank} > 16 is not supported by the MFMA kernels; raise reconstruction_tol')
        self.basis = vh[: self.rank].T.contiguous()
        self.singular_values = s
        self.relative_error = float(err[self.rank - 1])
        return self

    def _call(self, fn, n, dof, src, dst):
        import torch
        with torch.cuda.device(src.device):
            rc = fn(n, dof, self.rank, self.norm, self.scale, C.c_void_p(src.data_ptr()), C.c_void_p(self.basis.data_ptr()),
                    C.c_void_p(dst.data_ptr()), C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream))
        _lib.check(rc)

    def compress(self, field):
        """[..., dof] raw field values -> [..., rank] latent coefficients."""
        import torch
        dof = self.basis.shape[0]
        flat = field.double().contiguous().reshape(-1, dof)
        out = torch.empty((flat.shape[0], self.rank), dtype=torch.float64, device=flat.device)
        self._call(_lib.load().pem_svd_compress_f64_dev, flat.shape[0], dof, flat, out)
        return out.reshape(field.shape[:-1] + (self.rank,))

    def reconstruct(self, latent):
        """[..., rank] latent coefficients -> [..., dof] raw field values."""
        import torch
        dof = self.basis.shape[0]
        flat = latent.double().contiguous().reshape(-1, self.rank)
        out = torch.empty((flat.shape[0], dof), dtype=torch.float64, device=flat.device)
        self._call(_lib.load().pem_svd_reconstruct_f64_dev, flat.shape[0], dof, flat, out)
        return out.reshape(latent.shape[:-1] + (dof,))
