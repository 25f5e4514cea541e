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

Round 4 -- what the reference really trains (scripts/pem_v0/pem_v0_SPT-100.yml:273-280, scripts/gen_data.py:261-294,
scripts/fit_surr.py:101-133): a FIELD output (`j_ion`, 91 angles) enters the surrogate as the r latent coefficients of its SVD
compression (log10 norm, reconstruction_tol 0.01), next to the scalar outputs.  `qoi` may therefore name 'j_ion': a compression
set is evaluated (or a fitted `compression.SVDCompression` handed in), the true model at the grid nodes is evaluated and
compressed in one fused launch (`CoupledBatch.run_latent`), the surrogate interpolates 3 + r outputs, and `predict_fields` gives
the scalars and the reconstructed profile from ONE launch (`pem_sparse_predict_field_f64_dev`).  The tables the kernels read are
device-resident and grow by appending (they were rebuilt and uploaded on every refinement step), and a multi-index may have
up to five active dimensions of level four (three and three before).
"""
import ctypes as C
import itertools

import numpy as np

from . import _lib, sampling
from .batch import CoupledBatch
from .models.coupled import COUPLED_INPUTS

MAX_ACTIVE = 5      # include/pem_hip.h PEM_SURR_MAX_ACTIVE
MAX_LEVEL = 4       # include/pem_hip.h PEM_SURR_MAX_LEVEL
FIELDS = {'j_ion': _lib.NANGLE}     # field outputs the surrogate can carry as SVD latents: name -> degrees of freedom


def nodes(level: int) -> np.ndarray:
    if level == 0:
        return np.zeros(1)
    m = 2 ** level + 1
    x = -np.cos(np.pi * np.arange(m) / (m - 1))
    x[(m - 1) // 2] = 0.0                      # exact centre node (cos(pi/2) is 6e-17 in floating point)
    return x


class SparseGridSurrogate:
    def __init__(self, varied, fixed: dict | None = None, priors=None, qoi=('V_cc', 'div_angle', 'T_c'), device=None,
                 compression=None, num_compress: int = 500, compress_seed: int = 0, max_active: int = MAX_ACTIVE, max_level: int = MAX_LEVEL):
        """qoi: scalar outputs (V_cc, div_angle, T_c) and at most one field ('j_ion').  compression: a fitted
        `compression.SVDCompression` of the field (as `process_compression` leaves it on the system's variable); None: one is
        fitted here on `num_compress` true-model evaluations at uniform random points of the varied inputs (gen_data.py:73-76
        default: 500) with log10 norm and reconstruction_tol 0.01 (yml:273-280).  max_active / max_level: what the refinement may
        activate (<= MAX_ACTIVE, MAX_LEVEL)."""
        import torch
        self.priors = dict(sampling.PEM_V0_PRIORS if priors is None else priors)
        self.varied = tuple(varied)
        self.fixed = dict(fixed or {})
        missing = [k for k in COUPLED_INPUTS if k not in self.varied and k not in self.fixed]
        if missing:
            raise ValueError(f'inputs neither varied nor fixed: {missing}')
        self.qoi = tuple(qoi)
        fields = [k for k in self.qoi if k in FIELDS]
        if len(fields) > 1 or any(k not in FIELDS and k not in ('V_cc', 'div_angle', 'T_c') for k in self.qoi):
            raise ValueError(f'qoi: scalars V_cc / div_angle / T_c and at most one field of {sorted(FIELDS)}; got {self.qoi}')
        if not (1 <= max_active <= MAX_ACTIVE and 0 <= max_level <= MAX_LEVEL):
            raise ValueError(f'max_active <= {MAX_ACTIVE}, max_level <= {MAX_LEVEL}')
        self.max_active, self.max_level = int(max_active), int(max_level)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.D = len(self.varied)
        self.index_set = []            # activated multi-indices (downward closed)
        self.candidates = []
        self.values = {}               # beta -> [prod(m)][n_out] numpy array of true-model outputs at the grid nodes
        self.model_evals = 0
        self._tables = None
        self._rows = {}                # beta -> (first row, rows) of the device-resident value table
        self._dev_values = None        # [capacity][n_out] CUDA tensor, rows appended as grids are evaluated
        self._dev_rows = 0
        self.field = fields[0] if fields else None
        self.scalars = tuple(k for k in self.qoi if k not in FIELDS)
        self.compression = None
        if self.field:
            self.compression = compression if compression is not None else self._fit_compression(num_compress, compress_seed)
            if self.compression.basis is None or self.compression.basis.shape[0] != FIELDS[self.field]:
                raise ValueError('the compression map is not one of the 91-point profile')
        # outputs the kernels interpolate: the scalars, then the field's latent coefficients
        self.out_names = self.scalars + tuple(f'{self.field}_latent{i}' for i in range(self.compression.rank)) if self.field else self.scalars
        self.n_out = len(self.out_names)
        if self.n_out > 16:
            raise ValueError(f'{self.n_out} outputs (scalars + latents): the predict kernel keeps at most 16')
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
        full = {k: np.full(n, float(self.fixed[k])) for k in self.fixed}
        full.update(x)
        y = self._true_outputs(full, n).cpu().numpy()                           # (the copy synchronises)
        off = 0
        for b, g in zip(new, grids):
            self.values[b] = y[off:off + g.shape[1]]
            off += g.shape[1]
        self._append_rows(new, y)
        self.model_evals += n

    def _true_outputs(self, inputs: dict, n: int):
        """[n][n_out] CUDA tensor: the true model's scalars and -- fused, the profile is never stored -- the field's latents"""
        import torch
        fused = self.field and self.compression.rank <= _lib.FUSED_LATENT_MAX_RANK
        batch = CoupledBatch(n, device=self.device, profile=bool(self.field) and not fused)
        batch.set_inputs(inputs)
        if fused:
            lat = batch.run_latent(self.compression)                            # [n][rank]; V_cc / div_angle / T_c written as by run()
        else:
            batch.run()
            if self.field:                                                      # more latents than the fused launch keeps in registers:
                lat = self.compression.compress(batch.j_ion)                    # profile stored once, pem_svd_compress_f64_dev over it
        o = batch.outputs()
        cols = [o[k] for k in self.scalars]
        y = torch.stack(cols, dim=1) if cols else torch.empty((n, 0), dtype=torch.float64, device=self.device)
        return torch.cat([y, lat], dim=1) if self.field else y

    def _fit_compression(self, num: int, seed: int):
        """The SVD map of the field from `num` true-model evaluations at uniform random points of the varied inputs (the
        'compression' data set of gen_data.py:218-294, drawn here over the surrogate's own input box)."""
        import torch
        from .compression import SVDCompression
        rng = np.random.default_rng(seed)
        t = rng.uniform(-1.0, 1.0, (self.D, int(num)))
        full = {k: np.full(int(num), float(self.fixed[k])) for k in self.fixed}
        full.update(self.to_physical(t))
        batch = CoupledBatch(int(num), device=self.device, profile=True)
        batch.set_inputs(full)
        batch.run()
        torch.cuda.synchronize()
        keep = ~batch.invalid.bool()                                            # gen_data.py:277: NaN / invalid samples are dropped
        self.model_evals += int(num)
        return SVDCompression(norm='log10', reconstruction_tol=0.01).fit(batch.j_ion[keep])

    def _append_rows(self, betas, y):
        """the new grids' node values behind the device-resident table (grown by doubling): nothing is re-uploaded later"""
        import torch
        rows = y.shape[0]
        need = self._dev_rows + rows
        if self._dev_values is None or need > self._dev_values.shape[0]:
            cap = max(1024, 2 * need)
            grown = torch.empty((cap, self.n_out), dtype=torch.float64, device=self.device)
            if self._dev_values is not None:
                grown[:self._dev_rows] = self._dev_values[:self._dev_rows]
            self._dev_values = grown
        self._dev_values[self._dev_rows:need] = torch.from_numpy(np.ascontiguousarray(y)).to(self.device)
        off = self._dev_rows
        for b in betas:
            m = self.values[b].shape[0]
            self._rows[b] = (off, m)
            off += m
        self._dev_rows = need

    def rebuild_device_tables(self):
        """the device-resident value table from `values` (after `values` was restored from a file)"""
        self._rows, self._dev_values, self._dev_rows, self._tables = {}, None, 0, None
        for b, v in self.values.items():
            self._append_rows([b], np.asarray(v))

    # ---- index-set bookkeeping (shape of monte_carlo.py:714-747) ------------------------------------------------------
    def _admissible(self, beta):
        na, lv = sum(1 for l in beta if l > 0), max(beta)
        if na > self.max_active or lv > self.max_level:
            return False
        # the predict kernel keeps the outer dimensions' bases of the WHOLE table's largest grid shape in LDS:
        # ((most active dimensions - 1) x (2^highest level + 1) + D) doubles per thread of 256 within 160 KB
        na_t = max([na] + [sum(1 for l in b if l > 0) for b in self.values])
        lv_t = max([lv] + [max(b) for b in self.values])
        if (max(na_t - 1, 0) * ((1 << lv_t) + 1 if lv_t else 1) + self.D) * 256 * 8 > 160 * 1024:
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
        `unit_coefficients` -- every grid with coefficient 1 (for `grid_values`).  Only the small index / coefficient arrays are
        built here; the node values are the resident table's rows (`_append_rows`).  Returns (index, coefficients, values, grids,
        most active dimensions, highest level)."""
        import torch
        coefs = {b: 1 for b in index_set} if unit_coefficients else self.combination_coefficients(index_set)
        used = [b for b in index_set if coefs[b] != 0]
        idx = np.zeros((len(used), 2 + 2 * MAX_ACTIVE), dtype=np.int32)
        na_max, lv_max = 0, 0
        for i, beta in enumerate(used):
            active = [d for d in range(self.D) if beta[d] > 0]
            idx[i, 0] = len(active)
            idx[i, 1] = self._rows[beta][0]
            for a, d in enumerate(active):
                idx[i, 2 + a] = d
                idx[i, 2 + MAX_ACTIVE + a] = beta[d]
            # node order of the kernel: active dims in increasing dimension order, last one fastest == itertools.product
            na_max, lv_max = max(na_max, len(active)), max(lv_max, max(beta) if beta else 0)
        dev = self.device
        return (torch.from_numpy(idx).to(dev), torch.tensor([float(coefs[b]) for b in used], dtype=torch.float64, device=dev),
                self._dev_values, len(used), na_max, lv_max)

    def _tables_for(self, index_set):
        if index_set is None:
            if self._tables is None:
                self._tables = self._build_tables(self.index_set)
            return self._tables
        return self._build_tables(index_set)

    def predict(self, t, index_set=None):
        """t: [D][n] CUDA tensor of normalised coordinates -> [n_out][n] predictions: the scalars of `qoi`, then the field's
        latent coefficients (`out_names`)."""
        import torch
        idx, coef, vals, nb, na, lv = self._tables_for(index_set)
        t = t.to(device=self.device, dtype=torch.float64).contiguous()
        n = t.shape[1]
        out = torch.empty((self.n_out, n), dtype=torch.float64, device=self.device)
        p = lambda x: C.c_void_p(x.data_ptr())                                                             # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pem_sparse_predict_f64_dev(
                n, self.D, nb, p(idx), p(coef), p(vals), self.n_out, p(t), t.stride(0), p(out), out.stride(0), na, lv,
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def predict_fields(self, t, index_set=None):
        """t: [D][n] -> {scalar: (n,), field: (n, 91)}: what amisc's System.predict returns for a compressed variable -- the
        reconstructed field -- from ONE launch (`pem_sparse_predict_field_f64_dev`: the interpolated latents are turned into
        10^(latent @ basis^T) before they leave the chip)."""
        import torch
        if not self.field:
            y = self.predict(t, index_set)
            return {k: y[i] for i, k in enumerate(self.scalars)}
        idx, coef, vals, nb, na, lv = self._tables_for(index_set)
        t = t.to(device=self.device, dtype=torch.float64).contiguous()
        n = t.shape[1]
        c = self.compression
        out = torch.empty((self.n_out, n), dtype=torch.float64, device=self.device)
        field = torch.empty((n, FIELDS[self.field]), dtype=torch.float64, device=self.device)
        basis = c.basis.contiguous()
        p = lambda x: C.c_void_p(x.data_ptr())                                                             # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pem_sparse_predict_field_f64_dev(
                n, self.D, nb, p(idx), p(coef), p(vals), self.n_out, p(t), t.stride(0), p(out), out.stride(0), na, lv,
                len(self.scalars), c.rank, FIELDS[self.field], c.norm, c.scale, p(basis), p(field),
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        res = {k: out[i] for i, k in enumerate(self.scalars)}
        res[self.field] = field
        res[f'{self.field}_latent'] = out[len(self.scalars):].T
        return res

    def grid_values(self, t, index_set):
        """[len(index_set)][n_out][n]: the interpolant of every grid of `index_set` at the points t ([D][n] CUDA tensor), one
        launch (`pem_sparse_grid_values_f64_dev`).  A prediction with combination coefficients c is c @ grid_values."""
        import torch
        idx, coef, vals, nb, na, lv = self._build_tables(index_set, unit_coefficients=True)
        t = t.to(device=self.device, dtype=torch.float64).contiguous()
        n = t.shape[1]
        out = torch.empty((nb, self.n_out, n), dtype=torch.float64, device=self.device)
        p = lambda x: C.c_void_p(x.data_ptr())                                                             # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().pem_sparse_grid_values_f64_dev(
                n, self.D, nb, p(idx), p(coef), p(vals), self.n_out, p(t), t.stride(0), p(out), out.stride(1), na, lv,
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
            f = (torch.from_numpy(cmat).to(self.device) @ gv.reshape(len(every), -1)).reshape(1 + len(cands), self.n_out, -1)
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
