"""`PemV0System`: the call signature of the `amisc.System` the reference's drivers hold, for the coupled PEM-v0 graph.

scripts/gen_data.py and scripts/fit_surr.py never touch the models directly: they call `system.sample_inputs(...)`,
`system.predict(...)`, `system.inputs()/outputs()`, `var.normalize(...)`, `var.compression.compute_map(...)`,
`system.fit(...)`, `system.get_allocation()` (SURVEY.md Appendix C lists every call site).  amisc is third-party and
absent, so this is NOT its implementation: it is the same method names and argument meaning, as far as they can be read
off the call sites, bound to the device path of this package --

    sample_inputs -> sampling.Design (counter-based, on device)          gen_data.py:238, fit_surr.py (via fit)
    predict(use_model='best') -> pem_coupled_f64[_dev]                   gen_data.py:239-240
    predict() after fit()     -> surrogate.SparseGridSurrogate.predict   fit_surr.py:101-116
    Variable.normalize / .compression -> compression.SVDCompression      gen_data.py:246-248, 279-291

so that the drivers of `hallthrusterpem_amd.drivers` (`generate_data`, `process_compression`) read like the
reference's.  The variable table (category, distribution, norm) is data from pem_v0_SPT-100.yml.
The thruster component is the analytic TEST DOUBLE (tests/sim_hallthruster.jl:35-48), never the Julia solver.
Parity with amisc: UNPINNED.
"""
import logging
import pickle
from pathlib import Path

import numpy as np

from .compression import SVDCompression
from .models.coupled import COUPLED_INPUTS, pem_v0_coupled
from .sampling import LOGUNIFORM, NORMAL, PEM_V0_PRIORS, Design, Prior

COORDS_STR_ID = '_coords'        # amisc.typing.COORDS_STR_ID as used at plume.py:157, gen_data.py:281

CATEGORY = {'P_b': 'operating', 'V_a': 'operating', 'mdot_a': 'operating', 'sigma_cex': 'nuisance'}     # else calibration
NORM = {'P_b': 'log10', 'Pstar': ('linear', 1e6), 'P_T': ('linear', 1e6), 'mdot_a': ('linear', 1e6), 'a_1': 'log10',
        'c4': 'log10', 'c5': 'log10', 'sigma_cex': ('linear', 1e20)}                                    # yml `norm:` entries
OUTPUTS = ('V_cc', 'I_B0', 'T', 'j_ion', 'div_angle', 'T_c')


class Variable:
    """Name-like (hashes and compares as its name, so `var in outputs_dict` works as at gen_data.py:246)."""

    def __init__(self, name, category=None, prior: Prior | None = None, norm=None, compression=None):
        self.name, self.category, self.prior, self.norm, self.compression = name, category, prior, norm, compression

    def __str__(self):
        return self.name

    __repr__ = __str__

    def __hash__(self):
        return hash(self.name)

    def __eq__(self, other):
        return self.name == (other.name if isinstance(other, Variable) else other)

    def normalize(self, values, denorm: bool = False):
        """The yml `norm:` of the variable (log10, linear(s): multiply by s); numpy arrays and torch tensors."""
        if self.norm is None:
            return values
        is_np = not hasattr(values, 'is_cuda')
        if self.norm == 'log10':
            if denorm:
                return np.power(10.0, values) if is_np else 10.0 ** values
            if is_np:
                with np.errstate(divide='ignore', invalid='ignore'):
                    return np.log10(values)
            return values.log10()
        scale = self.norm[1]
        return values / scale if denorm else values * scale


class VariableList(list):
    def __getitem__(self, key):
        if isinstance(key, (str, Variable)):
            return next(v for v in self if v == key)
        return super().__getitem__(key)

    def __contains__(self, key):
        return any(v == key for v in self)


def to_model_dataset(samples: dict, variables):
    """Normalised sample dict -> (model inputs in physical units, coords dict): gen_data.py:242.  No input of the
    coupled graph is a field quantity, so the coords dict is empty."""
    out = {}
    for k, v in samples.items():
        var = variables[k] if k in variables else None
        out[str(k)] = var.normalize(v, denorm=True) if var is not None else v
    return out, {}


class Component:
    """What scripts/fit_surr.py:119-160 touches of an amisc component: `.name`, `.model_fidelity` (settable: the single-fidelity
    run of `train_surrogate` empties it), `.model_costs` ({alpha: cost of one evaluation at that fidelity}).  The three PEM-v0
    sub-models are evaluated by ONE coupled launch here, so they share one fidelity -- () -- and the cost unit is one coupled
    evaluation split evenly; the reference's only multi-fidelity component is the Julia thruster (out of scope)."""

    def __init__(self, name: str, share: float):
        self.name, self.model_fidelity, self.model_costs = name, (), {'()': float(share)}

    def __str__(self):
        return self.name


class PemV0System:
    COMPONENT_NAMES = ('Cathode', 'Thruster (analytic test double)', 'Plume')

    def __init__(self, root_dir=None, name: str = 'PEM_v0_SPT-100', priors=None, seed: int = 0, sweep_radius: float = 1.0):
        self.name, self.root_dir = name, root_dir
        self.priors = dict(PEM_V0_PRIORS if priors is None else priors)
        self.sweep_radius = float(sweep_radius)
        self._inputs = VariableList(Variable(k, CATEGORY.get(k, 'calibration'), self.priors[k], NORM.get(k))
                                    for k in COUPLED_INPUTS)
        self._outputs = VariableList(Variable(k) for k in OUTPUTS)
        self._outputs['j_ion'].norm = 'log10'                                                    # yml:273-280
        self._outputs['j_ion'].compression = FieldCompression(self._outputs['j_ion'], fields=('j_ion',),
                                                              reconstruction_tol=0.01)
        self.design = Design(priors=self.priors, seed=seed)
        self._drawn = 0                     # samples handed out so far: successive calls continue the same design
        self.surrogate = None
        self.train_history = []
        self.logger = logging.getLogger(name)
        self.components = [Component(nm, 1.0 / len(self.COMPONENT_NAMES)) for nm in self.COMPONENT_NAMES]

    @property
    def root_dir(self):
        return self._root_dir

    @root_dir.setter
    def root_dir(self, value):
        """Setting the root directory creates it, as amisc's `System.root_dir` does: gen_data.py:442-446 lists it right after
        assigning it and fit_surr.py:149 moves a system into a sub-directory that does not exist yet."""
        self._root_dir = Path(value) if value is not None else None
        if self._root_dir is not None:
            self._root_dir.mkdir(parents=True, exist_ok=True)

    # ------------------------------------------------------------------------------------------------ bookkeeping
    def __getitem__(self, name):
        """system[component name] (fit_surr.py:137: `system[comp.name].model_costs`)"""
        for c in self.components:
            if c.name == str(name):
                return c
        raise KeyError(name)

    def plot_allocation(self, *_, **__):
        """fit_surr.py:117 calls it between fit and get_allocation: there is one component-fidelity pair, nothing to draw"""
        return None

    def inputs(self):
        return self._inputs

    def outputs(self):
        return self._outputs

    def set_logger(self, stdout: bool = True, **_):
        if stdout and not self.logger.handlers:
            self.logger.addHandler(logging.StreamHandler())
            self.logger.setLevel(logging.INFO)

    def clear(self):
        self.surrogate, self.train_history = None, []

    # ---------------------------------------------------------------------------------------------------- sampling
    def sample_inputs(self, size, normalize: bool = True, use_pdf=False, as_tensor: bool = False):
        """`size` samples (int or shape) of every input, dict name -> array of that shape.

        use_pdf: False, True or a list of categories -- variables of those categories are drawn from their
        distribution, the others uniformly over their domain in normalised space (gen_data.py:238).  Every PEM-v0
        distribution is (log-)uniform over its domain except custom NORMAL priors, which fall back to
        mean +- 3 sigma when not drawn from their pdf.  normalize: return the yml-normalised values.
        as_tensor: CUDA tensors (no host copy) instead of numpy arrays."""
        shape = (size,) if isinstance(size, (int, np.integer)) else tuple(size)
        n = int(np.prod(shape))
        cats = ({'operating', 'calibration', 'nuisance'} if use_pdf is True else set() if not use_pdf else set(use_pdf))
        design = self.design
        swap = {k: Prior(0, p.a - 3 * p.b, p.a + 3 * p.b, 'domain of a normal prior') for k, p in self.priors.items()
                if p.kind == NORMAL and self._inputs[k].category not in cats}
        if swap:
            design = Design(priors={**self.priors, **swap}, seed=self.design.seed)
        x = design.sample(n, first_index=self._drawn)
        self._drawn += n
        out = {}
        for i, var in enumerate(self._inputs):
            v = var.normalize(x[i], denorm=False) if normalize else x[i]
            v = v.reshape(shape)
            out[var.name] = v if as_tensor else v.cpu().numpy()
        return out

    # --------------------------------------------------------------------------------------------------- prediction
    def predict(self, x: dict, use_model=None, normalized_inputs: bool = True, targets=None, model_dir=None,
                executor=None, verbose: bool = False):
        """Outputs for the samples `x` (dict name -> array/tensor, any common loop shape).

        use_model='best': the true coupled model, one `pem_coupled_f64` launch (host arrays) / `_dev` (CUDA tensors);
        use_model=None: the trained surrogate (`fit` first).  `model_dir` / `executor` are accepted and unused: the
        models write no files and the batch is one launch (gen_data.py:239-240 passes them)."""
        inputs = to_model_dataset(x, self._inputs)[0] if normalized_inputs else {str(k): v for k, v in x.items()}
        if use_model in ('best', 'true', 'high'):
            out = pem_v0_coupled(inputs, sweep_radius=self.sweep_radius, profile=True, coords=True)
            out = {k: v for k, v in out.items() if k != 'invalid'}
        elif use_model is None:
            if self.surrogate is None:
                raise RuntimeError('no surrogate has been trained: call fit() or predict(use_model="best")')
            out = self._predict_surrogate(inputs)
        else:
            raise ValueError(f"use_model={use_model!r}: the coupled graph has one fidelity ('best') and the surrogate (None)")
        if targets is not None:
            out = {k: v for k, v in out.items() if k in targets or k.endswith(COORDS_STR_ID)}
        return out

    def _predict_surrogate(self, inputs):
        import torch
        s = self.surrogate
        vals = [np.asarray(inputs[k].cpu() if hasattr(inputs[k], 'cpu') else inputs[k], dtype=np.float64) for k in s.varied]
        shape = np.broadcast_shapes(*[v.shape for v in vals])
        t = np.empty((s.D, int(np.prod(shape))))
        for d, k in enumerate(s.varied):
            p = s.priors[k]
            v = np.broadcast_to(vals[d], shape).reshape(-1)
            u = (np.log10(v) if p.kind == LOGUNIFORM else v)
            t[d] = 2.0 * (u - p.a) / (p.b - p.a) - 1.0
        y = s.predict_fields(torch.from_numpy(t))
        out = {k: y[k].cpu().numpy().reshape(shape) for k in s.scalars}
        if s.field:                                          # the compressed field comes back reconstructed, with its coordinates
            f = y[s.field].cpu().numpy()
            out[s.field] = f.reshape(shape + (f.shape[-1],))
            from .models.plume import _coords, angle_grid
            out[f'{s.field}{COORDS_STR_ID}'] = _coords(shape, angle_grid())      # plume.py:153-157: every element the same alpha_rad array
        return out

    # ----------------------------------------------------------------------------------------------------- training
    def fit(self, targets=None, max_iter: int = 20, max_tol: float = 1e-3, num_refine: int = 1000, varied=None,
            fixed: dict | None = None, seed: int = 0, test_set=None, **_):
        """Adaptive sparse-grid training over `varied` (default: every input not in `fixed`), keyword names of
        fit_surr.py:101-116.  Each iteration's (activated index, error indicator, model evaluations[, test error per
        target]) is appended to `train_history`."""
        from .surrogate import SparseGridSurrogate
        fixed = dict(fixed or {})
        varied = tuple(varied) if varied is not None else tuple(k for k in COUPLED_INPUTS if k not in fixed)
        # targets=None: every output the surrogate can carry, as the reference trains all of a system's outputs -- the scalars and
        # j_ion through the latent coefficients of its SVD map (process_compression's, when it has run; else one fitted for the box)
        qoi = tuple(targets) if targets else ('V_cc', 'div_angle', 'T_c', 'j_ion')
        if self.surrogate is None or self.surrogate.varied != varied or self.surrogate.qoi != qoi:
            comp = None
            if 'j_ion' in qoi and 'j_ion' in self._outputs and self._outputs['j_ion'].compression is not None \
                    and self._outputs['j_ion'].compression.svd.basis is not None:
                comp = self._outputs['j_ion'].compression.svd
            self.surrogate = SparseGridSurrogate(varied, fixed=fixed, priors=self.priors, qoi=qoi, compression=comp)
        for _it in range(max_iter):
            hist = self.surrogate.refine(max_iter=1, num_refine=num_refine, seed=seed + len(self.train_history))
            if not hist:
                break
            beta, indicator, evals = hist[0]
            entry = {'added': beta, 'indicator': indicator, 'model_evals': evals}
            if test_set is not None:
                xt, yt = test_set
                pred = self._predict_surrogate(xt)
                # relative L2 error per target (fit_surr.py:121-133 plots exactly this); a log10-normalised field in its norm
                nrm = lambda k, v: np.log10(np.asarray(v, dtype=np.float64)) if k == 'j_ion' else np.asarray(v, dtype=np.float64)   # noqa: E731
                entry['test_error'] = {k: float(np.linalg.norm(nrm(k, pred[k]) - nrm(k, yt[k])) / np.linalg.norm(nrm(k, yt[k])))
                                       for k in qoi if k in yt}
            self.train_history.append(entry)
            if indicator < max_tol:
                break
        return self.train_history

    def get_allocation(self):
        """(cost_alloc, model_cost, overhead_cost, model_evals) as unpacked at fit_surr.py:119: one component-fidelity
        pair here, cost in units of one coupled evaluation, no overhead bookkeeping."""
        evals = np.array([h['model_evals'] for h in self.train_history], dtype=np.float64)
        per_iter = np.diff(evals, prepend=0.0)
        total = float(evals[-1]) if evals.size else 0.0
        # per (component, alpha), as amisc keys them; every evaluation of the coupled graph evaluates all three components once
        cost_alloc = {c.name: {str(c.model_fidelity): total * c.model_costs['()']} for c in self.components}
        model_cost = {c.name: dict(c.model_costs) for c in self.components}
        return cost_alloc, model_cost, 0.0, per_iter

    # ------------------------------------------------------------------------------------------------ persistence
    def save_to_file(self, filename, save_dir=None):
        """State needed to reuse the system (compression maps, trained surrogate), pickled -- the reference writes
        amisc YAML (gen_data.py:294); the file name is kept, the format is this package's."""
        path = Path(save_dir if save_dir is not None else self.root_dir or '.') / filename
        state = {'name': self.name, 'drawn': self._drawn, 'seed': self.design.seed, 'train_history': self.train_history,
                 'compression': {v.name: v.compression.state() for v in self._outputs if v.compression is not None},
                 'surrogate': None if self.surrogate is None else
                 {'varied': self.surrogate.varied, 'fixed': self.surrogate.fixed, 'qoi': self.surrogate.qoi,
                  'index_set': self.surrogate.index_set, 'candidates': self.surrogate.candidates,
                  'values': self.surrogate.values, 'model_evals': self.surrogate.model_evals,
                  'max_active': self.surrogate.max_active, 'max_level': self.surrogate.max_level,
                  'compression': None if self.surrogate.compression is None else
                  {'rank': self.surrogate.compression.rank, 'basis': self.surrogate.compression.basis.cpu().numpy(),
                   'relative_error': getattr(self.surrogate.compression, 'relative_error', None)}}}
        with open(path, 'wb') as fd:
            pickle.dump(state, fd)
        return path

    @classmethod
    def load_from_file(cls, filename, root_dir=None, **kw):
        with open(filename, 'rb') as fd:
            state = pickle.load(fd)
        self = cls(root_dir=root_dir if root_dir is not None else Path(filename).parent, name=state['name'],
                   seed=state['seed'], **kw)
        self._drawn, self.train_history = state['drawn'], state['train_history']
        for k, st in state['compression'].items():
            self._outputs[k].compression.load_state(st)
        if state['surrogate'] is not None:
            from .surrogate import SparseGridSurrogate
            st = state['surrogate']
            comp = None
            if st.get('compression') is not None:
                import torch
                from .compression import SVDCompression
                comp = SVDCompression(norm='log10', reconstruction_tol=0.01, rank=st['compression']['rank'])
                comp.basis = torch.from_numpy(st['compression']['basis']).cuda()
                comp.relative_error = st['compression']['relative_error']
            s = SparseGridSurrogate(st['varied'], fixed=st['fixed'], priors=self.priors, qoi=st['qoi'], compression=comp,
                                    max_active=st.get('max_active', 3), max_level=st.get('max_level', 3))
            s.index_set, s.candidates, s.values, s.model_evals = st['index_set'], st['candidates'], st['values'], st['model_evals']
            s.rebuild_device_tables()
            self.surrogate = s
        return self


class FieldCompression:
    """`var.compression` as gen_data.py:279-291 uses it: `.method`, `.fields`, `.coords`, `.compute_map(data_matrix)`;
    plus `.compress` / `.reconstruct` on RAW field values (the norm is fused into the HIP kernels)."""
    method = 'svd'

    def __init__(self, var: Variable, fields, reconstruction_tol: float = 0.01, rank=None):
        self.var, self.fields, self.coords = var, tuple(fields), None
        norm = 'none' if var.norm is None else ('log10' if var.norm == 'log10' else 'linear')
        scale = 1.0 if var.norm in (None, 'log10') else var.norm[1]
        self.svd = SVDCompression(norm=norm, scale=scale, reconstruction_tol=reconstruction_tol, rank=rank)

    def compute_map(self, data_matrix: dict):
        """data_matrix: field -> (num_samples, dof) NORMALISED values (as gen_data.py:287-289 builds it)."""
        import torch
        a = data_matrix[self.fields[0]]
        a = a if hasattr(a, 'is_cuda') else torch.as_tensor(np.asarray(a, dtype=np.float64))
        self.svd.fit(a.cuda().double(), normalized=True)
        return self

    rank = property(lambda self: self.svd.rank)
    relative_error = property(lambda self: getattr(self.svd, 'relative_error', None))

    def compress(self, field):
        return self.svd.compress(field)

    def reconstruct(self, latent):
        return self.svd.reconstruct(latent)

    def state(self):
        b = self.svd.basis
        return {'rank': self.svd.rank, 'basis': None if b is None else b.cpu().numpy(),
                'relative_error': getattr(self.svd, 'relative_error', None),
                'coords': None if self.coords is None else np.asarray(self.coords)}

    def load_state(self, st):
        import torch
        self.svd.rank, self.coords = st['rank'], st['coords']
        if st['basis'] is not None:
            self.svd.basis = torch.from_numpy(st['basis']).cuda()
            self.svd.relative_error = st['relative_error']
