"""Preallocated device-resident batches: the unit of work of the Monte-Carlo / Sobol' sampling loops.

A `CoupledBatch` owns the 15 SoA input arrays and the output arrays of `n` samples in HBM and evaluates
them with one `pem_coupled_f64_dev` launch on torch's current stream -- no allocation, no host copy and
no Python marshalling beyond one ctypes call per launch.  torch is used for device memory and streams only.
"""
import ctypes as C

from . import _lib, constants
from .models.coupled import COUPLED_INPUTS

QOI_NAMES = ('V_cc', 'div_angle', 'T_c')      # the reduced QoIs gathered across GPUs (24 B / sample)

# algorithmic HBM bytes per evaluation, fp64
BYTES_PER_EVAL_COUPLED = 15 * 8 + (1 + 91 + 1 + 1) * 8          # 872
BYTES_PER_EVAL_REDUCED = 15 * 8 + 3 * 8                         # 144
BYTES_PER_EVAL_MIXED = 15 * 8 + 3 * 8 + 91 * 4                  # 508 (fp32 profile)
BYTES_PER_EVAL_PLUME = 9 * 8 + (91 + 1) * 8                     # 808 (824 with T -> T_c)
BYTES_PER_EVAL_CATHODE = 7 * 8                                  # 56


class CoupledBatch:
    def __init__(self, n: int, device=None, profile: bool = True, sweep_radius: float = 1.0, mixed: bool = False,
                 thruster_qoi: bool = True, layout: str = 'soa'):
        """layout 'soa': `inputs` is [15][n], one row per variable (the dict-of-arrays the reference's callables take, as
        one tensor).  layout 'tile': `inputs` is [ceil(n / 64)][15][64] -- the 15 rows of every 64-sample kernel tile in one
        contiguous 7.5 KB block (`pem_coupled_tiled_f64_dev`; filled by `set_inputs`, `load_soa` or `Design.fill_tiled`);
        results are bit-identical.  fp64 profile or reduced-QoI mode only."""
        import torch
        _lib.load()
        _lib.require_device()
        self.n = int(n)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.profile = bool(profile)
        self.mixed = bool(mixed) and self.profile          # fp64 arithmetic, fp32 storage of the profile
        self.radius = float(sweep_radius)
        f64 = dict(dtype=torch.float64, device=self.device)
        if layout not in ('soa', 'tile'):
            raise ValueError("layout must be 'soa' or 'tile'")
        self.layout = layout
        if layout == 'tile':
            if self.mixed:
                raise NotImplementedError('the tile-interleaved input layout has an fp64-profile and a reduced-QoI entry point only')
            self.inputs = torch.zeros(((self.n + 63) // 64, len(COUPLED_INPUTS), 64), **f64)
        else:
            self.inputs = torch.empty((len(COUPLED_INPUTS), self.n), **f64)     # SoA: one row per variable
        self.qoi = torch.empty((len(QOI_NAMES), self.n), **f64)                  # V_cc, div_angle, T_c
        # the coupling variables I_B0 / T are optional outputs (not among SURVEY section 8d's 872 bytes per evaluation)
        self.I_B0 = torch.empty(self.n, **f64) if thruster_qoi else None
        self.T = torch.empty(self.n, **f64) if thruster_qoi else None
        self.j_ion = (torch.empty((self.n, _lib.NANGLE), dtype=torch.float32 if self.mixed else torch.float64,
                                  device=self.device) if self.profile else None)
        self.invalid = torch.empty(self.n, dtype=torch.uint8, device=self.device)
        self._bind()

    def _bind(self):
        p = lambda t: C.c_void_p(t.data_ptr())                                   # noqa: E731
        self._in_ptrs = [p(self.inputs[i]) for i in range(len(COUPLED_INPUTS))] if self.layout == 'soa' else None
        self._out_ptrs = [p(self.qoi[0]), p(self.I_B0) if self.I_B0 is not None else None,
                          p(self.T) if self.T is not None else None, p(self.j_ion) if self.profile else None,
                          p(self.qoi[1]), p(self.qoi[2]), p(self.invalid)]
        # bytes per sample behind each output pointer: what `run(first, count)` advances them by
        jb = (4 if self.mixed else 8) * _lib.NANGLE
        self._out_strides = [8, 8, 8, jb, 8, 8, 1]

    def _range_ptrs(self, first: int, qoi_out=None):
        """Pointers of samples first.. of every array; `qoi_out` ([3][>= count] tensor) redirects V_cc / div_angle / T_c
        (the multi-GPU pipeline writes each chunk's QoIs into its own contiguous send buffer)."""
        off = lambda ptr, nbytes: None if ptr is None else C.c_void_p(ptr.value + nbytes)   # noqa: E731
        ins = [off(q, 8 * first) for q in self._in_ptrs] if self._in_ptrs is not None else None
        outs = [off(q, st * first) for q, st in zip(self._out_ptrs, self._out_strides)]
        if qoi_out is not None:
            outs[0], outs[4], outs[5] = (C.c_void_p(qoi_out[i].data_ptr()) for i in range(3))
        return ins, outs

    def set_inputs(self, values: dict):
        """Copy a dict of arrays/tensors (keys = COUPLED_INPUTS) into the batch."""
        import torch
        if self.layout == 'tile':
            rows = torch.stack([torch.as_tensor(values[k], dtype=torch.float64).to(self.device).expand(self.n) for k in COUPLED_INPUTS])
            return self.load_soa(rows)
        for i, k in enumerate(COUPLED_INPUTS):
            self.inputs[i].copy_(torch.as_tensor(values[k], dtype=torch.float64).to(self.device).expand(self.n))

    def load_soa(self, rows, first: int = 0):
        """Copy a [15][m] tensor of input rows into samples first .. first+m-1 of the batch, whatever its layout (`first` a
        multiple of 64 for the tile layout)."""
        m = rows.shape[1]
        if self.layout == 'soa':
            self.inputs[:, first:first + m].copy_(rows)
            return
        if first % 64:
            raise ValueError('a tile-interleaved batch is loaded from a multiple of 64 samples on')
        full = m // 64
        t0 = first // 64
        if full:
            self.inputs[t0:t0 + full].copy_(rows[:, :full * 64].reshape(len(COUPLED_INPUTS), full, 64).permute(1, 0, 2))
        if m % 64:
            self.inputs[t0 + full, :, :m % 64].copy_(rows[:, full * 64:])

    def inputs_soa(self):
        """The inputs as a [15][n] tensor (a view for layout 'soa', a copy for 'tile')."""
        if self.layout == 'soa':
            return self.inputs
        return self.inputs.permute(1, 0, 2).reshape(len(COUPLED_INPUTS), -1)[:, :self.n].contiguous()

    def run(self, stream=None, first: int = 0, count: int | None = None, qoi_out=None):
        """Enqueue one coupled evaluation (asynchronous) of the whole batch, or of samples first .. first+count-1.
        The profile rows of a range must start 16-byte aligned, as the kernel's stores require: `first` even for the fp64
        profile, a multiple of 4 for the fp32 (mixed) one; a multiple of 64 for a tile-interleaved batch."""
        import torch
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        fn = _lib.load().pem_coupled_mixed_dev if self.mixed else _lib.load().pem_coupled_f64_dev
        if first == 0 and count is None and qoi_out is None:
            ins, outs, count = self._in_ptrs, self._out_ptrs, self.n
        else:
            count = self.n - first if count is None else int(count)
            if first < 0 or count < 0 or first + count > self.n:
                raise ValueError(f'range [{first}, {first + count}) does not fit a batch of {self.n} samples')
            # the profile rows of the range must start 16-byte aligned (the kernel's 16-byte stores): 728-byte fp64 rows
            # need an even `first`, 364-byte fp32 rows (mixed) a multiple of 4
            if self.profile and (first * self._out_strides[3]) % 16:
                raise ValueError(f'a range must start at a multiple of {4 if self.mixed else 2} samples: its profile rows '
                                 f'({self._out_strides[3]} bytes each) have to start 16-byte aligned')
            if self.layout == 'tile' and first % 64:
                raise ValueError('a range of a tile-interleaved batch must start at a multiple of 64 samples')
            ins, outs = self._range_ptrs(int(first), qoi_out)
        if self.layout == 'tile':
            x = C.c_void_p(self.inputs.data_ptr() + (int(first) // 64) * len(COUPLED_INPUTS) * 64 * 8)
            rc = _lib.load().pem_coupled_tiled_f64_dev(count, constants.TORR_2_PA, self.radius, x, *outs, C.c_void_p(s.cuda_stream))
        else:
            rc = fn(count, constants.TORR_2_PA, self.radius, *ins, *outs, C.c_void_p(s.cuda_stream))
        _lib.check(rc)

    def run_mc(self, design, first_index: int = 0, write_inputs: bool = False, swap_dim: int = -1, stream=None,
               first: int = 0, count: int | None = None):
        """Fused Monte-Carlo step: generate samples first_index .. first_index+count-1 of `design` (a sampling.Design over
        the 15 coupled inputs) inside the kernel and evaluate them into samples first .. first+count-1 of this batch (default:
        the whole batch); `write_inputs` also stores them in `self.inputs`.  `first` must be even, as for `run`."""
        import torch
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        if self.mixed:
            raise NotImplementedError('fused Monte-Carlo mode writes an fp64 profile or none')
        if write_inputs and self.layout != 'soa':
            raise NotImplementedError("the fused Monte-Carlo launch writes its inputs as SoA rows: use layout='soa' with write_inputs")
        count = self.n - first if count is None else int(count)
        if first < 0 or count < 0 or first + count > self.n or first & 1:
            raise ValueError(f'range [{first}, {first + count}) does not fit a batch of {self.n} samples or starts at an odd sample')
        outs = self._out_ptrs if first == 0 else self._range_ptrs(int(first))[1]
        ptr = lambda arr: C.c_void_p(arr.ctypes.data)                                           # noqa: E731
        rc = _lib.load().pem_coupled_mc_f64_dev(
            count, int(first_index), design.seed, design.stream, int(swap_dim), ptr(design.kind), ptr(design.a), ptr(design.b),
            constants.TORR_2_PA, self.radius, C.c_void_p(self.inputs.data_ptr() + 8 * int(first)) if write_inputs else None,
            self.inputs.stride(0), *outs, C.c_void_p(s.cuda_stream))
        _lib.check(rc)

    def run_loglik(self, likelihood, out=None, stream=None):
        """Coupled evaluation + `likelihood.JionLikelihood.per_sample` in one launch (`pem_coupled_loglik_f64_dev`): the
        profile is reduced against the measurements in LDS and never written.  Sample i belongs to condition
        i mod Ne.  Returns the (n,) per-sample sums; V_cc / div_angle / T_c / invalid are written as by `run`."""
        import torch
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        if self.layout != 'soa':
            raise NotImplementedError("the fused likelihood launch reads SoA inputs: use layout='soa'")
        if out is None:
            out = torch.empty(self.n, dtype=torch.float64, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())                                   # noqa: E731
        lk = likelihood
        rc = _lib.load().pem_coupled_loglik_f64_dev(
            self.n, constants.TORR_2_PA, self.radius, *self._in_ptrs, lk.n_cond, lk.n_ang, p(lk.kidx), p(lk.weight),
            p(lk.y), p(lk.inv_std), p(self.qoi[0]), p(self.qoi[1]), p(self.qoi[2]), p(out), p(self.invalid),
            C.c_void_p(s.cuda_stream))
        _lib.check(rc)
        return out

    def run_latent(self, compression, out=None, stream=None):
        """Coupled evaluation + `compression.SVDCompression.compress(j_ion)` in one launch
        (`pem_coupled_latent_f64_dev`): the latents are accumulated in the registers of the angle loop, the profile is
        never stored.  Returns the (n, rank) latents; V_cc / div_angle / T_c / invalid are written as by `run`."""
        import torch
        from .compression import NORM_LINEAR
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        if self.layout != 'soa':
            raise NotImplementedError("the fused compression launch reads SoA inputs: use layout='soa'")
        c = compression
        if c.norm == NORM_LINEAR:
            raise NotImplementedError('the fused mode covers norm none / log10 (j_ion); use run() + compress()')
        basis = c.basis.contiguous()
        if basis.shape[0] != _lib.NANGLE:
            raise ValueError('the compression map is not one of the 91-point profile')
        if out is None:
            out = torch.empty((self.n, c.rank), dtype=torch.float64, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())                                   # noqa: E731
        rc = _lib.load().pem_coupled_latent_f64_dev(
            self.n, constants.TORR_2_PA, self.radius, *self._in_ptrs, int(c.rank), int(c.norm), p(basis), p(out),
            p(self.qoi[0]), p(self.qoi[1]), p(self.qoi[2]), p(self.invalid), C.c_void_p(s.cuda_stream))
        _lib.check(rc)
        return out

    def outputs(self) -> dict:
        out = {'V_cc': self.qoi[0], 'div_angle': self.qoi[1], 'T_c': self.qoi[2], 'invalid': self.invalid.bool()}
        if self.I_B0 is not None:
            out['I_B0'], out['T'] = self.I_B0, self.T
        if self.profile:
            out['j_ion'] = self.j_ion
        return out

    @property
    def bytes_per_eval(self) -> int:
        if not self.profile:
            return BYTES_PER_EVAL_REDUCED
        return BYTES_PER_EVAL_MIXED if self.mixed else BYTES_PER_EVAL_COUPLED
