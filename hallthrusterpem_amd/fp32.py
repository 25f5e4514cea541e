"""Single-precision arithmetic for the scalar QoIs (SURVEY.md section 8b "fp32/mixed entry points optional with a tolerance
report", section 8d config 5: "fp32 run compared with fp64 on identical inputs; report max / 99.9-pct relative error per
QoI").  csrc/pem_fp32.hip holds the kernels; this module is the host side:

  CoupledBatchF32       device-resident [15][n] float inputs -> [3][n] float QoIs (V_cc, div_angle, T_c), one launch
  compare_with_fp64     the tolerance report: the fp64 reduced-QoI kernel and the fp32 one on IDENTICAL inputs (the
                        float-rounded design, widened back to double for the fp64 run), per-QoI error statistics and
                        the two kernels' durations
  saltelli_sums         the fused Saltelli design (one launch: design rows, n_varied + 2 evaluations per base sample and
                        the estimator sums); drivers.sobol_indices(..., precision='fp32') is built on it
"""
import ctypes as C

import numpy as np

from . import _lib, constants
from .batch import QOI_NAMES
from .models.coupled import COUPLED_INPUTS


class CoupledBatchF32:
    def __init__(self, n: int, device=None, sweep_radius: float = 1.0):
        import torch
        _lib.load()
        _lib.require_device()
        self.n = int(n)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.radius = float(sweep_radius)
        self.inputs = torch.empty((len(COUPLED_INPUTS), self.n), dtype=torch.float32, device=self.device)
        self.qoi = torch.empty((len(QOI_NAMES), self.n), dtype=torch.float32, device=self.device)
        self.invalid = torch.empty(self.n, dtype=torch.uint8, device=self.device)

    bytes_per_eval = 15 * 4 + 3 * 4        # 72: half of the fp64 reduced-QoI mode's 144

    def run(self, stream=None):
        import torch
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        p = lambda t: C.c_void_p(t.data_ptr())                                   # noqa: E731
        _lib.check(_lib.load().pem_coupled_f32_dev(self.n, constants.TORR_2_PA, self.radius, p(self.inputs), self.inputs.stride(0),
                                                   p(self.qoi), self.qoi.stride(0), p(self.invalid), C.c_void_p(s.cuda_stream)))

    def outputs(self) -> dict:
        return {'V_cc': self.qoi[0], 'div_angle': self.qoi[1], 'T_c': self.qoi[2], 'invalid': self.invalid.bool()}


def _event_ms(fn, reps=20):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def compare_with_fp64(n: int, fill=None, seed: int = 2) -> dict:
    """fp32 arithmetic against fp64 arithmetic on identical inputs.  `fill(batch)` writes the fp64 inputs of a
    `batch.CoupledBatch` (default: the counter-based PEM-v0 prior design, seed `seed`); they are rounded to float, the fp32
    kernel runs on the floats and the fp64 reduced-QoI kernel on the same floats widened to double.  Returns, per QoI,
    the max / 99.9th-percentile / median relative error of the fp32 result, and both kernels' durations."""
    import torch
    from .batch import CoupledBatch
    from .sampling import Design
    f64 = CoupledBatch(n, profile=False, thruster_qoi=False)
    if fill is None:
        Design(seed=seed).fill(f64.inputs)
    else:
        fill(f64)
    f32 = CoupledBatchF32(n, device=f64.device)
    f32.inputs.copy_(f64.inputs)               # rounds to float
    f64.inputs.copy_(f32.inputs)               # ... and the fp64 run sees exactly those values
    f64.run()
    f32.run()
    torch.cuda.synchronize()
    report = {'samples': n, 'inputs': 'identical: the float-rounded design, widened to double for the fp64 kernel',
              'invalid_flags_identical': bool(torch.equal(f64.invalid, f32.invalid)), 'qoi': {}}
    for i, name in enumerate(QOI_NAMES):
        ref = f64.qoi[i]
        rel = ((f32.qoi[i].double() - ref).abs() / ref.abs().clamp_min(1e-300))
        rel = rel[torch.isfinite(rel) & (ref != 0)]
        k999 = max(1, int(0.999 * rel.numel()))
        report['qoi'][name] = {'max_rel': float(rel.max()), 'p999_rel': float(rel.kthvalue(k999).values), 'median_rel': float(rel.median()),
                               'compared': int(rel.numel())}
    ms64, ms32 = _event_ms(f64.run), _event_ms(f32.run)
    report['kernel_us'] = {'fp64_reduced_qoi (144 B/eval)': 1e3 * ms64, 'fp32 (72 B/eval)': 1e3 * ms32, 'speedup': ms64 / ms32}
    report['evals_per_s'] = {'fp64_reduced_qoi': n / (ms64 * 1e-3), 'fp32': n / (ms32 * 1e-3)}
    report['fp32_GBs'] = CoupledBatchF32.bytes_per_eval * n / (ms32 * 1e-3) / 1e9
    return report


def saltelli_sums(design, varied, n_base: int, first_index: int = 0, radius: float = 1.0, n_blocks: int | None = None, device=None, stream=None,
                  precision: str = 'fp32'):
    """One launch of the fused Saltelli design (`pem_saltelli_f32_dev` / `pem_saltelli_f64_dev`, csrc/pem_saltelli.hip): base
    samples first_index .. first_index+n_base-1 of `design` (rows A: its stream, B: stream + 1), `varied` = indices of the
    inputs that get an AB block; `precision` picks the model (fp32 arithmetic on the design rounded to float, or fp64).
    Returns (sums [2 + 2 nv][3] float64 CUDA tensor, flags [2] int64: non-physical thruster results, invalid plume samples)."""
    import torch
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    s = torch.cuda.current_stream(dev) if stream is None else stream
    varied = np.ascontiguousarray(varied, dtype=np.int32)
    nv = int(varied.size)
    if n_blocks is None:
        # workgroups of 256 base samples at a time: two per CU hold the fp64 model's 198 registers per lane; the fp32 model (127)
        # runs four waves per SIMD and keeps gaining up to eight workgroups per CU (0.38 -> 0.28 ms for the 2e7-evaluation design,
        # profiles/saltelli_r03.txt)
        n_blocks = 512 if precision != 'fp32' else 2048
    partial = torch.empty((n_blocks, 2 + 2 * nv, len(QOI_NAMES)), dtype=torch.float64, device=dev)
    flags = torch.empty((n_blocks, 2), dtype=torch.int64, device=dev)
    ptr = lambda arr: C.c_void_p(arr.ctypes.data)                                               # noqa: E731
    with torch.cuda.device(dev):
        fn = _lib.load().pem_saltelli_f32_dev if precision == 'fp32' else _lib.load().pem_saltelli_f64_dev
        _lib.check(fn(int(n_base), int(first_index), design.seed, design.stream, ptr(design.kind), ptr(design.a), ptr(design.b), nv,
                      ptr(varied), constants.TORR_2_PA, float(radius), C.c_void_p(partial.data_ptr()), C.c_void_p(flags.data_ptr()),
                      int(n_blocks), C.c_void_p(s.cuda_stream)))
    return partial.sum(dim=0), flags.sum(dim=0)
