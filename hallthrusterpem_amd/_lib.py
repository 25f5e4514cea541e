"""ctypes binding of libpem_hip.so (C ABI declared in include/pem_hip.h).

There is deliberately no CPU fallback here: if the shared library is missing, or no HIP device is
present, the model functions raise.  The CPU oracle under oracle/ is test infrastructure and is
never imported from this package.
"""
import ctypes as C
import importlib.util
import os
import threading
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / 'libpem_hip.so'

PEM_OK, PEM_ERR_INVALID_ARG, PEM_ERR_HIP, PEM_ERR_NO_DEVICE = 0, 1, 2, 3
NANGLE = 91
FUSED_LATENT_MAX_RANK = 8     # PEM_FUSED_LATENT_MAX_RANK (include/pem_hip.h): latents the fused model -> compression launch keeps

_dp = C.c_void_p          # every array crosses the boundary as a raw pointer
_sz = C.c_size_t
_f8 = C.c_double

# name -> (restype, argtypes); one entry per declaration in include/pem_hip.h
SIGNATURES = {
    'pem_version': (C.c_char_p, []),
    'pem_last_error': (C.c_char_p, []),
    'pem_device_count': (C.c_int, []),
    'pem_init': (C.c_int, [C.c_int]),
    'pem_synchronize': (C.c_int, [_dp]),
    'pem_set_lanes_per_sample': (C.c_int, [C.c_int]),
    'pem_angle_grid': (C.POINTER(C.c_double), []),
    'pem_persistent_grid': (C.c_int, [_sz, C.c_int, C.c_int, C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    'pem_coupled_occupancy': (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'pem_cathode_f64_dev': (C.c_int, [_sz] + [_dp] * 6 + [_f8, _dp, _dp]),
    'pem_cathode_f64': (C.c_int, [_sz] + [_dp] * 6 + [_f8, _dp]),
    'pem_plume_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _f8] + [_dp] * 10 + [_dp] * 4 + [_dp]),
    'pem_plume_f64': (C.c_int, [_sz, C.c_int, _dp, _f8] + [_dp] * 10 + [_dp] * 4),
    'pem_thruster_f64_dev': (C.c_int, [_sz] + [_dp] * 4 + [_dp] * 8 + [_dp]),
    'pem_thruster_f64': (C.c_int, [_sz] + [_dp] * 4 + [_dp] * 8),
    'pem_thruster_uion_f64_dev': (C.c_int, [_sz, _dp, _f8, _f8, C.c_int, _dp, _dp, _dp]),
    'pem_thruster_filter_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _dp, _f8, C.c_int, _dp, _dp, _dp, _dp]),
    'pem_coupled_f64_dev': (C.c_int, [_sz, _f8, _f8] + [_dp] * 15 + [_dp] * 7 + [_dp]),
    'pem_coupled_f64': (C.c_int, [_sz, _f8, _f8] + [_dp] * 15 + [_dp] * 7),
    'pem_coupled_tiled_f64_dev': (C.c_int, [_sz, _f8, _f8, _dp] + [_dp] * 7 + [_dp]),
    'pem_coupled_mixed_dev': (C.c_int, [_sz, _f8, _f8] + [_dp] * 15 + [_dp] * 7 + [_dp]),
    'pem_coupled_loglik_f64_dev': (C.c_int, [_sz, _f8, _f8] + [_dp] * 15 + [C.c_int, C.c_int] + [_dp] * 4 + [_dp] * 5 + [_dp]),
    'pem_coupled_latent_f64_dev': (C.c_int, [_sz, _f8, _f8] + [_dp] * 15 + [C.c_int, C.c_int, _dp, _dp] + [_dp] * 4 + [_dp]),
    'pem_loglik_marginal_f64_dev': (C.c_int, [_sz, C.c_int, C.c_int, _dp, _dp, _dp, _f8, _f8, _dp, _dp, _dp]),
    'pem_log_prior_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]),
    'pem_jion_loglik_f64_dev': (C.c_int, [_sz, C.c_int, C.c_int] + [_dp] * 6 + [_dp]),
    'pem_svd_compress_f64_dev': (C.c_int, [_sz, C.c_int, C.c_int, C.c_int, _f8, _dp, _dp, _dp, _dp]),
    'pem_svd_reconstruct_f64_dev': (C.c_int, [_sz, C.c_int, C.c_int, C.c_int, _f8, _dp, _dp, _dp, _dp]),
    'pem_coupled_mc_f64_dev': (C.c_int, [_sz, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, _dp, _dp, _dp, _f8, _f8, _dp, _sz] + [_dp] * 7 + [_dp]),
    'pem_sparse_predict_f64_dev': (C.c_int, [_sz, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _sz, _dp, _sz, C.c_int, C.c_int, _dp]),
    'pem_sparse_grid_values_f64_dev': (C.c_int, [_sz, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _sz, _dp, _sz, C.c_int, C.c_int, _dp]),
    'pem_sparse_predict_field_f64_dev': (C.c_int, [_sz, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _sz, _dp, _sz, C.c_int, C.c_int,
                                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp]),
    'pem_key_minmax_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, _dp, _dp, _dp, _dp]),
    'pem_range_hist_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, C.c_int, _dp, _dp, C.c_int, _dp, _dp]),
    'pem_range_narrow_dev': (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    'pem_qsel_bins': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'pem_qsel_minmax_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, _dp, _dp, _dp, _dp]),
    'pem_qsel_hist1_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, _dp, _dp, C.c_int, _dp, _dp]),
    'pem_qsel_decide1_dev': (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    'pem_qsel_hist2_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, _dp, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp]),
    'pem_qsel_decide2_dev': (C.c_int, [C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    'pem_qsel_compact_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, _dp, _dp, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_uint32, _dp, _dp, _dp]),
    'pem_qsel_select_dev': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint32, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    'pem_quantiles_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    'pem_quantiles_strided_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, _sz, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    'pem_quantiles_last_path': (C.c_int, []),
    'pem_coupled_mc_stats_f64_dev': (C.c_int, [_sz, C.c_uint64, C.c_uint64, C.c_uint32, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _sz,
                                               _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp,
                                               C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp]),
    'pem_row_masks_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _sz, _dp, _dp, _dp, _dp, _dp]),
    'pem_campaign_masks_f64_dev': (C.c_int, [_sz, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _sz, _dp, _dp, C.c_int, _dp, _dp,
                                             C.c_int, _dp]),
    'pem_sobol_partial_f64_dev': (C.c_int, [_sz, C.c_int, _sz, _dp, _dp, _dp, _dp, C.c_int, _dp]),
    'pem_coupled_f32_dev': (C.c_int, [_sz, C.c_float, C.c_float, _dp, _sz, _dp, _sz, _dp, _dp]),
    'pem_saltelli_f32_dev': (C.c_int, [_sz, C.c_uint64, C.c_uint64, C.c_uint32, _dp, _dp, _dp, C.c_int, _dp, C.c_float, C.c_float, _dp, _dp, C.c_int, _dp]),
    'pem_saltelli_f64_dev': (C.c_int, [_sz, C.c_uint64, C.c_uint64, C.c_uint32, _dp, _dp, _dp, C.c_int, _dp, C.c_double, C.c_double, _dp, _dp, C.c_int, _dp]),
    'pem_sample_f64_dev': (C.c_int, [_sz, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _sz, _dp]),
    'pem_sample_tiled_f64_dev': (C.c_int, [_sz, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp]),
    'pem_sample_lhs_f64_dev': (C.c_int, [_sz, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, _dp, _dp, _dp, _dp, _sz, _dp]),
}

_lib = None
_hip_runtime = None
_lock = threading.Lock()


class PemHipError(RuntimeError):
    """A libpem_hip entry point returned a non-zero status."""

    def __init__(self, code, message):
        super().__init__(f'libpem_hip error {code}: {message}')
        self.code = code


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one process
    cannot both own the GPU ("No HIP GPUs are available" from whichever initialises second), and device
    pointers/streams are only meaningful inside one runtime.  So if torch is installed, map ITS runtime
    first: libpem_hip.so's DT_NEEDED libamdhip64.so.7 then resolves to the already-loaded SONAME.
    Set PEM_HIP_RUNTIME=system to skip this and use /opt/rocm's runtime (numpy-only use)."""
    if os.environ.get('PEM_HIP_RUNTIME', '').lower() == 'system':
        return None
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = Path(list(spec.submodule_search_locations)[0]) / 'lib' / 'libamdhip64.so'
    if not cand.exists():
        return None
    return C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def load():
    """Load libpem_hip.so and bind every symbol.  The in-tree library is rebuilt first when it is missing or older than
    one of its sources (build.needs_build) and hipcc is available -- the .so is git-ignored, so a stale binary would
    otherwise be tested silently after a source edit.  PEM_HIP_LIB names an experimental build and is loaded as is."""
    global _lib, _hip_runtime
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = Path(os.environ.get('PEM_HIP_LIB', LIB_PATH))      # PEM_HIP_LIB: experimental builds (tools/build_variant.sh)
        if 'PEM_HIP_LIB' not in os.environ:
            from . import build as _build
            if _build.needs_build():
                if _build.have_hipcc():
                    _build.build()
                elif not path.exists():
                    raise RuntimeError('libpem_hip.so is missing and hipcc is not available to build it')
                else:
                    import warnings
                    warnings.warn('libpem_hip.so is older than its sources and hipcc is not available: loading the stale library')
        _hip_runtime = _share_torch_hip_runtime()
        lib = C.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError here = the library does not match the header
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int):
    if rc != PEM_OK:
        raise PemHipError(rc, load().pem_last_error().decode(errors='replace'))


def device_count() -> int:
    return int(load().pem_device_count())


def set_device(index: int):
    """Select the GPU of this process (one process per GPU): `pem_init(index)` makes it the device of the host-pointer
    entry points for every thread of the process, and torch's current device is set to match."""
    check(load().pem_init(int(index)))
    try:
        import torch
        torch.cuda.set_device(int(index))
    except ImportError:
        pass


def persistent_grid(n: int, cus: int, wg_per_cu: int, memory_bound: bool = True):
    """(workgroups, samples per round) of a persistent launch over n samples: pure arithmetic, runs without a GPU."""
    wg, spr = _sz(0), _sz(0)
    check(load().pem_persistent_grid(int(n), int(cus), int(wg_per_cu), 1 if memory_bound else 0, C.byref(wg), C.byref(spr)))
    return int(wg.value), int(spr.value)


def coupled_occupancy(profile_mode: int = 1):
    """(compute units, resident workgroups per CU) of the coupled kernel on the current device; needs the GPU."""
    cus, per = C.c_int(0), C.c_int(0)
    check(load().pem_coupled_occupancy(int(profile_mode), C.byref(cus), C.byref(per)))
    return int(cus.value), int(per.value)


def require_device():
    if device_count() < 1:
        raise PemHipError(PEM_ERR_NO_DEVICE, 'no HIP device visible; hallthrusterpem_amd has no CPU path')
