"""Thruster stage -- the parts of `hallmd.models.thruster` (src/hallmd/models/thruster.py) that are
arithmetic the reference owns, plus the analytic stage that couples cathode -> plume on the GPU.

In scope:
  * PEM <-> HallThruster.jl name mapping  `_convert_to_julia` / `_convert_to_pem`   (thruster.py:93-137)
  * fidelity -> grid / charge states / CFL time step  `_default_model_fidelity`      (thruster.py:140-181)
  * input formatting incl. the anomalous-coefficient rescale  `_format_hallthruster_jl_input`
                                                                                     (thruster.py:184-278)
  * the non-physical / shock filters of `hallthruster_jl`  `check_thruster_outputs`  (thruster.py:490-502)
  * `thruster_analytic`: tests/sim_hallthruster.jl:35-48, the reference's own closed-form stand-in for the
    solver, evaluated batched on the GPU.  It is a TEST DOUBLE: it lets V_cc -> (I_B0, T) -> plume be
    coupled without the external 1-D fluid solver, and says nothing about real thruster physics.

Out of scope: launching HallThruster.jl (a third-party Julia program run as a subprocess,
thruster.py:281-375) and reading device files from disk (hallmd.utils.load_thruster).
"""
import copy
import random
import string
import warnings
from pathlib import Path

import numpy as np

from .. import _lib, _marshal as m, constants

__all__ = ['PEM_TO_JULIA', 'thruster_analytic', 'check_thruster_outputs']

# PEM variable name -> path into the HallThruster.jl input/output structure.  These are interface
# names shared with the solver's JSON schema (the reference keeps them in models/pem_to_julia.json).
_CFG, _ANOM = 'config', 'anom_model'
PEM_TO_JULIA = {
    'P_b': [_CFG, 'background_pressure_Torr'], 'mdot_a': [_CFG, 'anode_mass_flow_rate'],
    'V_cc': [_CFG, 'cathode_coupling_voltage'], 'u_n': [_CFG, 'neutral_velocity'], 'T_e': [_CFG, 'cathode_Tev'],
    'l_t': [_CFG, 'transition_length'], 'V_a': [_CFG, 'discharge_voltage'],
    'dz': [_CFG, _ANOM, 'dz'], 'z0': [_CFG, _ANOM, 'z0'], 'p0': [_CFG, _ANOM, 'pstar'], 'alpha': [_CFG, _ANOM, 'alpha'],
    'a_1': [_CFG, _ANOM, 'model', 'c1'], 'a_2': [_CFG, _ANOM, 'model', 'c2'],
    'anom_min': [_CFG, _ANOM, 'model', 'hall_min'], 'anom_max': [_CFG, _ANOM, 'model', 'hall_max'],
    'anom_center': [_CFG, _ANOM, 'model', 'center'], 'anom_width': [_CFG, _ANOM, 'model', 'width'],
    'anom_scale': [_CFG, _ANOM, 'model', 'anom_scale'], 'anom_barrier_scale': [_CFG, _ANOM, 'model', 'barrier_scale'],
    'anom_shift_length': [_CFG, _ANOM, 'shift_length'], 'f_n': [_CFG, 'neutral_ingestion_multiplier'],
    'c_w': [_CFG, 'wall_loss_model', 'loss_scale'], 'ncharge': [_CFG, 'ncharge'], 'B_hat': [_CFG, 'magnetic_field_scale'],
    'num_cells': ['simulation', 'grid', 'num_cells'], 'dt': ['simulation', 'dt'],
    'I_B0': ['output', 'average', 'ion_current'], 'I_d': ['output', 'average', 'discharge_current'],
    'T': ['output', 'average', 'thrust'], 'eta_c': ['output', 'average', 'current_eff'],
    'eta_m': ['output', 'average', 'mass_eff'], 'eta_v': ['output', 'average', 'voltage_eff'],
    'eta_a': ['output', 'average', 'anode_eff'], 'u_ion': ['output', 'average', 'ui', 0],
    'u_ion_coords': ['output', 'average', 'z'],
}


def _convert_to_julia(pem_data: dict, julia_data: dict, pem_to_julia: dict):
    """Write every PEM value into `julia_data` (in place) at the path `pem_to_julia` gives it, creating the
    containers on the way: a dict where the next path element is a str, a list where it is an int, lists
    grown to reach an index.  The leaf itself is assigned, not created, so an out-of-range list index at the
    END of a path raises IndexError as in the reference (thruster.py:93-118; tests/test_thruster.py:43-61)."""
    for name, value in pem_data.items():
        if name not in pem_to_julia:
            raise KeyError(f"Cannot convert PEM data variable {name} since it is not in the provided conversion map")
        path = pem_to_julia[name]
        node = julia_data
        for here, after in zip(path[:-1], path[1:]):
            fresh = (lambda: {}) if isinstance(after, str) else (lambda: [])
            if isinstance(node, dict):
                if not node.get(here):
                    node.setdefault(here, fresh())
            elif isinstance(node, list):
                while len(node) <= here:
                    node.append(fresh())
            node = node[here]
        node[path[-1]] = value


def _convert_to_pem(julia_data: dict, pem_to_julia: dict) -> dict:
    """Collect the PEM outputs: every mapped path that starts at "output" and exists (thruster.py:121-137)."""
    found = {}
    for name, path in pem_to_julia.items():
        if path[0] != 'output':
            continue
        node = julia_data
        try:
            for key in path:
                node = node[key]
        except (KeyError, IndexError):
            continue
        found[name] = node
    return found


def _default_model_fidelity(model_fidelity: tuple, json_config: dict, cfl: float = 0.2) -> dict:
    """(f0, f1) -> num_cells = 50 (f0 + 2), ncharge = f1 + 1 and a uniform time step from the CFL condition on
    the ion exhaust speed (thruster.py:140-181)."""
    if model_fidelity == ():
        model_fidelity = (2, 2)
    num_cells = 50 * (model_fidelity[0] + 2)
    ncharge = model_fidelity[1] + 1
    cfg = json_config.get('config', {})
    domain = cfg.get('domain', [0, 0.08])
    v_anode = cfg.get('discharge_voltage', 300)
    v_cathode = cfg.get('cathode_coupling_voltage', 0)
    gas = cfg.get('propellant', 'Xenon')
    if gas not in constants.MOLECULAR_WEIGHTS:
        warnings.warn(f"Could not find propellant {gas}; defaulting to Xenon for the CFL time step estimate.")
        gas = 'Xenon'
    ion_mass = constants.MOLECULAR_WEIGHTS[gas] / constants.AVOGADRO_CONSTANT / 1000      # kg
    cell = float(domain[1]) / (num_cells + 1)
    speed = np.sqrt(2 * ncharge * constants.FUNDAMENTAL_CHARGE * (v_anode - v_cathode) / ion_mass)
    return {'num_cells': num_cells, 'ncharge': ncharge, 'dt': float(cfl * cell / speed)}


def _format_hallthruster_jl_input(thruster_inputs: dict, pem_to_julia: dict, thruster=None, config=None,
                                  simulation=None, postprocess=None, model_fidelity=(2, 2), output_path=None,
                                  fidelity_function=None) -> dict:
    """Build the `{'config', 'simulation', 'postprocess'}` dict HallThruster.run_simulation expects
    (thruster.py:184-278).  `thruster` must be a dict (or None): reading a device directory from disk
    belongs to hallmd.utils.load_thruster, which is out of scope here."""
    doc = {key: copy.deepcopy(val) if val is not None else {}
           for key, val in (('config', config), ('simulation', simulation), ('postprocess', postprocess))}
    if isinstance(thruster, (str, Path)):
        raise NotImplementedError('pass the thruster as a dict (hallmd.utils.load_thruster reads device files)')
    if thruster is not None:
        doc['config']['thruster'] = thruster
    duration = doc['simulation'].get('duration', 1e-3)
    doc['postprocess']['average_start_time'] = doc['postprocess'].get('average_start_time', 0.5 * duration)
    _convert_to_julia(thruster_inputs, doc, pem_to_julia)
    if model_fidelity is not None:
        fid = (fidelity_function or _default_model_fidelity)(model_fidelity, doc)
        _convert_to_julia(fid, doc, pem_to_julia)
    if output_path is not None:
        stem = 'hallthruster_jl'
        if name := doc['config'].get('thruster', {}).get('name'):
            stem += f'_{name}'
        if volts := doc['config'].get('discharge_voltage'):
            stem += f'_{round(volts)}V'
        if flow := doc['config'].get('anode_mass_flow_rate'):
            stem += f'_{flow:.1e}kg_s'
        stem += '_' + ''.join(random.choices(string.ascii_uppercase + string.digits, k=4)) + '.json'
        doc['postprocess']['output_file'] = str((Path(output_path) / stem).resolve())
    # the PEM's a_2 / anom_max are RATIOS to a_1 / anom_min (thruster.py:266-276)
    if anom := doc['config'].get('anom_model'):
        if anom.get('type') in ('LogisticPressureShift', 'SimpleLogisticShift'):
            anom = anom.get('model', {})
        kind = anom.get('type', 'TwoZoneBohm')
        if kind == 'TwoZoneBohm' and thruster_inputs.get('a_2') is not None:
            anom['c2'] = anom['c2'] * anom.get('c1', 0.00625)
        elif kind == 'GaussianBohm' and thruster_inputs.get('anom_max') is not None:
            anom['hall_max'] = anom['hall_max'] * anom.get('hall_min', 0.00625)
    return doc


def check_thruster_outputs(outputs: dict, shock_threshold: float | None = None):
    """The two filters `hallthruster_jl` applies to a finished run (thruster.py:490-502), batched.

    Scalar QoIs raise ValueError exactly as the reference does; array QoIs return a boolean `bad` mask
    (True = the reference would have raised for that sample) so a sampling loop can drop them.  CUDA tensors are
    filtered on the device (`pem_thruster_filter_f64_dev`) and a CUDA bool tensor comes back."""
    if m.any_device_tensor([v for v in outputs.values() if v is not None]):
        return _check_on_device(outputs, shock_threshold)
    thrust = np.asarray(outputs.get('T', 0), dtype=np.float64)
    beam = np.asarray(outputs.get('I_B0', 0), dtype=np.float64)
    bad = (thrust < 0) | (beam < 0)
    where_max = None
    if shock_threshold is not None and outputs.get('u_ion') is not None and outputs.get('u_ion_coords') is not None:
        u = np.asarray(outputs['u_ion'], dtype=np.float64)
        z = np.asarray(outputs['u_ion_coords'], dtype=np.float64)
        idx = np.argmax(u, axis=-1)
        where_max = np.take_along_axis(np.broadcast_to(z, u.shape), idx[..., None], axis=-1)[..., 0]
        bad = bad | (where_max < shock_threshold)
    if bad.ndim == 0:
        if thrust < 0 or beam < 0:
            raise ValueError(f'Exception due to non-physical case: thrust={thrust} N, beam current={beam} A')
        if bad:
            raise ValueError(f'Exception due to shock-like behavior: max ion velocity occurs at z={float(where_max):.3f} m')
    return bad


def _check_on_device(outputs: dict, shock_threshold):
    import ctypes as C

    import torch
    lib = _lib.load()
    dev = m.pick_device([v for v in outputs.values() if v is not None])
    u, z = outputs.get('u_ion'), outputs.get('u_ion_coords')
    use_shock = shock_threshold is not None and u is not None and z is not None
    ref = u if u is not None else outputs.get('T', outputs.get('I_B0'))
    n = int(ref.shape[0]) if ref.dim() > 0 else 1
    with torch.cuda.device(dev):
        T = m.dev_flat(outputs['T'], (n,), dev) if outputs.get('T') is not None else None
        IB = m.dev_flat(outputs['I_B0'], (n,), dev) if outputs.get('I_B0') is not None else None
        ncells = int(u.shape[-1]) if use_shock else 0
        uu = u.to(device=dev, dtype=torch.float64).contiguous().reshape(n, ncells) if use_shock else None
        zz = m.dev_flat(z, (ncells,), dev) if use_shock else None
        flags = torch.empty(n, dtype=torch.uint8, device=dev)
        _lib.check(lib.pem_thruster_filter_f64_dev(n, ncells, m.t_ptr(uu), m.t_ptr(zz),
                                                    float(shock_threshold) if use_shock else 0.0, int(use_shock),
                                                    m.t_ptr(T), m.t_ptr(IB), m.t_ptr(flags), m.current_stream_ptr(dev)))
    return flags != 0


_IN = ('V_a', 'V_cc', 'mdot_a', 'a_1')
_OUT = ('I_B0', 'I_d', 'T', 'eta_c', 'eta_m', 'eta_v', 'eta_a', 'v_exh')


def thruster_analytic(inputs: dict, num_cells: int | None = None, domain=(0.0, 0.08)) -> dict:
    """Closed-form thruster stage of tests/sim_hallthruster.jl:35-48, batched on the GPU (a test double).

    :param inputs: `V_a` (V), `V_cc` (V), `mdot_a` (kg/s), `a_1` (the script's anom_model c1).
    :param num_cells: if given, also return the ion velocity profile `u_ion` (..., num_cells) on
                      `u_ion_coords` = range(domain[0], domain[1], length=num_cells) (script lines 46-47).
    :returns: `I_B0`, `I_d`, `T`, `eta_c`, `eta_m`, `eta_v`, `eta_a`, `v_exh`, each over the loop shape.
    """
    vals = [inputs[k] for k in _IN]
    lib = _lib.load()
    shape = m.loop_shape(vals)
    n = int(np.prod(shape))
    import torch
    on_device = m.any_device_tensor(vals)
    if on_device:
        dev = m.pick_device(vals)
        with torch.cuda.device(dev):
            flat = [m.dev_flat(v, shape, dev) for v in vals]
            outs = [torch.empty(n, dtype=torch.float64, device=dev) for _ in _OUT]
            _lib.check(lib.pem_thruster_f64_dev(n, *[m.t_ptr(t) for t in flat], *[m.t_ptr(t) for t in outs],
                                                m.current_stream_ptr(dev)))
        ret = {k: o.reshape(shape) for k, o in zip(_OUT, outs)}
    else:
        flat = [m.host_flat(v, shape) for v in vals]
        outs = [np.empty(n, dtype=np.float64) for _ in _OUT]
        _lib.check(lib.pem_thruster_f64(n, *[m.np_ptr(a) for a in flat], *[m.np_ptr(a) for a in outs]))
        ret = {k: o.reshape(shape) for k, o in zip(_OUT, outs)}
    if num_cells is not None:
        _lib.require_device()
        dev = m.pick_device(vals) if on_device else torch.device('cuda', torch.cuda.current_device())
        with torch.cuda.device(dev):
            v = m.dev_flat(ret['v_exh'], shape, dev)
            z = torch.empty(int(num_cells), dtype=torch.float64, device=dev)
            u = torch.empty((n, int(num_cells)), dtype=torch.float64, device=dev)
            _lib.check(lib.pem_thruster_uion_f64_dev(n, m.t_ptr(v), float(domain[0]), float(domain[1]), int(num_cells),
                                                      m.t_ptr(z), m.t_ptr(u), m.current_stream_ptr(dev)))
        u = u.reshape(shape + (int(num_cells),))
        ret['u_ion'] = u if on_device else u.cpu().numpy()
        ret['u_ion_coords'] = z if on_device else z.cpu().numpy()
    return ret
