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

  * `hallthruster_jl`: the model callable itself (thruster.py:378-512), same signature and return keys; the run between
    "format the inputs" and "convert the outputs" goes through a pluggable `run_simulation(json_input, ...)` whose
    default, `run_analytic_double`, evaluates what tests/sim_hallthruster.jl computes -- on the GPU, batched.

Out of scope: launching HallThruster.jl (a third-party Julia program run as a subprocess, thruster.py:281-375); a
caller who has it passes `run_simulation=hallmd.models.thruster.run_hallthruster_jl`.
"""
import copy
import json
import random
import string
import time
import warnings
from pathlib import Path

import numpy as np

from .. import _lib, _marshal as m, constants

__all__ = ['PEM_TO_JULIA', 'HALLTHRUSTER_VERSION_DEFAULT', 'hallthruster_jl', 'run_analytic_double', 'thruster_analytic',
           'check_thruster_outputs']

HALLTHRUSTER_VERSION_DEFAULT = '0.18.7'      # thruster.py:38; only names the Julia environment handed to a custom backend

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
    energy = 2 * ncharge * constants.FUNDAMENTAL_CHARGE * (v_anode - v_cathode) / ion_mass
    if m.is_torch(energy):                                  # a batch of runs whose voltages live on the GPU
        return {'num_cells': num_cells, 'ncharge': ncharge, 'dt': cfl * cell / energy.sqrt()}
    dt = cfl * cell / np.sqrt(energy)
    return {'num_cells': num_cells, 'ncharge': ncharge, 'dt': float(dt) if np.ndim(dt) == 0 else dt}


def _format_hallthruster_jl_input(thruster_inputs: dict, pem_to_julia: dict, thruster=None, config=None,
                                  simulation=None, postprocess=None, model_fidelity=(2, 2), output_path=None,
                                  fidelity_function=None) -> dict:
    """Build the `{'config', 'simulation', 'postprocess'}` dict HallThruster.run_simulation expects
    (thruster.py:184-278).  A `thruster` given as a str / Path is a device directory read with utils.load_thruster."""
    doc = {key: copy.deepcopy(val) if val is not None else {}
           for key, val in (('config', config), ('simulation', simulation), ('postprocess', postprocess))}
    if isinstance(thruster, (str, Path)):
        from ..utils import load_thruster
        thruster = load_thruster(thruster)
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
        # (a batch of runs shares one file: voltage and flow rate go into the name only when they are single numbers)
        volts, flow = doc['config'].get('discharge_voltage'), doc['config'].get('anode_mass_flow_rate')
        if np.ndim(volts) == 0 and not m.is_torch(volts) and volts:
            stem += f'_{round(volts)}V'
        if np.ndim(flow) == 0 and not m.is_torch(flow) and flow:
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


def run_analytic_double(json_input: dict | str | Path, jl_env=None, jl_script=None, **kwargs) -> dict:
    """Default `run_simulation` backend of `hallthruster_jl`: the reference's analytic stand-in for
    `HallThruster.run_simulation` (tests/sim_hallthruster.jl), evaluated on the GPU through `pem_thruster_f64[_dev]` and
    `pem_thruster_uion_f64_dev`.  Call signature of `run_hallthruster_jl` (thruster.py:281-375); `jl_env`, `jl_script`
    and the subprocess keyword arguments have nothing to act on and are ignored.

    Reads what the script reads (sim_hallthruster.jl:26-33): config.discharge_voltage, config.cathode_coupling_voltage,
    config.anode_mass_flow_rate, config.anom_model.model.c1, simulation.grid.num_cells, config.domain -- scalars for
    one run, arrays (numpy or CUDA tensors) of one loop shape for a batch of runs.  Returns the structure the script
    writes (sim_hallthruster.jl:50-66): `{'output': {'average': {...}}, 'config', 'simulation', 'postprocess'}`, with
    plain floats / lists for a single run (what json.load of the script's file gives) and arrays for a batch, and
    writes it to postprocess.output_file when that is set.  A TEST DOUBLE: no thruster physics."""
    if isinstance(json_input, dict):
        doc = json_input
    else:
        with open(json_input, 'r') as fp:
            doc = json.load(fp)
    if 'input' in doc and 'config' not in doc:        # an output file fed back in (thruster.py:309-310)
        doc = doc['input']
    try:
        cfg = doc['config']
        ins = {'V_a': cfg['discharge_voltage'], 'V_cc': cfg['cathode_coupling_voltage'], 'mdot_a': cfg['anode_mass_flow_rate'],
               'a_1': cfg['anom_model']['model']['c1']}
        ncells = int(doc['simulation']['grid']['num_cells'])
        domain = cfg['domain']
    except KeyError as e:
        raise KeyError(f'run_analytic_double: the input has no {e.args[0]!r} (sim_hallthruster.jl:26-31 reads '
                       f'config.discharge_voltage, .cathode_coupling_voltage, .anode_mass_flow_rate, .anom_model.model.c1, '
                       f'.domain and simulation.grid.num_cells)') from e
    single = all(np.ndim(v) == 0 for v in ins.values() if not m.is_torch(v)) and not any(m.is_torch(v) and v.dim() > 0 for v in ins.values())
    q = thruster_analytic(ins, num_cells=ncells, domain=(float(domain[0]), float(domain[1])))
    names = {'thrust': 'T', 'ion_current': 'I_B0', 'current_eff': 'eta_c', 'discharge_current': 'I_d', 'v_exh': 'v_exh',
             'mass_eff': 'eta_m', 'voltage_eff': 'eta_v', 'anode_eff': 'eta_a'}
    if single:
        host = {k: (v.cpu().numpy() if m.is_torch(v) else np.asarray(v)) for k, v in q.items()}
        avg = {jl: float(host[pem].reshape(-1)[0]) for jl, pem in names.items()}
        avg['ui'] = [host['u_ion'].reshape(-1, ncells)[0].tolist()]
        avg['z'] = host['u_ion_coords'].tolist()
    else:
        avg = {jl: q[pem] for jl, pem in names.items()}
        avg['ui'] = [q['u_ion']]
        avg['z'] = q['u_ion_coords']
    out = {'output': {'average': avg}, 'config': doc['config'], 'simulation': doc['simulation'], 'postprocess': doc.get('postprocess', {})}
    if target := doc.get('postprocess', {}).get('output_file'):
        def plain(o):
            if m.is_torch(o):
                o = o.detach().cpu().numpy()
            if isinstance(o, np.ndarray):
                return o.tolist()
            if isinstance(o, (np.floating, np.integer)):
                return o.item()
            if isinstance(o, Path):
                return str(o)
            raise TypeError(f'{type(o).__name__} is not JSON serialisable')
        Path(target).parent.mkdir(parents=True, exist_ok=True)
        with open(target, 'w', encoding='utf-8') as fp:
            json.dump(out, fp, ensure_ascii=False, indent=4, default=plain)
    return out


def hallthruster_jl(thruster_inputs: dict | None = None, thruster: Path | str | dict = 'SPT-100', config: dict | None = None,
                    simulation: dict | None = None, postprocess: dict | None = None, model_fidelity: tuple = (2, 2),
                    output_path: str | Path | None = None, version: str = HALLTHRUSTER_VERSION_DEFAULT,
                    pem_to_julia: dict | None = None, fidelity_function=None, julia_script: str | Path | None = None,
                    run_kwargs: dict | None = None, shock_threshold: float | None = None, run_simulation=None) -> dict:
    """The thruster component of the PEM (`hallmd.models.thruster.hallthruster_jl`, thruster.py:378-512;
    pem_v0_SPT-100.yml:63): the reference's signature, argument meaning, return keys and error behaviour.

    format the PEM inputs (`_format_hallthruster_jl_input`) -> `run_simulation(json_data, jl_env=, jl_script=,
    **run_kwargs)` -> `_convert_to_pem` -> ValueError for a non-physical run (thrust or beam current < 0) or, with
    `shock_threshold`, for an ion velocity that peaks before that axial position -> `model_cost` (s), `output_path`
    (relative to `output_path`, when one was given) and the raw `thruster_output`.

    `run_simulation` is the one addition to the reference's signature.  The reference always launches HallThruster.jl
    (`run_hallthruster_jl`, a Julia subprocess: out of scope here); the default here is `run_analytic_double`, the
    reference's own analytic stand-in for that run (tests/sim_hallthruster.jl, what tests/test_thruster.py:70-114 runs
    through `julia_script`), evaluated on the GPU.  Pass `hallmd.models.thruster.run_hallthruster_jl` to run the real
    solver.  `version` / `julia_script` / `run_kwargs` are handed to the backend as in the reference.

    Batches: the reference runs one simulation per call.  Here `thruster_inputs` may also hold arrays (numpy or CUDA
    tensors) of one loop shape; the QoIs then come back as arrays of that shape, `model_cost` is the wall time per
    sample, and the two filters, instead of raising, return the samples the reference would have raised for as NaN
    (all QoIs of that sample) with their messages in `errors` {flat index: message}."""
    mapping = copy.deepcopy(PEM_TO_JULIA)
    if pem_to_julia is not None:
        mapping.update(pem_to_julia)
    thruster_inputs = {} if thruster_inputs is None else thruster_inputs
    json_data = _format_hallthruster_jl_input(thruster_inputs, thruster=thruster, config=config, simulation=simulation,
                                              postprocess=postprocess, model_fidelity=model_fidelity, output_path=output_path,
                                              pem_to_julia=mapping, fidelity_function=fidelity_function)
    jl_env = None
    if version is not None:
        jl_env = Path('~/.julia/environments/').expanduser() / f'hallthruster_{version}'      # thruster.py:83-90
    backend = run_analytic_double if run_simulation is None else run_simulation
    kwargs = {'check': True} if run_kwargs is None else run_kwargs
    t1 = time.time()
    sim_results = backend(json_data, jl_env=jl_env, jl_script=julia_script, **kwargs)
    if any(m.is_torch(v) and v.is_cuda for v in sim_results.get('output', {}).get('average', {}).values()):
        import torch
        torch.cuda.synchronize()
    t2 = time.time()
    out = _convert_to_pem(sim_results, mapping)

    thrust, beam = out.get('T', 0), out.get('I_B0', 0)
    batch = np.ndim(thrust) > 0 or np.ndim(beam) > 0 or (m.is_torch(thrust) and thrust.dim() > 0)
    if not batch:
        if thrust < 0 or beam < 0:
            raise ValueError(f'Exception due to non-physical case: thrust={thrust} N, beam current={beam} A')
        if shock_threshold is not None:
            z, u = out.get('u_ion_coords'), out.get('u_ion')
            if z is not None and u is not None:
                if (z_max := z[int(np.argmax(u))]) < shock_threshold:
                    raise ValueError(f'Exception due to shock-like behavior: max ion velocity occurs at z={z_max:.3f} m')
        out['model_cost'] = t2 - t1
    else:
        bad = check_thruster_outputs(out, shock_threshold)
        bad_h = bad.cpu().numpy() if m.is_torch(bad) else np.asarray(bad)
        n = int(bad_h.size)
        if bad_h.any():
            nonphys = ((_to_host(thrust) < 0) | (_to_host(beam) < 0)).reshape(-1)
            out['errors'] = {int(i): ('Exception due to non-physical case' if nonphys[i] else 'Exception due to shock-like behavior')
                             for i in np.flatnonzero(bad_h.reshape(-1))}
            for key, val in out.items():
                if key in ('u_ion_coords', 'errors'):
                    continue
                if m.is_torch(val):
                    val[bad if val.dim() == bad.dim() else bad.unsqueeze(-1).expand_as(val)] = float('nan')
                elif isinstance(val, np.ndarray):
                    val[bad_h if val.ndim == bad_h.ndim else np.broadcast_to(bad_h[..., None], val.shape)] = np.nan
        out['model_cost'] = np.full(bad_h.shape, (t2 - t1) / max(n, 1))
    if output_path is not None:
        written = Path(json_data['postprocess'].get('output_file'))
        out['output_path'] = written.relative_to(Path(output_path).resolve()).as_posix()
    out['thruster_output'] = sim_results
    return out


def _to_host(x):
    return x.detach().cpu().numpy() if m.is_torch(x) else np.asarray(x, dtype=np.float64)
