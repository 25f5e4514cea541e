"""ctypes binding of liboracle.so (oracle/pem_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- the
product package `hallthrusterpem_amd` never imports this module (tests/test_boundary.py checks).
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
NANGLE = 91
_lib = None


def build(force: bool = False) -> Path:
    so = HERE / 'liboracle.so'
    src = HERE / 'pem_oracle.c'
    if force or not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
        subprocess.run(['make', '-C', str(HERE), '-s', '-B', 'liboracle.so'], check=True)
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.oracle_normaliser.restype = C.c_double
        _lib.oracle_normaliser.argtypes = [C.c_double]
        _lib.oracle_model_fidelity.restype = C.c_double
        _lib.oracle_angle_grid.restype = C.POINTER(C.c_double)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if n is not None and a.shape != (n,):
        a = np.ascontiguousarray(np.broadcast_to(a, (n,)))
    return a


def set_threads(n: int) -> int:
    return int(lib().oracle_set_threads(C.c_int(n)))


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def angle_grid():
    return np.ctypeslib.as_array(lib().oracle_angle_grid(), shape=(NANGLE,)).copy()


def normaliser(alpha):
    a = np.atleast_1d(np.asarray(alpha, dtype=np.float64))
    return np.array([lib().oracle_normaliser(C.c_double(v)) for v in a.ravel()]).reshape(a.shape)


def cathode(P_b, V_a, T_e, V_vac, Pstar, P_T, torr2pa):
    n = np.broadcast(P_b, V_a, T_e, V_vac, Pstar, P_T).size
    arrs = [_f64(x, n) for x in (P_b, V_a, T_e, V_vac, Pstar, P_T)]
    out = np.empty(n)
    rc = lib().oracle_cathode_f64(C.c_long(n), *[_p(a) for a in arrs], C.c_double(torr2pa), _p(out))
    assert rc == 0
    return out


def plume(P_b, c0, c1, c2, c3, c4, c5, sigma_cex, I_B0, torr2pa, T=None, radii=(1.0,)):
    ins = (P_b, c0, c1, c2, c3, c4, c5, sigma_cex, I_B0)
    n = np.broadcast(*ins).size
    arrs = [_f64(x, n) for x in ins]
    rad = _f64(np.atleast_1d(radii))
    R = rad.size
    Tarr = None if T is None else _f64(T, n)
    j = np.empty((n, NANGLE, R))
    div = np.empty((n, R))
    tc = np.empty((n, R)) if T is not None else None
    inv = np.zeros(n, dtype=np.uint8)
    rc = lib().oracle_plume_f64(C.c_long(n), C.c_int(R), _p(rad), C.c_double(torr2pa), *[_p(a) for a in arrs],
                                _p(Tarr), _p(j), _p(div), _p(tc), _p(inv))
    assert rc == 0
    return {'j_ion': j, 'div_angle': div, 'T_c': tc, 'invalid': inv.astype(bool)}


def plume_terms(P_b, c0, c1, c2, c3, c4, c5, sigma_cex, I_B0, torr2pa, radii=(1.0,)):
    """Sizes of the terms behind each plume result (oracle_plume_terms_f64): what tests/parity_rules.py turns into
    per-entry tolerances.  Returns arrays of shape (n, R) -- X1, X2, j_cex, decay, den, num, den_abs, num_abs -- and the
    beam widths a1, a2 of shape (n,)."""
    ins = (P_b, c0, c1, c2, c3, c4, c5, sigma_cex, I_B0)
    n = np.broadcast(*ins).size
    arrs = [_f64(x, n) for x in ins]
    rad = _f64(np.atleast_1d(radii))
    t = np.empty((n, rad.size, 8))
    al = np.empty((n, 2))
    rc = lib().oracle_plume_terms_f64(C.c_long(n), C.c_int(rad.size), _p(rad), C.c_double(torr2pa), *[_p(a) for a in arrs], _p(t), _p(al))
    assert rc == 0
    out = {k: t[:, :, i] for i, k in enumerate(('X1', 'X2', 'j_cex', 'decay', 'den', 'num', 'den_abs', 'num_abs'))}
    out['a1'], out['a2'], out['radii'] = al[:, 0], al[:, 1], rad
    return out


def thruster(V_a, V_cc, mdot_a, a_1):
    n = np.broadcast(V_a, V_cc, mdot_a, a_1).size
    arrs = [_f64(x, n) for x in (V_a, V_cc, mdot_a, a_1)]
    names = ['I_B0', 'I_d', 'T', 'eta_c', 'eta_m', 'eta_v', 'eta_a', 'v_exh']
    outs = {k: np.empty(n) for k in names}
    rc = lib().oracle_thruster_f64(C.c_long(n), *[_p(a) for a in arrs], *[_p(outs[k]) for k in names])
    assert rc == 0
    return outs


def thruster_uion(v_exh, z0, z1, ncells):
    v = _f64(np.atleast_1d(v_exh))
    z = np.empty(ncells)
    u = np.empty((v.size, ncells))
    rc = lib().oracle_thruster_uion_f64(C.c_long(v.size), _p(v), C.c_double(z0), C.c_double(z1), C.c_int(ncells),
                                        _p(z), _p(u))
    assert rc == 0
    return z, u


COUPLED_INPUTS = ('P_b', 'V_a', 'T_e', 'V_vac', 'Pstar', 'P_T', 'mdot_a', 'a_1',
                  'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')


def coupled(inputs: dict, torr2pa, radius=1.0):
    n = np.broadcast(*[inputs[k] for k in COUPLED_INPUTS]).size
    arrs = [_f64(inputs[k], n) for k in COUPLED_INPUTS]
    out = {'V_cc': np.empty(n), 'I_B0': np.empty(n), 'T': np.empty(n), 'j_ion': np.empty((n, NANGLE)),
           'div_angle': np.empty(n), 'T_c': np.empty(n)}
    inv = np.zeros(n, dtype=np.uint8)
    rc = lib().oracle_coupled_f64(C.c_long(n), C.c_double(torr2pa), C.c_double(radius), *[_p(a) for a in arrs],
                                  _p(out['V_cc']), _p(out['I_B0']), _p(out['T']), _p(out['j_ion']),
                                  _p(out['div_angle']), _p(out['T_c']), _p(inv))
    assert rc == 0
    out['invalid'] = inv.astype(bool)
    return out


def model_fidelity(f0, f1, domain_hi, anode_pot, cathode_pot, mol_weight, avogadro, charge, cfl=0.2):
    nc, nq = C.c_int(), C.c_int()
    dt = lib().oracle_model_fidelity(C.c_int(f0), C.c_int(f1), C.c_double(domain_hi), C.c_double(anode_pot),
                                     C.c_double(cathode_pot), C.c_double(mol_weight), C.c_double(avogadro),
                                     C.c_double(charge), C.c_double(cfl), C.byref(nc), C.byref(nq))
    return {'num_cells': nc.value, 'ncharge': nq.value, 'dt': float(dt)}
