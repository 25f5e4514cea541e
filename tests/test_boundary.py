"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/pem_hip.h declares, the product never touches the oracle, and it fails loudly without a GPU."""
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / 'include' / 'pem_hip.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pem_[a-z0-9_]+)\s*\(', text)))


def test_header_and_library_agree():
    from hallthrusterpem_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    assert sorted(_lib.SIGNATURES) == declared          # the ctypes table binds exactly the header
    for name in declared:
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.pem_version()
    grid = np.ctypeslib.as_array(lib.pem_angle_grid(), shape=(91,))
    assert np.array_equal(grid, np.linspace(0, np.pi / 2, 91))        # plume.py:53, bit for bit


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under hallthrusterpem_amd/ may import, load or link it."""
    for path in (ROOT / 'hallthrusterpem_amd').rglob('*'):
        if path.suffix not in ('.py', '.hip', '.h', '.cpp'):
            continue
        src = path.read_text()
        assert not re.search(r'^\s*(import|from)\s+oracle\b', src, flags=re.M), path
        assert 'liboracle' not in src and 'oracle_ctypes' not in src and 'pem_oracle' not in src, path


def test_no_gpu_means_loud_failure():
    from hallthrusterpem_amd import _lib
    from hallthrusterpem_amd.models import cathode_coupling, current_density, pem_v0_coupled, thruster_analytic
    if _lib.device_count() > 0:
        pytest.skip('a HIP device is present')
    one = {'P_b': 1e-5, 'V_a': 300., 'T_e': 3., 'V_vac': 30., 'Pstar': 2e-5, 'P_T': 5e-5, 'mdot_a': 5e-6, 'a_1': 0.01,
           'c0': 0.5, 'c1': 0.5, 'c2': -8., 'c3': 0.3, 'c4': 1e20, 'c5': 1e16, 'sigma_cex': 55e-20, 'I_B0': 3., 'V_cc': 30.}
    for fn in (cathode_coupling, current_density, pem_v0_coupled, thruster_analytic):
        with pytest.raises(_lib.PemHipError) as ei:
            fn(one)
        assert ei.value.code == _lib.PEM_ERR_NO_DEVICE
    with pytest.raises(_lib.PemHipError):
        _lib.require_device()


def test_loop_shape_rules():
    from hallthrusterpem_amd import _marshal as m
    assert m.loop_shape([1.0, 2.0]) == (1,)                            # all-scalar call -> leading axis of 1
    assert m.loop_shape([np.zeros((3, 4)), 2.0, np.zeros(4)]) == (3, 4)
    assert m.loop_shape([np.zeros(0), 1.0]) == (0,)
    a = m.host_flat(2.5, (2, 3))
    assert a.shape == (6,) and a.flags.c_contiguous and np.all(a == 2.5)
    with pytest.raises(ValueError):
        m.loop_shape([np.zeros(3), np.zeros(4)])
