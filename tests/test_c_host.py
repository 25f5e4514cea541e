"""The C ABI from a plain C program (examples/c_host.c): no Python and no PyTorch in the process, the library on the
system HIP runtime alone.  CPU: it compiles and links against include/pem_hip.h + libpem_hip.so.  GPU: it runs, and its
results equal the Python path's on the same inputs."""
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _build(tmp_path) -> Path:
    sys.path.insert(0, str(ROOT))
    from hallthrusterpem_amd import build
    lib = build.build()
    exe = tmp_path / 'c_host'
    cmd = ['gcc', '-O2', f'-I{ROOT / "include"}', str(ROOT / 'examples' / 'c_host.c'), f'-L{lib.parent}', '-lpem_hip',
           f'-Wl,-rpath,{lib.parent}', '-Wl,-rpath-link,/opt/rocm/lib', '-lm', '-o', str(exe)]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def test_c_host_compiles_and_links(tmp_path):
    assert _build(tmp_path).exists()


def _lcg_inputs(n):
    """The generator of examples/c_host.c, restated."""
    s = np.uint64(12345)
    u = np.empty((n, 15))
    a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    with np.errstate(over='ignore'):
        for i in range(n):
            for d in range(15):
                s = s * a + c
                u[i, d] = float(s >> np.uint64(11)) / 9007199254740992.0
    u = u.T
    return {'P_b': 10.0 ** (-8.0 + 4.0 * u[0]), 'V_a': 200.0 + 200.0 * u[1], 'T_e': 1.0 + 4.0 * u[2], 'V_vac': 60.0 * u[3],
            'Pstar': 1e-5 + 9e-5 * u[4], 'P_T': 1e-5 + 9e-5 * u[5], 'mdot_a': 2e-6 + 5e-6 * u[6],
            'a_1': 10.0 ** (-2.5 + 1.5 * u[7]), 'c0': u[8], 'c1': 0.1 + 0.8 * u[9], 'c2': -15.0 + 30.0 * u[10],
            'c3': 0.2 + 1.370796 * u[11], 'c4': 10.0 ** (18.0 + 4.0 * u[12]), 'c5': 10.0 ** (14.0 + 4.0 * u[13]),
            'sigma_cex': 51e-20 + 7e-20 * u[14]}


@pytest.mark.gpu
def test_c_host_runs_and_matches_the_python_path(tmp_path):
    n = 3000
    out = subprocess.run([str(_build(tmp_path)), str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    vals = {k: float(v) for k, v in re.findall(r'(\w+(?:\[\d+\])?)=([-+0-9.eE]+|nan|inf)', out.stdout)}
    from hallthrusterpem_amd.models import pem_v0_coupled
    x = _lcg_inputs(n)
    ref = pem_v0_coupled(x)
    # libm's pow() in the C program and numpy's 10.0 ** x may differ in the last bit of an input: compare to 1e-9
    assert vals['n'] == n and vals['invalid'] == int(ref['invalid'].sum())
    assert np.isclose(vals['sum_V_cc'], ref['V_cc'].sum(), rtol=1e-9) and np.isclose(vals['sum_div'], ref['div_angle'].sum(), rtol=1e-9)
    assert np.isclose(vals['sum_j_ion'], ref['j_ion'].sum(), rtol=1e-9)
    assert np.isclose(vals['V_cc'], ref['V_cc'][0], rtol=1e-9) and np.isclose(vals['div_angle'], ref['div_angle'][0], rtol=1e-9)
    assert np.isclose(vals['j_ion[0]'], ref['j_ion'][0, 0], rtol=1e-9) and np.isclose(vals['j_ion[90]'], ref['j_ion'][0, 90], rtol=1e-9)
