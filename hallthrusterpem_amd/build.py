"""Compile libpem_hip.so for gfx950 in-tree with hipcc (no GPU needed to build)."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SRCS = [PKG / 'csrc' / 'pem_kernels.hip', PKG / 'csrc' / 'pem_sampler.hip', PKG / 'csrc' / 'pem_svd.hip', PKG / 'csrc' / 'pem_likelihood.hip', PKG / 'csrc' / 'pem_surrogate.hip']
LIB = PKG / 'libpem_hip.so'
DEPS = SRCS + [PKG / 'csrc' / 'pem_tables.h', PKG / 'csrc' / 'pem_common.h', PKG / 'csrc' / 'pem_philox.h', ROOT / 'include' / 'pem_hip.h']


def hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not Path(exe).exists():
        raise RuntimeError('hipcc not found: libpem_hip.so cannot be built (ROCm toolchain required)')
    return exe


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(d.exists() and d.stat().st_mtime > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -shared -> hallthrusterpem_amd/libpem_hip.so"""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           f'-I{ROOT / "include"}', f'-I{PKG / "csrc"}', *[str(s) for s in SRCS], '-o', str(LIB)]
    if verbose:
        print(' '.join(cmd))
    env = dict(os.environ)
    subprocess.run(cmd, check=True, env=env)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
