"""Compile libpem_hip.so for gfx950 in-tree with hipcc (no GPU needed to build)."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SRCS = [PKG / 'csrc' / 'pem_kernels.hip', PKG / 'csrc' / 'pem_sampler.hip', PKG / 'csrc' / 'pem_svd.hip', PKG / 'csrc' / 'pem_likelihood.hip', PKG / 'csrc' / 'pem_surrogate.hip', PKG / 'csrc' / 'pem_fp32.hip', PKG / 'csrc' / 'pem_saltelli.hip', PKG / 'csrc' / 'pem_latent.hip', PKG / 'csrc' / 'pem_quantile.hip']
LIB = PKG / 'libpem_hip.so'
DEPS = SRCS + sorted((PKG / 'csrc').glob('*.h')) + [ROOT / 'include' / 'pem_hip.h']


def hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not Path(exe).exists():
        raise RuntimeError('hipcc not found: libpem_hip.so cannot be built (ROCm toolchain required)')
    return exe


def have_hipcc() -> bool:
    return Path(shutil.which('hipcc') or '/opt/rocm/bin/hipcc').exists()


STAMP = PKG / 'libpem_hip.so.srchash'


def source_hash() -> str:
    """Digest of every file the library is compiled from (content, not mtime: a snapshot copy may not keep times)."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(d.name.encode())
        h.update(d.read_bytes() if d.exists() else b'<missing>')
    return h.hexdigest()


def needs_build() -> bool:
    """True when the library is missing or was built from other sources than the ones in the tree."""
    if not LIB.exists() or not STAMP.exists():
        return True
    return STAMP.read_text().strip() != source_hash()


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
    STAMP.write_text(source_hash() + '\n')
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
