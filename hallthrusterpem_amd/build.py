"""Compile libpem_hip.so for gfx950 in-tree with hipcc (no GPU needed to build)."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
SRCS = [PKG / 'csrc' / 'pem_kernels.hip', PKG / 'csrc' / 'pem_sampler.hip', PKG / 'csrc' / 'pem_svd.hip', PKG / 'csrc' / 'pem_likelihood.hip', PKG / 'csrc' / 'pem_surrogate.hip', PKG / 'csrc' / 'pem_fp32.hip', PKG / 'csrc' / 'pem_saltelli.hip', PKG / 'csrc' / 'pem_latent.hip', PKG / 'csrc' / 'pem_quantile.hip', PKG / 'csrc' / 'pem_masks.hip']
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


def _object_hash(src: Path, flags) -> str:
    """Digest of everything one object file depends on: its source, every header, the flags."""
    import hashlib
    h = hashlib.sha256()
    h.update(' '.join(flags).encode())
    for d in [src] + [d for d in DEPS if d.suffix == '.h']:
        h.update(d.name.encode())
        h.update(d.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 -> hallthrusterpem_amd/libpem_hip.so.

    One object per translation unit, compiled in parallel and kept under hallthrusterpem_amd/build/ keyed by the hash of
    its source + headers + flags (an edit of one kernel file recompiles that file only; `force` recompiles everything),
    then one link.  The whole build runs under an exclusive file lock and the library and its stamp are moved into place
    atomically, so that N ranks started after a source edit (bench.py --gpus N, the mp.spawn tests) build once and never
    load a half-written library."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    if not force and not needs_build():
        return LIB
    objdir = PKG / 'build'
    objdir.mkdir(exist_ok=True)
    with open(objdir / '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():          # another process built it while this one waited for the lock
            return LIB
        flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', f'-I{ROOT / "include"}', f'-I{PKG / "csrc"}']
        cc = hipcc()
        env = dict(os.environ)

        def compile_one(src: Path) -> Path:
            obj = objdir / (src.stem + '.o')
            stamp = objdir / (src.stem + '.o.hash')
            digest = _object_hash(src, flags)
            if not force and obj.exists() and stamp.exists() and stamp.read_text().strip() == digest:
                return obj
            cmd = [cc, *flags, '-c', str(src), '-o', str(obj)]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True, env=env)
            stamp.write_text(digest + '\n')
            return obj

        workers = max(1, min(len(SRCS), int(os.environ.get('PEM_BUILD_JOBS', os.cpu_count() or 1))))
        with ThreadPoolExecutor(workers) as pool:
            objs = list(pool.map(compile_one, SRCS))
        tmp_lib = objdir / f'libpem_hip.so.{os.getpid()}'
        cmd = [cc, '--offload-arch=gfx950', '-fPIC', '-shared', *[str(o) for o in objs], '-o', str(tmp_lib)]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True, env=env)
        tmp_stamp = objdir / f'srchash.{os.getpid()}'
        tmp_stamp.write_text(source_hash() + '\n')
        os.replace(tmp_lib, LIB)
        os.replace(tmp_stamp, STAMP)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
