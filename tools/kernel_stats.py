#!/usr/bin/env python3
"""Register / spill / LDS statistics of every gfx950 kernel in the .hip sources (no GPU needed).

    python tools/kernel_stats.py [file.hip ...] [--grep plume_r1] [--insts]

Compiles each source with `hipcc --offload-arch=gfx950 -S` semantics (-save-temps into a scratch directory), reads the
amdhsa metadata of the device assembly and prints one line per kernel: VGPRs, SGPRs, SGPR / VGPR spills, scratch bytes,
kernel-argument bytes, static LDS.  --insts adds static instruction counts (VALU / SALU / DS / VMEM / v_readlane+v_writelane).
"""
import argparse
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / 'hallthrusterpem_amd' / 'csrc'


def demangle(name: str) -> str:
    exe = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
    if not exe:
        return name
    out = subprocess.run([exe, name], capture_output=True, text=True).stdout.strip()
    out = re.sub(r'^void ', '', out)
    return re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', out).replace('(anonymous namespace)::', '')


def assembly(src: Path, extra) -> str:
    tmp = Path(tempfile.mkdtemp(prefix='kstats_'))
    try:
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', f'-I{ROOT / "include"}', f'-I{CSRC}',
               str(src), '-o', str(tmp / 'k.o'), '-save-temps=obj', *extra]
        subprocess.run(cmd, check=True, capture_output=True)
        return next(tmp.glob('*gfx950.s')).read_text()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def inst_counts(asm: str):
    """static instruction mix per kernel symbol"""
    out, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r'^(\w+):\s*(;.*)?$', line)
        if m and not line.startswith('.'):
            cur = m.group(1)
            out[cur] = dict(valu=0, salu=0, ds=0, vmem=0, lane=0, mfma=0, trans=0)
            continue
        if cur is None:
            continue
        t = line.strip().split(' ')[0]
        c = out[cur]
        if t.startswith(('v_readlane', 'v_writelane')):
            c['lane'] += 1
        elif t.startswith('v_mfma'):
            c['mfma'] += 1
        elif t.startswith('v_'):
            c['valu'] += 1
            if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)_', t):
                c['trans'] += 1
        elif t.startswith('s_') and not t.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_endpgm')):
            c['salu'] += 1
        elif t.startswith('ds_'):
            c['ds'] += 1
        elif t.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            c['vmem'] += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('files', nargs='*')
    ap.add_argument('--grep', default='')
    ap.add_argument('--insts', action='store_true')
    ap.add_argument('-D', action='append', default=[])
    args = ap.parse_args()
    files = [Path(f) for f in args.files] or sorted(CSRC.glob('*.hip'))
    for src in files:
        asm = assembly(src, [f'-D{d}' for d in args.D])
        counts = inst_counts(asm) if args.insts else {}
        md = asm[asm.index('amdhsa.kernels:'):]
        print(f'== {src.name}')
        for b in re.split(r'\n  - \.agpr_count', md)[1:]:
            sym = re.search(r'\.name:\s+(\S+)', b).group(1)
            name = demangle(sym)
            if args.grep and args.grep not in name:
                continue
            g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, b).group(1))
            line = (f'{name[:78]:78s} vgpr {g("vgpr_count"):3d} sgpr {g("sgpr_count"):3d} s-spill {g("sgpr_spill_count"):3d} '
                    f'v-spill {g("vgpr_spill_count"):3d} scratch {g("private_segment_fixed_size"):4d} kernarg {g("kernarg_segment_size"):4d} '
                    f'lds {g("group_segment_fixed_size"):6d}')
            if args.insts and sym in counts:
                c = counts[sym]
                line += f' | valu {c["valu"]} (trans {c["trans"]}) salu {c["salu"]} ds {c["ds"]} vmem {c["vmem"]} lane-moves {c["lane"]} mfma {c["mfma"]}'
            print(line)


if __name__ == '__main__':
    sys.exit(main())
