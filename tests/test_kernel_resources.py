"""The headline kernel's static resources, checked where it is compiled (no GPU needed: hipcc cross-compiles gfx950).

`plume_r1_kernel<4, true, 1, false>` -- the coupled fp64-profile launch bench.py times -- is at 244 vector registers with no
scratch; two waves per SIMD need it at or below 256, and any scratch would put spill traffic into a kernel whose bound is
HBM writes.  An edit that pushes it over either line fails here, before it reaches the GPU box.
The full table of every kernel is profiles/kernel_resources_r04.md (tools/kernel_stats.py)."""
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.skipif(shutil.which('hipcc') is None and not Path('/opt/rocm/bin/hipcc').exists(), reason='hipcc not available')
def test_headline_kernel_fits_two_waves_per_simd_without_scratch():
    out = subprocess.run([sys.executable, str(ROOT / 'tools' / 'kernel_stats.py'), str(ROOT / 'hallthrusterpem_amd' / 'csrc' / 'pem_kernels.hip'),
                          ], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = {}
    for line in out.stdout.splitlines():
        m = re.match(r'(\w+(?:<[^>]*>)?)\s+vgpr\s+(\d+) sgpr\s+(\d+) s-spill\s+(\d+) v-spill\s+(\d+) scratch\s+(\d+)', line)
        if m:
            rows[m.group(1)] = dict(vgpr=int(m.group(2)), vspill=int(m.group(5)), scratch=int(m.group(6)))
    head = rows.get('plume_r1_kernel<4, true, 1, false, 0, false>')
    assert head is not None, sorted(rows)
    assert head['vgpr'] <= 256 and head['vspill'] == 0 and head['scratch'] == 0, head
    # round 4: no kernel of this translation unit -- every plume_r* form, the fused Monte-Carlo modes, the
    # counting launches of the fused campaign statistics -- spills a vector register or touches scratch
    assert len(rows) > 40, len(rows)
    bad = {name: r for name, r in rows.items()
           if r['vspill'] or r['scratch'] or (name.startswith('plume_r1_kernel<4, true') and r['vgpr'] > 256)}
    assert not bad, bad
