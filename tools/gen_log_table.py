#!/usr/bin/env python3
"""Generate hallthrusterpem_amd/csrc/pem_log_table.h: the table behind pem::pem_log10_tab (csrc/pem_math.h), the fp64
log10 of the `norm: log10` of j_ion (pem_v0_SPT-100.yml:273-280) in the SVD-compression kernels.

Method (Tang-style table look-up instead of the division + degree-7 series of pem_log10):
  x = 2^e m, m in [1/2, 1) from v_frexp;  i = top 10 bits of m's mantissa;  entry i = {c_i, T_i}:
      r = fma(m, c_i, -1)            c_i ~ 1 / (centre of interval i), so |r| <= 2^-11 (2^-10 in entry 0)
      log10(x) = (e - low_i) log10(2) + T_i + log1p(r) / ln(10)
  with low_i = [i < 424] (m < ~sqrt(1/2): the interval is then read as part of [1, sqrt 2) of the next binade down, so
  that results near x = 1 never come from cancelling e log10(2) against a table value), T_i = -log10(c_i) (+ log10(2)
  for low_i) computed in 40 digits FROM THE ROUNDED c_i, so that the identity above is exact up to the polynomial (and c_i
  picked, Gal-style, among the doubles within 600 ulp of 1/centre so that T_i is a double to within 1e-3 ulp), and the
  two entries next to x = 1 pinned to c = 2 (i = 0) and c = 1 (i = 1023) with T = 0: there r = x - 1 exactly and the
  result keeps full relative accuracy.  log1p(r)/ln 10 = r (a1 + r (a2 + ... + r a6)), a_k = (-1)^(k+1) / (k ln 10):
  truncation r^7 / 7 < 2^-62 |r|.

Self-check: compiles the same algorithm in C (libm fma = the device's v_fma_f64) and compares it with libm's log10 on
2e7 arguments (log-uniform over the whole range, denormals, a dense sweep around 1, every interval edge): the largest
error against log10l (x87 extended, 64-bit mantissa) is printed and must be <= 1.5 ulp (libm's own is printed beside it).

Run:  python tools/gen_log_table.py   (rewrites the header; the header is committed)
"""
import subprocess
import sys
import tempfile
from pathlib import Path

import mpmath as mp
import numpy as np

ROOT = Path(__file__).resolve().parents[1]
OUT = ROOT / 'hallthrusterpem_amd' / 'csrc' / 'pem_log_table.h'
NBITS = 10
N = 1 << NBITS
LOW_BELOW = 424          # entries i < 424 cover m < 0.70703125: read as 2m in [1, 1.4140625)

CHECK_C = r'''
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#define PEM_TABLE_DECL static const
#include "pem_log_table.h"
static double tab_log10(double x) {
    int e; double m = frexp(x, &e);                      /* m in [1/2, 1), denormals normalised */
    uint64_t b; memcpy(&b, &m, 8);
    const int i = (int)((b >> (52 - PEM_LOG_NBITS)) & (PEM_LOG_N - 1));
    e -= i < PEM_LOG_LOW_BELOW;
    const double c = PEM_LOG10_TAB[2 * i], T = PEM_LOG10_TAB[2 * i + 1];
    const double r = fma(m, c, -1.0);
    double p = PEM_LOG_A6;
    p = fma(p, r, PEM_LOG_A5); p = fma(p, r, PEM_LOG_A4); p = fma(p, r, PEM_LOG_A3); p = fma(p, r, PEM_LOG_A2); p = fma(p, r, PEM_LOG_A1);
    const double de = (double)e;
    return fma(de, 3.01029995663611771306e-01, T) + fma(de, 3.69423907715893078616e-13, p * r);
}
static double ulps(double a, long double b) {             /* distance from the 64-bit-mantissa value, in ulp of its double */
    int ex; frexp((double)b, &ex);
    return (double)(fabsl((long double)a - b) / ldexpl(1.0L, ex - 53));
}
int main(void) {
    double worst = 0, at = 0, libm = 0; uint64_t s = 88172645463325252ull; long n = 0;
    #define TRY(v) do { double x_ = (v); long double t_ = log10l((long double)x_); double u_ = ulps(tab_log10(x_), t_); double v_ = ulps(log10(x_), t_); ++n; if (u_ > worst) { worst = u_; at = x_; } if (v_ > libm) libm = v_; } while (0)
    for (long k = 0; k < 12000000; ++k) {                /* log-uniform over all positive doubles incl. denormals */
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t b = s & 0x7fefffffffffffffull; double x; memcpy(&x, &b, 8);
        if (x > 0) TRY(x);
    }
    for (long k = -3000000; k <= 3000000; ++k) TRY(1.0 + k * 1.1102230246251565e-16 * (1 + (k & 1023)));   /* around 1 */
    for (long k = 0; k < 3000000; ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; TRY(0.25 + (double)(s >> 11) * 0x1p-53 * 3.75); }
    for (int i = 0; i <= PEM_LOG_N; ++i)                  /* both sides of every interval edge, three binades */
        for (int sh = -1; sh <= 1; ++sh) {
            const double edge = ldexp(0.5 + 0.5 * i / PEM_LOG_N, sh);
            TRY(edge); TRY(nextafter(edge, 0)); TRY(nextafter(edge, 4));
        }
    TRY(1e-20); TRY(5e-324); TRY(1.7976931348623157e308); TRY(2.2250738585072014e-308);
    printf("%ld arguments: worst error against log10l (64-bit mantissa) %.3f ulp at x = %a; libm's own log10 on the same arguments: %.3f ulp\n", n, worst, at, libm);
    return worst <= 1.5 ? 0 : 1;
}
'''


def main():
    mp.mp.dps = 40
    rows = []
    worst_miss = 0.0
    for i in range(N):
        centre = mp.mpf(1) / 2 * (1 + (mp.mpf(i) + mp.mpf(1) / 2) / N)        # in m units, m in [1/2, 1)
        low = i < LOW_BELOW
        if i == 0:
            c = 2.0
        elif i == N - 1:
            c = 1.0
        else:
            # Gal's accurate tables: among the doubles within 600 ulp of 1/centre (r barely moves) take the one whose
            # T_i lies closest to a double -- within 2^-9 ulp or better -- so that rounding T_i costs nothing
            c0 = float(1 / centre)
            best = (1.0, c0)
            c = float(np.nextafter(c0, 0.0))
            cands = [c0]
            lo_c, hi_c = c0, c0
            for _ in range(600):
                lo_c, hi_c = float(np.nextafter(lo_c, 0.0)), float(np.nextafter(hi_c, 4.0))
                cands += [lo_c, hi_c]
            for cand in cands:
                Tm = -mp.log10(mp.mpf(cand)) + (mp.log10(2) if low else 0)
                Tf = float(Tm)
                miss = abs(Tm - mp.mpf(Tf)) / mp.mpf(float(np.spacing(abs(Tf))))
                if miss < best[0]:
                    best = (float(miss), cand)
                    if miss < 1.0 / 1024:
                        break
            worst_miss = max(worst_miss, best[0])
            c = best[1]
        T = -mp.log10(mp.mpf(c)) + (mp.log10(2) if low else 0)
        if i in (0, N - 1):
            assert T == 0
        rows.append((c, float(T)))
    ln10 = mp.log(10)
    coef = [float((-1) ** (k + 1) / (k * ln10)) for k in range(1, 7)]
    lines = ['// pem_log_table.h -- GENERATED by tools/gen_log_table.py; do not edit.',
             '// {c_i, T_i} of pem::pem_log10_tab (csrc/pem_math.h): r = fma(m, c_i, -1), log10 x = (e - [i < LOW_BELOW]) log10 2 + T_i + r P(r).',
             '#pragma once',
             '#ifndef PEM_TABLE_DECL',
             '#define PEM_TABLE_DECL static const',
             '#endif',
             f'#define PEM_LOG_NBITS {NBITS}',
             f'#define PEM_LOG_N {N}',
             f'#define PEM_LOG_LOW_BELOW {LOW_BELOW}']
    for k, a in enumerate(coef, 1):
        lines.append(f'#define PEM_LOG_A{k} {a.hex()}   /* {a!r} = (-1)^{k + 1} / ({k} ln 10) */')
    lines.append(f'PEM_TABLE_DECL double PEM_LOG10_TAB[{2 * N}] = {{')
    for i in range(0, N, 2):
        lines.append('    ' + ', '.join(f'{c.hex()}, {t.hex()}' for c, t in rows[i:i + 2]) + ',')
    lines.append('};')
    OUT.write_text('\n'.join(lines) + '\n')
    print(f'wrote {OUT} ({N} entries, {16 * N} bytes); every T_i within {worst_miss:.1e} ulp of its double')
    with tempfile.TemporaryDirectory() as tmp:
        src = Path(tmp) / 'check.c'
        src.write_text(CHECK_C)
        exe = Path(tmp) / 'check'
        subprocess.run(['gcc', '-O2', '-ffp-contract=off', f'-I{OUT.parent}', str(src), '-o', str(exe), '-lm'], check=True)
        rc = subprocess.run([str(exe)]).returncode
    if rc != 0:
        print('SELF-CHECK FAILED: more than 1.5 ulp from log10l')
    return rc


if __name__ == '__main__':
    sys.exit(main())
