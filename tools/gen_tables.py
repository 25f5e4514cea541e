#!/usr/bin/env python3
"""Generate hallthrusterpem_amd/csrc/pem_tables.h: the constant tables the HIP kernels stage in LDS.

The beam normaliser of the plume model is D(a) = 2 pi Int_0^{pi/2} exp(-(t/a)^2) sin t dt, which is
identically the complex-erfi bracket of the reference (src/hallmd/models/plume.py:64-85; SURVEY.md
section 0).  Two tables evaluate it without any complex arithmetic:

  * DPOLY     As a function of u = 1/a^2, D(u) = 2 pi Int_0^{pi/2} exp(-u t^2) sin t dt is entire.  On
              u in [0, 16] (|a| >= 0.25) it is tabulated as 32 polynomials of degree 11 on intervals of
              width 1/2:  D(u) = sum_j DPOLY[i][j] x^j,  i = floor(2u),  x = 4u - 2i - 1 in [-1, 1]
              (Chebyshev interpolants at 50 digits, converted to monomials).
  * DAWSON    S(y) = sum_n (-y)^n / (2n+1)!!, so that D(a) = pi a^2 S(a^2/2) for |a| < 0.25 (there the
              part of the integral beyond pi/2 is < 1e-18 of D).

  * SIMPSON   per angle index k (plume.py:53 grid, alpha_k = k*(pi/2)/90) the folded weights of the
              divergence integrals of plume.py:117-123 -- the profile is flipped there, so angle k sits at
              grid point m = 90-k:  CDEN[k] = w_m cos(x_m), CNUM[k] = w_m cos(x_m) sin(x_m), w_m = the
              composite-Simpson weight scipy.integrate.simpson(x=alpha_rad) gives point m (irregular-
              spacing formula, evaluated in the same double arithmetic).

  * QPOLY     the two divergence integrals as FUNCTIONS of the beam width.  With f_k = X1 e_k(a1) + X2 e_k(a2),
              e_k(a) = exp(-(k h / a)^2), the Simpson sums are den = X1 Qd(a1) + X2 Qd(a2), num = X1 Qn(a1) + X2 Qn(a2)
              with Qd(a) = sum_k CDEN[k] e_k(a), Qn(a) = sum_k CNUM[k] e_k(a): when no profile is wanted the 91-term
              loop is two table look-ups per beam.  Region A, u = 1/a^2 in [0, 16] like DPOLY (32 intervals);
              region B, |a| in [QA_MIN, 0.25), 64 equal intervals in |a|; degree 11 both; coefficients of Qd and Qn
              interleaved: QPOLY[(interval*NDC + j)*2 + {0, 1}].  Below QA_MIN the kernel runs the loop.

Run:  python tools/gen_tables.py   (rewrites the header, ~40 s; the header is committed)
"""
from pathlib import Path

import mpmath as mp
import numpy as np

OUT = Path(__file__).resolve().parents[1] / 'hallthrusterpem_amd' / 'csrc' / 'pem_tables.h'
NANGLE = 91
NDI = 32      # intervals of width 1/2 in u = 1/alpha^2 covering [0, 16]
NDC = 12      # coefficients per interval (degree 11)
NDAW = 10
NQB = 64          # region-B intervals of QPOLY
QA_MIN = 0.03     # below this beam width the Simpson functionals are summed term by term


def hexd(x):
    return float(x).hex()


def d_of_u(u):
    return 2 * mp.pi * mp.quad(lambda t: mp.exp(-u * t * t) * mp.sin(t), mp.linspace(0, mp.pi / 2, 17))


def dpoly_tables():
    """Chebyshev interpolation of D(u) on [i/2, (i+1)/2] at NDC nodes, as monomial coefficients in x."""
    mp.mp.dps = 50
    T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]                    # Chebyshev T_j as monomial coefficient lists
    for j in range(2, NDC):
        cur = [mp.mpf(0)] + [2 * c for c in T[j - 1]]
        for k, c in enumerate(T[j - 2]):
            cur[k] -= c
        T.append(cur)
    nodes = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / NDC) for k in range(NDC)]
    rows = []
    for i in range(NDI):
        mid, half = (mp.mpf(i) + mp.mpf(1) / 2) / 2, mp.mpf(1) / 4
        vals = [d_of_u(mid + half * x) for x in nodes]
        cheb = [sum(vals[k] * mp.cos(mp.pi * j * (k + mp.mpf(1) / 2) / NDC) for k in range(NDC)) * 2 / NDC
                for j in range(NDC)]
        cheb[0] /= 2
        mono = [mp.mpf(0)] * NDC
        for j, cj in enumerate(cheb):
            for k, tk in enumerate(T[j]):
                mono[k] += cj * tk
        rows.append(mono)
    return rows


def cheb_monomials(f, lo, hi):
    """Degree NDC-1 Chebyshev interpolant of f on [lo, hi] as monomial coefficients in x in [-1, 1]."""
    T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for j in range(2, NDC):
        cur = [mp.mpf(0)] + [2 * c for c in T[j - 1]]
        for k, c in enumerate(T[j - 2]):
            cur[k] -= c
        T.append(cur)
    nodes = [mp.cos(mp.pi * (k + mp.mpf(1) / 2) / NDC) for k in range(NDC)]
    mid, half = (lo + hi) / 2, (hi - lo) / 2
    vals = [f(mid + half * x) for x in nodes]
    cheb = [sum(vals[k] * mp.cos(mp.pi * j * (k + mp.mpf(1) / 2) / NDC) for k in range(NDC)) * 2 / NDC for j in range(NDC)]
    cheb[0] /= 2
    mono = [mp.mpf(0)] * NDC
    for j, cj in enumerate(cheb):
        for k, tk in enumerate(T[j]):
            mono[k] += cj * tk
    return mono


def q_functional(c, u):
    """sum_k c[k] exp(-k^2 h^2 u) at 50 digits (c: the double weights the kernel uses, taken exactly)."""
    s = (mp.pi / 180) ** 2 * u
    return mp.fsum(mp.mpf(float(c[k])) * mp.exp(-(k * k) * s) for k in range(NANGLE))


def qpoly_tables(cden, cnum):
    mp.mp.dps = 50
    rows = []
    for i in range(NDI):                                         # region A: u in [i/2, (i+1)/2]
        lo, hi = mp.mpf(i) / 2, mp.mpf(i + 1) / 2
        rows.append((cheb_monomials(lambda u: q_functional(cden, u), lo, hi),
                     cheb_monomials(lambda u: q_functional(cnum, u), lo, hi)))
    width = (mp.mpf('0.25') - mp.mpf(QA_MIN)) / NQB
    for i in range(NQB):                                         # region B: |a| in [QA_MIN + i w, QA_MIN + (i+1) w]
        lo, hi = mp.mpf(QA_MIN) + i * width, mp.mpf(QA_MIN) + (i + 1) * width
        rows.append((cheb_monomials(lambda a: q_functional(cden, 1 / (a * a)), lo, hi),
                     cheb_monomials(lambda a: q_functional(cnum, 1 / (a * a)), lo, hi)))
    return rows


def simpson_weights(x):
    n = len(x)
    w = np.zeros(n)
    for i in range(0, n - 2, 2):
        h0, h1 = x[i + 1] - x[i], x[i + 2] - x[i + 1]
        hsum, hprod, h0divh1 = h0 + h1, h0 * h1, h0 / h1
        f = hsum / 6.0
        w[i] += f * (2.0 - 1.0 / h0divh1)
        w[i + 1] += f * (hsum * (hsum / hprod))
        w[i + 2] += f * (2.0 - h0divh1)
    return w


def main():
    dpoly = dpoly_tables()

    alpha = np.linspace(0, np.pi / 2, NANGLE)
    w = simpson_weights(alpha)
    cden = np.empty(NANGLE)
    cnum = np.empty(NANGLE)
    for k in range(NANGLE):
        m = NANGLE - 1 - k
        cden[k] = w[m] * np.cos(alpha[m])
        cnum[k] = cden[k] * np.sin(alpha[m])

    qpoly = qpoly_tables(cden, cnum)

    daw = []
    dfact = mp.mpf(1)
    for n in range(NDAW):
        if n > 0:
            dfact *= (2 * n + 1)
        daw.append((-1) ** n / dfact)

    lines = ['// GENERATED by tools/gen_tables.py -- do not edit by hand.',
             '// Constant tables of the PEM-v0 plume kernels (see the generator for their definitions).',
             '#pragma once', '',
             '#ifndef PEM_TABLE_DECL',
             '#define PEM_TABLE_DECL static const',
             '#endif', '',
             f'#define PEM_NDAW {NDAW}',
             f'#define PEM_NDI {NDI}',
             f'#define PEM_NDC {NDC}', '',
             '// folded Simpson weights of the flipped divergence integrals, indexed by angle k',
             f'PEM_TABLE_DECL double PEM_SIMPSON_CDEN[{NANGLE}] = {{']
    lines += [f'    {hexd(v)},' for v in cden]
    lines += ['};', f'PEM_TABLE_DECL double PEM_SIMPSON_CNUM[{NANGLE}] = {{']
    lines += [f'    {hexd(v)},' for v in cnum]
    lines += ['};', '', '// S(y) = sum_n DAW[n] y^n,  D(a) = pi a^2 S(a^2/2) for |a| < 0.25',
              f'PEM_TABLE_DECL double PEM_DAWSON[{NDAW}] = {{']
    lines += [f'    {hexd(v)},  // {mp.nstr(v, 17)}' for v in daw]
    lines += ['};', '', '// D(u), u = 1/a^2 in [0,16]: row i = floor(2u), D = sum_j DPOLY[i*NDC+j] x^j, x = 4u - 2i - 1',
              f'PEM_TABLE_DECL double PEM_DPOLY[{NDI * NDC}] = {{']
    for row in dpoly:
        lines.append('    ' + ', '.join(hexd(v) for v in row) + ',')
    lines += ['};', '',
              '// Qd(a), Qn(a): the Simpson functionals of the divergence integrals (see the generator); rows 0..NDI-1: u in',
              '// [0,16] as DPOLY; rows NDI..NDI+NQB-1: |a| in [QA_MIN, 0.25), x = 2 (t - i) - 1, t = (|a| - QA_MIN) * QB_SCALE',
              f'#define PEM_NQB {NQB}',
              f'#define PEM_QA_MIN {QA_MIN!r}',
              f'#define PEM_QB_SCALE {float(NQB / (0.25 - QA_MIN))!r}',
              f'PEM_TABLE_DECL double PEM_QPOLY[{(NDI + NQB) * NDC * 2}] = {{']
    for qd, qn in qpoly:
        lines.append('    ' + ', '.join(f'{hexd(a)}, {hexd(b)}' for a, b in zip(qd, qn)) + ',')
    lines += ['};', '']
    OUT.write_text('\n'.join(lines))
    print('wrote', OUT)

    # self-checks against 40-digit quadrature, evaluating exactly as the kernel does (double Horner)
    mp.mp.dps = 40
    worst = 0
    rng = np.random.default_rng(0)
    for u in np.concatenate([rng.uniform(0, 16, 300), [0.0, 1e-4, 0.5, 15.999999, 16.0 - 1e-12, 3.5e-4]]):
        i = min(int(2 * u), NDI - 1)
        x = 4 * u - 2 * i - 1
        acc = 0.0
        for cj in reversed([float(v) for v in dpoly[i]]):
            acc = acc * x + cj
        ex = d_of_u(mp.mpf(float(u)))
        worst = max(worst, abs(acc - ex) / ex)
    print('worst relative error of the D(u) polynomial table:', mp.nstr(worst, 3))
    worst = 0
    for a in [1e-3, 0.05, 0.2, 0.2499]:
        ex = 2 * mp.pi * mp.quad(lambda t: mp.exp(-(t / a) ** 2) * mp.sin(t), mp.linspace(0, min(mp.pi / 2, 12 * a), 9))
        y = a * a / 2
        got = np.pi * a * a * sum(float(c) * y ** n for n, c in enumerate(daw))
        worst = max(worst, abs(got - ex) / ex)
    print('worst relative error of the small-alpha series:', mp.nstr(worst, 3))
    # QPOLY evaluated as the kernel does (double Horner, both regions) against the 40-digit sums
    worst = 0
    scale = float(NQB / (0.25 - QA_MIN))
    for a in np.concatenate([rng.uniform(QA_MIN, 0.25, 200), 1 / np.sqrt(rng.uniform(1e-6, 16, 300)),
                             [QA_MIN, 0.25 - 1e-15, 0.25, 1.5707963, 15.7, 53.0, 1e6]]):
        if a >= 0.25:
            u = 1.0 / (a * a)
            i = min(int(2 * u), NDI - 1)
            x = 4 * u - 2 * i - 1
        else:
            t = (a - QA_MIN) * scale
            i = min(int(t), NQB - 1)
            x = 2 * (t - i) - 1
            i += NDI
        for which, c in enumerate((cden, cnum)):
            acc = 0.0
            for cj in reversed([float(v) for v in qpoly[i][which]]):
                acc = acc * x + cj
            ex = q_functional(c, 1 / mp.mpf(float(a)) ** 2)
            worst = max(worst, abs(acc - ex) / abs(ex))
    print('worst relative error of the Qd/Qn polynomial tables:', mp.nstr(worst, 3))


if __name__ == '__main__':
    main()
