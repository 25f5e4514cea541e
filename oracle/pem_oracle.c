/*
 * pem_oracle.c -- CPU restatement of the PEM-v0 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP kernels in hallthrusterpem_amd/csrc.  It may be
 * loaded only by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg, and
 * only as the checker / the reported CPU baseline -- never by the product path.
 *
 * It restates, in plain C and fp64, the arithmetic of the reference (paths relative to the
 * upstream repository root):
 *   oracle_cathode_f64      src/hallmd/models/cathode.py:24-38
 *   oracle_plume_f64        src/hallmd/models/plume.py:39-140
 *   oracle_thruster_f64     tests/sim_hallthruster.jl:35-48   (the reference's own analytic test
 *                           double for HallThruster.jl -- NOT the 1-D fluid solver)
 *   oracle_coupled_f64      cathode -> thruster test double -> plume, wired as
 *                           scripts/pem_v0/pem_v0_SPT-100.yml:4-6,62-63,215-219 wires them
 *   oracle_model_fidelity   src/hallmd/models/thruster.py:140-181
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against the .npz files under tests/golden,
 * which tests/golden/make_golden.py produced by importing and running the reference itself
 * (numpy 2.2.6 / scipy 1.15.3) in the development container.
 *
 * One deliberate difference in METHOD (not in result): the reference evaluates the beam
 * normaliser with six complex scipy.special.erfi calls (plume.py:64-85).  That expression is
 * identically  D(a) = 2*pi * Int_0^{pi/2} exp(-(t/a)^2) sin(t) dt  (checked to
 * <= 5e-16 relative with 40-digit mpmath).  Here D is computed from the integral with a 32-point
 * Gauss-Legendre rule on [0, min(pi/2, 6.5|a|)] (beyond 6.5|a| the integrand is < 5e-19 of its
 * peak), which the golden vectors pin to <= 1e-13 relative on a in [1e-3, 53].  Two reference
 * side effects of the erfi form are kept: D(0) is NaN, and D is NaN for |a| > 53.28349511409265
 * where erfi(a/2) overflows in the reference.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PEM_NANGLE 91
#define ORACLE_PI 3.14159265358979323846264338327950288

static int g_threads = 0; /* 0 = OpenMP default */

int oracle_set_threads(int n) {
    g_threads = n;
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ----------------------------------------------------------------------------------------------
 * cathode.py:24-38.  k = TORR_2_PA multiplies the three pressures exactly as the reference does
 * (it cancels algebraically, but the roundings are kept).  Each product/sum is rounded separately,
 * as numpy does -- no fused multiply-add (see Makefile: -ffp-contract=off).
 * ---------------------------------------------------------------------------------------------- */
static inline double cathode_one(double P_b, double V_a, double T_e, double V_vac, double Pstar, double P_T,
                                 double k) {
    double PB = P_b * k;     /* cathode.py:26 */
    double PS = Pstar * k;   /* cathode.py:30 */
    double PT = P_T * k;     /* cathode.py:31 */
    double lg = log(1.0 + PB / PT);
    double V = V_vac + T_e * lg;                 /* cathode.py:34, left to right */
    V = V - (T_e / (PT + PS)) * PB;
    if (V < 0.0) V = 0.0;                        /* cathode.py:35  (NaN compares false -> stays NaN) */
    if (V > V_a) V = V_a;                        /* cathode.py:36-37 */
    return V;
}

int oracle_cathode_f64(long n, const double* P_b, const double* V_a, const double* T_e, const double* V_vac,
                       const double* Pstar, const double* P_T, double torr2pa, double* V_cc) {
    if (n < 0) return 1;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i)
        V_cc[i] = cathode_one(P_b[i], V_a[i], T_e[i], V_vac[i], Pstar[i], P_T[i], torr2pa);
    return 0;
}

/* ----------------------------------------------------------------------------------------------
 * Tables shared by every sample: the angle grid of plume.py:53 and the composite-Simpson weights
 * scipy.integrate.simpson(..., x=alpha_rad) applies on it (odd point count -> the irregular-
 * spacing formula of scipy/integrate/_quadrature.py:_basic_simpson, restated in simpson_weights).
 * ---------------------------------------------------------------------------------------------- */
static double g_alpha[PEM_NANGLE];   /* np.linspace(0, pi/2, 91)                       plume.py:53   */
static double g_cos[PEM_NANGLE];     /* cos(alpha_rad)                                 plume.py:118  */
static double g_sin[PEM_NANGLE];     /* sin(alpha_rad)                                 plume.py:119  */
static double g_w[PEM_NANGLE];       /* Simpson weight of grid point m                 plume.py:122  */
static double g_glx[32], g_glw[32];  /* Gauss-Legendre nodes/weights on [-1, 1]                       */
static int g_tables_ready = 0;

static void simpson_weights(const double* x, int n, double* w) {
    /* weights such that sum_m w[m]*y[m] equals scipy's _basic_simpson(y, 0, n-2, x) term by term */
    for (int m = 0; m < n; ++m) w[m] = 0.0;
    for (int i = 0; i + 2 < n; i += 2) {
        double h0 = x[i + 1] - x[i], h1 = x[i + 2] - x[i + 1];
        double hsum = h0 + h1, hprod = h0 * h1, h0divh1 = (h1 != 0.0) ? h0 / h1 : 0.0;
        double inv = (h0divh1 != 0.0) ? 1.0 / h0divh1 : 0.0;
        double q = (hprod != 0.0) ? hsum / hprod : 0.0;
        double f = hsum / 6.0;
        w[i] += f * (2.0 - inv);
        w[i + 1] += f * (hsum * q);
        w[i + 2] += f * (2.0 - h0divh1);
    }
}

static void gauss_legendre(int n, double* x, double* w) {
    /* Newton iteration on P_n, started from the Chebyshev guess; converges to the last bit */
    for (int i = 0; i < n; ++i) {
        long double z = cosl(3.14159265358979323846264338327950288L * (i + 0.75L) / (n + 0.5L)), pp = 0;
        for (int it = 0; it < 100; ++it) {
            long double p1 = 1, p2 = 0;
            for (int j = 1; j <= n; ++j) {
                long double p3 = p2;
                p2 = p1;
                p1 = ((2 * j - 1) * z * p2 - (j - 1) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1);
            long double dz = p1 / pp;
            z -= dz;
            if (fabsl(dz) < 1e-19L) break;
        }
        x[i] = (double)z;
        w[i] = (double)(2 / ((1 - z * z) * pp * pp));
    }
}

static void init_tables(void) {
    if (g_tables_ready) return;
#pragma omp critical(oracle_tables)
    {
        if (!g_tables_ready) {
            /* np.linspace(0, pi/2, 91): start + arange(91)*step with step = (stop-start)/90, last = stop */
            double stop = ORACLE_PI / 2, step = stop / 90.0;
            for (int m = 0; m < PEM_NANGLE; ++m) g_alpha[m] = (double)m * step;
            g_alpha[PEM_NANGLE - 1] = stop;
            for (int m = 0; m < PEM_NANGLE; ++m) {
                g_cos[m] = cos(g_alpha[m]);
                g_sin[m] = sin(g_alpha[m]);
            }
            simpson_weights(g_alpha, PEM_NANGLE, g_w);
            gauss_legendre(32, g_glx, g_glw);
            g_tables_ready = 1;
        }
    }
}

const double* oracle_angle_grid(void) {
    init_tables();
    return g_alpha;
}

/* D(a): the bracket of plume.py:64-73 / 75-85 without the (1-c0) or c0 numerator. */
double oracle_normaliser(double a) {
    init_tables();
    if (isnan(a) || a == 0.0) return NAN;                 /* reference: 0 * inf / NaN propagation   */
    double aa = fabs(a);                                  /* D is even in a (erfi is odd)           */
    if (aa > 53.28349511409265) return NAN;               /* reference: erfi(a/2) overflow -> NaN   */
    double T = fmin(ORACLE_PI / 2, 6.5 * aa), half = 0.5 * T, sum = 0.0;
    for (int i = 0; i < 32; ++i) {
        double t = half * (g_glx[i] + 1.0), u = t / aa;
        sum += g_glw[i] * exp(-(u * u)) * sin(t);
    }
    return 2.0 * ORACLE_PI * half * sum;
}

/* One sample of plume.py:39-140 for all R radii.  j_ion: [91][R] (row-major, as numpy lays out the
 * trailing (91, R) axes), div/Tc: [R].  Returns the sample's `invalid_idx` (plume.py:105). */
static int plume_one(double P_b, double c0, double c1, double c2, double c3, double c4, double c5,
                     double sigma, double I_B0, const double* T, int R, const double* radii, double k,
                     double* j_ion, double* div, double* Tc) {
    double P_B = P_b * k;                               /* plume.py:40 */
    double n = c4 * P_B + c5;                           /* plume.py:56 */
    double a1 = c2 * P_B + c3;                          /* plume.py:59 */
    if (a1 > ORACLE_PI / 2) a1 = ORACLE_PI / 2;         /* plume.py:60 (upper clip only) */
    double a2 = a1 / c1;                                /* plume.py:61 */
    double A1 = (1.0 - c0) / oracle_normaliser(a1);     /* plume.py:64-73 */
    double A2 = c0 / oracle_normaliser(a2);             /* plume.py:75-85 */

    double g1[PEM_NANGLE], g2[PEM_NANGLE];
    for (int m = 0; m < PEM_NANGLE; ++m) {              /* the two Gaussians of plume.py:99-100 */
        double u1 = g_alpha[m] / a1, u2 = g_alpha[m] / a2;
        g1[m] = exp(-(u1 * u1));
        g2[m] = exp(-(u2 * u2));
    }
    int invalid = (a1 <= 0.0);                          /* plume.py:105, first term */
    for (int r = 0; r < R; ++r) {
        double rad = radii[r];
        double decay = exp(-rad * n * sigma);                           /* plume.py:95 */
        double j_cex = I_B0 * (1.0 - decay) / (2.0 * ORACLE_PI * (rad * rad));   /* plume.py:96 */
        double base = I_B0 * decay / (rad * rad);                       /* plume.py:98 */
        double num = 0.0, den = 0.0;
        for (int m = 0; m < PEM_NANGLE; ++m) {
            double jb = base * A1 * g1[m], js = base * A2 * g2[m];      /* plume.py:99-100 */
            double j = jb + js + j_cex;                                 /* plume.py:102 */
            j_ion[(size_t)m * R + r] = j;
            if (j <= 0.0) invalid = 1;                                  /* plume.py:105, second term */
            /* plume.py:117-123: the beam part, flipped, at grid point mm = 90 - m */
            int mm = PEM_NANGLE - 1 - m;
            double d = (jb + js) * g_cos[mm];
            den += g_w[mm] * d;
            num += g_w[mm] * (d * g_sin[mm]);
        }
        double cos_div = num / den;                                     /* plume.py:124 */
        if (isinf(cos_div) && cos_div > 0) cos_div = NAN;               /* plume.py:125 */
        div[r] = acos(cos_div);                                         /* plume.py:127 */
        if (Tc) Tc[r] = (*T) * cos_div;                                 /* plume.py:137 */
    }
    if (invalid)                                                        /* plume.py:106 (j_ion only) */
        for (int i = 0; i < PEM_NANGLE * R; ++i) j_ion[i] = 1e-20;
    return invalid;
}

int oracle_plume_f64(long n, int R, const double* radii, double torr2pa, const double* P_b, const double* c0,
                     const double* c1, const double* c2, const double* c3, const double* c4, const double* c5,
                     const double* sigma_cex, const double* I_B0, const double* T, double* j_ion,
                     double* div_angle, double* T_c, unsigned char* invalid) {
    if (n < 0 || R < 1) return 1;
    init_tables();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        int inv = plume_one(P_b[i], c0[i], c1[i], c2[i], c3[i], c4[i], c5[i], sigma_cex[i], I_B0[i],
                            T ? &T[i] : NULL, R, radii, torr2pa, j_ion + (size_t)i * PEM_NANGLE * R,
                            div_angle + (size_t)i * R, (T && T_c) ? T_c + (size_t)i * R : NULL);
        if (invalid) invalid[i] = (unsigned char)inv;
    }
    return 0;
}

/* The term decomposition behind one plume result, for the conditioning-aware tolerances of the parity tests
 * (tests/parity_rules.py): j_ion[m] = X1 g1[m] + X2 g2[m] + j_cex and cos_div = num / den are sums that cancel for
 * inputs outside the priors (negative amplitudes or densities); a bound on the error of such a sum needs the size of
 * its terms, not of its result.  terms: [n][R][8] = {X1, X2, j_cex, decay, den, num, den_abs, num_abs} with
 * X1 = base*A1, X2 = base*A2 and den_abs / num_abs the same Simpson sums over |w_m| (|jb| + |js|) ...; alpha: [n][2]. */
int oracle_plume_terms_f64(long n, int R, const double* radii, double torr2pa, const double* P_b, const double* c0,
                           const double* c1, const double* c2, const double* c3, const double* c4, const double* c5,
                           const double* sigma_cex, const double* I_B0, double* terms, double* alpha) {
    if (n < 0 || R < 1) return 1;
    init_tables();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double P_B = P_b[i] * torr2pa;
        double nn = c4[i] * P_B + c5[i];
        double a1 = c2[i] * P_B + c3[i];
        if (a1 > ORACLE_PI / 2) a1 = ORACLE_PI / 2;
        double a2 = a1 / c1[i];
        double A1 = (1.0 - c0[i]) / oracle_normaliser(a1), A2 = c0[i] / oracle_normaliser(a2);
        alpha[2 * i] = a1;
        alpha[2 * i + 1] = a2;
        for (int r = 0; r < R; ++r) {
            double rad = radii[r];
            double decay = exp(-rad * nn * sigma_cex[i]);
            double base = I_B0[i] * decay / (rad * rad);
            double X1 = base * A1, X2 = base * A2;
            double den = 0, num = 0, dabs = 0, nabs = 0;
            for (int m = 0; m < PEM_NANGLE; ++m) {
                double u1 = g_alpha[m] / a1, u2 = g_alpha[m] / a2;
                double jb = X1 * exp(-(u1 * u1)), js = X2 * exp(-(u2 * u2));
                int mm = PEM_NANGLE - 1 - m;
                double d = (jb + js) * g_cos[mm], da = (fabs(jb) + fabs(js)) * fabs(g_cos[mm]);
                den += g_w[mm] * d;
                num += g_w[mm] * (d * g_sin[mm]);
                dabs += fabs(g_w[mm]) * da;
                nabs += fabs(g_w[mm]) * (da * fabs(g_sin[mm]));
            }
            double* t = terms + ((size_t)i * R + r) * 8;
            t[0] = X1; t[1] = X2;
            t[2] = I_B0[i] * (1.0 - decay) / (2.0 * ORACLE_PI * (rad * rad));
            t[3] = decay; t[4] = den; t[5] = num; t[6] = dabs; t[7] = nabs;
        }
    }
    return 0;
}

/* ----------------------------------------------------------------------------------------------
 * tests/sim_hallthruster.jl:35-48 -- the reference's analytic stand-in for HallThruster.jl.
 * q and m_ion are the script's own literals (1.6e-19, 2.18e-25), not CODATA values.
 * ---------------------------------------------------------------------------------------------- */
static inline void thruster_one(double V_a, double V_cc, double mdot, double a1, double* I_B0, double* I_d,
                                double* T, double* eta_c, double* eta_m, double* eta_v, double* eta_a,
                                double* v_exh) {
    const double q = 1.6e-19, m_ion = 2.18e-25;
    double beam = (q / m_ion) * mdot;                           /* :37 */
    double ceff = 1.0 - a1 * 2.0;                               /* :38 */
    double Id = beam / ceff;                                    /* :39 */
    double v = sqrt(2.0 * q * (V_a - V_cc) / m_ion);            /* :40 */
    double thrust = mdot * v;                                   /* :41 */
    if (I_B0) *I_B0 = beam;
    if (I_d) *I_d = Id;
    if (T) *T = thrust;
    if (eta_c) *eta_c = ceff;
    if (eta_m) *eta_m = 1.0 - a1 * 5.0;                         /* :42 */
    if (eta_v) *eta_v = 1.0 - a1 * 2.0;                         /* :43 */
    if (eta_a) *eta_a = 0.5 * (thrust * thrust) / (mdot * V_a * Id);   /* :44 */
    if (v_exh) *v_exh = v;
}

int oracle_thruster_f64(long n, const double* V_a, const double* V_cc, const double* mdot_a, const double* a_1,
                        double* I_B0, double* I_d, double* T, double* eta_c, double* eta_m, double* eta_v,
                        double* eta_a, double* v_exh) {
    if (n < 0) return 1;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i)
        thruster_one(V_a[i], V_cc[i], mdot_a[i], a_1[i], I_B0 ? I_B0 + i : NULL, I_d ? I_d + i : NULL,
                     T ? T + i : NULL, eta_c ? eta_c + i : NULL, eta_m ? eta_m + i : NULL,
                     eta_v ? eta_v + i : NULL, eta_a ? eta_a + i : NULL, v_exh ? v_exh + i : NULL);
    return 0;
}

/* u_ion(z) of sim_hallthruster.jl:46-47 on z = range(z0, z1, length = ncells) */
int oracle_thruster_uion_f64(long n, const double* v_exh, double z0, double z1, int ncells, double* z,
                             double* u_ion) {
    if (n < 0 || ncells < 2) return 1;
    for (int c = 0; c < ncells; ++c) z[c] = z0 + (z1 - z0) * ((double)c / (double)(ncells - 1));
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i)
        for (int c = 0; c < ncells; ++c)
            u_ion[(size_t)i * ncells + c] = v_exh[i] / (1.0 + exp(-100.0 * (z[c] - 0.04)));
    return 0;
}

/* cathode -> thruster test double -> plume (R = 1 at `radius`), the graph of pem_v0_SPT-100.yml */
int oracle_coupled_f64(long n, double torr2pa, double radius, const double* P_b, const double* V_a,
                       const double* T_e, const double* V_vac, const double* Pstar, const double* P_T,
                       const double* mdot_a, const double* a_1, const double* c0, const double* c1,
                       const double* c2, const double* c3, const double* c4, const double* c5,
                       const double* sigma_cex, double* V_cc, double* I_B0, double* T, double* j_ion,
                       double* div_angle, double* T_c, unsigned char* invalid) {
    if (n < 0) return 1;
    init_tables();
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        double vcc = cathode_one(P_b[i], V_a[i], T_e[i], V_vac[i], Pstar[i], P_T[i], torr2pa);
        double ib0, thrust;
        thruster_one(V_a[i], vcc, mdot_a[i], a_1[i], &ib0, NULL, &thrust, NULL, NULL, NULL, NULL, NULL);
        double div, tc;
        int inv = plume_one(P_b[i], c0[i], c1[i], c2[i], c3[i], c4[i], c5[i], sigma_cex[i], ib0, &thrust, 1,
                            &radius, torr2pa, j_ion + (size_t)i * PEM_NANGLE, &div, &tc);
        V_cc[i] = vcc;
        if (I_B0) I_B0[i] = ib0;
        if (T) T[i] = thrust;
        div_angle[i] = div;
        T_c[i] = tc;
        if (invalid) invalid[i] = (unsigned char)inv;
    }
    return 0;
}

/* thruster.py:140-181 with the caller-resolved config values: returns dt, writes num_cells/ncharge */
double oracle_model_fidelity(int f0, int f1, double domain_hi, double anode_pot, double cathode_pot,
                             double mol_weight, double avogadro, double charge, double cfl, int* num_cells,
                             int* ncharge) {
    int nc = 50 * (f0 + 2), nq = f1 + 1;                 /* thruster.py:159-160 */
    double mi = mol_weight / avogadro / 1000.0;          /* thruster.py:176 */
    double dx = domain_hi / (double)(nc + 1);            /* thruster.py:177 */
    double u = sqrt(2.0 * nq * charge * (anode_pot - cathode_pot) / mi);   /* thruster.py:178 */
    if (num_cells) *num_cells = nc;
    if (ncharge) *ncharge = nq;
    return cfl * dx / u;                                 /* thruster.py:179 */
}
