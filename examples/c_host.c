/* A plain C host of libpem_hip.so: no Python, no PyTorch -- the C ABI of include/pem_hip.h is all there is.
 *
 *   gcc -O2 -Iinclude examples/c_host.c -Lhallthrusterpem_amd -lpem_hip -Wl,-rpath,$PWD/hallthrusterpem_amd \
 *       -Wl,-rpath-link,/opt/rocm/lib -lm -o examples/c_host && examples/c_host 100000
 *
 * Evaluates n samples of the coupled cathode -> thruster (analytic test double) -> plume model through the
 * host-pointer entry point and prints a few results and a checksum (tests/test_c_host.py compares them with the
 * Python path). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pem_hip.h"

/* a tiny deterministic generator: the inputs only have to be the same ones the test regenerates */
static double unit(uint64_t* s) {
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(*s >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 10000;
    if (pem_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", pem_last_error());
        return 2;
    }
    double* in[15];
    for (int i = 0; i < 15; ++i) in[i] = malloc(n * sizeof(double));
    uint64_t s = 12345;
    for (size_t i = 0; i < n; ++i) {
        in[0][i] = pow(10.0, -8.0 + 4.0 * unit(&s)); /* P_b   */
        in[1][i] = 200.0 + 200.0 * unit(&s);          /* V_a   */
        in[2][i] = 1.0 + 4.0 * unit(&s);              /* T_e   */
        in[3][i] = 60.0 * unit(&s);                   /* V_vac */
        in[4][i] = 1e-5 + 9e-5 * unit(&s);            /* Pstar */
        in[5][i] = 1e-5 + 9e-5 * unit(&s);            /* P_T   */
        in[6][i] = 2e-6 + 5e-6 * unit(&s);            /* mdot_a */
        in[7][i] = pow(10.0, -2.5 + 1.5 * unit(&s));  /* a_1   */
        in[8][i] = unit(&s);                          /* c0    */
        in[9][i] = 0.1 + 0.8 * unit(&s);              /* c1    */
        in[10][i] = -15.0 + 30.0 * unit(&s);          /* c2    */
        in[11][i] = 0.2 + 1.370796 * unit(&s);        /* c3    */
        in[12][i] = pow(10.0, 18.0 + 4.0 * unit(&s)); /* c4    */
        in[13][i] = pow(10.0, 14.0 + 4.0 * unit(&s)); /* c5    */
        in[14][i] = 51e-20 + 7e-20 * unit(&s);        /* sigma_cex */
    }
    double *V_cc = malloc(n * 8), *I_B0 = malloc(n * 8), *T = malloc(n * 8), *div = malloc(n * 8), *Tc = malloc(n * 8);
    double* j_ion = malloc(n * PEM_NANGLE * 8);
    uint8_t* invalid = malloc(n);
    const int rc = pem_coupled_f64(n, 133.322, 1.0, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], in[8], in[9],
                                   in[10], in[11], in[12], in[13], in[14], V_cc, I_B0, T, j_ion, div, Tc, invalid);
    if (rc != PEM_OK) {
        fprintf(stderr, "pem_coupled_f64 failed (%d): %s\n", rc, pem_last_error());
        return 1;
    }
    double sum_v = 0.0, sum_j = 0.0, sum_d = 0.0;
    size_t n_invalid = 0;
    for (size_t i = 0; i < n; ++i) {
        sum_v += V_cc[i];
        sum_d += div[i];
        n_invalid += invalid[i];
        for (int k = 0; k < PEM_NANGLE; ++k) sum_j += j_ion[i * PEM_NANGLE + k];
    }
    printf("%s\n", pem_version());
    printf("n=%zu sum_V_cc=%.17g sum_div=%.17g sum_j_ion=%.17g invalid=%zu\n", n, sum_v, sum_d, sum_j, n_invalid);
    printf("sample0 V_cc=%.17g div_angle=%.17g T_c=%.17g j_ion[0]=%.17g j_ion[90]=%.17g\n", V_cc[0], div[0], Tc[0], j_ion[0],
           j_ion[90]);
    return 0;
}
