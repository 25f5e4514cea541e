// pem_latent.h -- internal interface between the C-ABI entry point pem_coupled_latent_f64_dev (pem_kernels.hip) and the
// lane-per-sample fused compression kernel (pem_latent.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pem {

struct LatentArgs {
    long long n;
    double torr2pa, radius;
    const double *P_b, *V_a, *T_e, *V_vac, *Pstar, *P_T, *mdot_a, *a_1, *c0, *c1, *c2, *c3, *c4, *c5, *sigma;
    double *V_cc, *div, *Tc;
    uint8_t* invalid;
};

// latent[n][rank] = norm(j_ion) @ basis[91][rank] (+ V_cc, div_angle, T_c, invalid), 1 <= rank <= 8
int launch_coupled_latent(const LatentArgs& a, int rank, bool log_norm, const double* basis, double* latent, hipStream_t st);

}  // namespace pem
