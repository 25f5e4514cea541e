/*
 * pem_hip.h -- C ABI of libpem_hip.so: the MI355X (gfx950) batched evaluator for the PEM-v0
 * cathode -> thruster -> plume sub-models of JANUS-Institute/HallThrusterPEM (hallmd 0.3.0).
 *
 * Every entry point replaces one vectorised Python model callable of the reference (paths are
 * relative to the upstream repository root).  The reference has no FFI of its own -- its models
 * are NumPy functions called as  model(inputs: dict[str, ndarray]) -> dict[str, ndarray]  by amisc
 * (scripts/pem_v0/pem_v0_SPT-100.yml:6,63,216) -- so the boundary is drawn one level below that
 * call: plain fp64 SoA buffers, one value per Monte-Carlo sample, caller-owned outputs.
 *
 *   pem_cathode_f64*    src/hallmd/models/cathode.py:16-38      cathode_coupling()
 *   pem_plume_f64*      src/hallmd/models/plume.py:21-159       current_density()
 *   pem_thruster_f64*   tests/sim_hallthruster.jl:35-48         the reference's analytic stand-in
 *                        for HallThruster.jl (a TEST DOUBLE of the thruster stage; the 1-D fluid
 *                        solver itself is a third-party Julia program and out of scope)
 *   pem_coupled_f64*    the three stages fused, wired as pem_v0_SPT-100.yml wires the components
 *                        (V_cc: cathode -> thruster; I_B0: thruster -> plume; T -> T_c)
 *
 * Conventions
 *   - All arrays are contiguous fp64 of length n (one entry per sample) unless stated.
 *   - `*_dev` functions take DEVICE pointers (HBM of the current HIP device) and enqueue on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream) without synchronising.
 *     Functions without the suffix take HOST pointers, stage through device memory and return
 *     after the results are back in the caller's buffers (calls of up to 256 KB of arrays: the
 *     kernels work on a pinned host buffer directly, without the two copies).
 *   - Nothing is retained past the call.  Inputs are never written.
 *   - j_ion is laid out [n][91][n_radii] (row-major), exactly numpy's (..., 91, R) result of
 *     plume.py:102; angle k is k degrees from the thruster centreline (plume.py:53).
 *   - Physics failures are data, not errors, as in the reference: an invalid plume sample
 *     (alpha1 <= 0 or any j_ion <= 0, plume.py:105) has its j_ion row set to 1e-20 and, if
 *     `invalid` is non-NULL, invalid[i] = 1.  NaN inputs propagate.
 *   - `torr2pa` is pem_core.constants.TORR_2_PA, which the reference imports from an
 *     un-vendored package (cathode.py:10, plume.py:12); it is a run-time argument here.
 *   - Return value: PEM_OK or a PEM_ERR_* code; pem_last_error() describes the last failure on
 *     the calling thread.  There is no CPU fallback: without a HIP device every compute entry
 *     point fails with PEM_ERR_NO_DEVICE.
 */
#ifndef PEM_HIP_H
#define PEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEM_NANGLE 91 /* points of the fixed 0..90 degree sweep, plume.py:53 */

#define PEM_OK 0
#define PEM_ERR_INVALID_ARG 1
#define PEM_ERR_HIP 2
#define PEM_ERR_NO_DEVICE 3

typedef void* pem_stream_t; /* hipStream_t */

/* ---- library state ------------------------------------------------------------------------- */
const char* pem_version(void);
const char* pem_last_error(void);
int pem_device_count(void);                 /* number of HIP devices, 0 if none / no driver        */
/* hipSetDevice(device) and make it the process default of the host-pointer entry points, whichever thread calls them
 * (without pem_init they run on the calling thread's current device).  The *_dev entry points follow their stream. */
int pem_init(int device);
int pem_synchronize(pem_stream_t stream);   /* hipStreamSynchronize                                */
/* Tuning knob of the plume/coupled kernels: lanes that share one sample (2, 4 or 8).
 * 0 restores the default.  Returns the value in effect. */
int pem_set_lanes_per_sample(int lanes);
/* The 91-point angle grid (host memory, valid for the life of the library): j_ion_coords.    */
const double* pem_angle_grid(void);
/* Launch geometry of the persistent coupled kernel, for callers that cut one GPU's shard into range launches (the
 * multi-GPU pipeline of SURVEY.md section 8e; the reference's only parallel call site is the executor.map of
 * scripts/gen_data.py:448-460, which has no notion of a launch).  A persistent wave walks 64-sample tiles
 * w, w + waves, ...: a range that is a whole number of `samples_per_round` leaves no wave slot idle in its last round.
 * pem_persistent_grid is pure arithmetic (no device needed): the workgroups a launch over n samples takes on a device
 * with `cus` compute units holding `wg_per_cu` workgroups each (memory_bound, the profile-writing modes: balanced rounds
 * up to three rounds of work, beyond that a one-shot grid of one tile per wave that the dispatcher deals out) and the
 * samples the workgroups resident at one time cover.  pem_coupled_occupancy reports `cus` and `wg_per_cu` of
 * pem_coupled_f64_dev (profile_mode 1), pem_coupled_mixed_dev (2) or the reduced-QoI launch (0, j_ion == NULL) on the
 * calling thread's current device.                                                                                    */
int pem_persistent_grid(size_t n, int cus, int wg_per_cu, int memory_bound, size_t* workgroups, size_t* samples_per_round);
int pem_coupled_occupancy(int profile_mode, int* cus, int* wg_per_cu);

/* ---- cathode_coupling  (cathode.py:16-38) --------------------------------------------------- */
int pem_cathode_f64_dev(size_t n, const double* P_b, const double* V_a, const double* T_e,
                        const double* V_vac, const double* Pstar, const double* P_T, double torr2pa,
                        double* V_cc, pem_stream_t stream);
int pem_cathode_f64(size_t n, const double* P_b, const double* V_a, const double* T_e,
                    const double* V_vac, const double* Pstar, const double* P_T, double torr2pa,
                    double* V_cc);

/* ---- current_density  (plume.py:21-159) ------------------------------------------------------
 * radii: HOST array of n_radii sweep radii in metres (the `sweep_radius` argument), also for the
 * _dev form (up to 256 radii travel in the kernel arguments; beyond that, and for one radius with a j_ion that is not
 * 16-byte aligned, the _dev form copies them and waits for the stream before returning).  T / T_c: optional thrust
 * in, corrected thrust out (plume.py:136-140); pass NULL for
 * both to skip.  j_ion: [n][91][n_radii]; div_angle, T_c: [n][n_radii]; invalid: [n] or NULL.   */
int pem_plume_f64_dev(size_t n, int n_radii, const double* radii, double torr2pa, const double* P_b,
                      const double* c0, const double* c1, const double* c2, const double* c3,
                      const double* c4, const double* c5, const double* sigma_cex, const double* I_B0,
                      const double* T, double* j_ion, double* div_angle, double* T_c, uint8_t* invalid,
                      pem_stream_t stream);
int pem_plume_f64(size_t n, int n_radii, const double* radii, double torr2pa, const double* P_b,
                  const double* c0, const double* c1, const double* c2, const double* c3,
                  const double* c4, const double* c5, const double* sigma_cex, const double* I_B0,
                  const double* T, double* j_ion, double* div_angle, double* T_c, uint8_t* invalid);

/* ---- analytic thruster stage  (tests/sim_hallthruster.jl:35-48; a test double) ---------------
 * Any output pointer may be NULL.                                                              */
int pem_thruster_f64_dev(size_t n, const double* V_a, const double* V_cc, const double* mdot_a,
                         const double* a_1, double* I_B0, double* I_d, double* T, double* eta_c,
                         double* eta_m, double* eta_v, double* eta_a, double* v_exh, pem_stream_t stream);
int pem_thruster_f64(size_t n, const double* V_a, const double* V_cc, const double* mdot_a,
                     const double* a_1, double* I_B0, double* I_d, double* T, double* eta_c,
                     double* eta_m, double* eta_v, double* eta_a, double* v_exh);

/* u_ion(z) = v_exh / (1 + exp(-100 (z - 0.04))) on z = range(z0, z1, length = ncells)  (sim_hallthruster.jl:46-47).
 * z: [ncells] device array or NULL; u_ion: [n][ncells].                                               */
int pem_thruster_uion_f64_dev(size_t n, const double* v_exh, double z0, double z1, int ncells, double* z,
                              double* u_ion, pem_stream_t stream);
/* The two post-run filters of hallthruster_jl (thruster.py:490-502), batched.  flags[i] bit 0: T < 0 or
 * I_B0 < 0 (the reference raises "non-physical case"); bit 1 (only if use_shock): z[argmax(u_ion[i])] <
 * shock_threshold (the reference raises "shock-like behavior").  T / I_B0 may be NULL (treated as 0).  */
int pem_thruster_filter_f64_dev(size_t n, int ncells, const double* u_ion, const double* z, double shock_threshold,
                                int use_shock, const double* T, const double* I_B0, uint8_t* flags,
                                pem_stream_t stream);

/* ---- coupled cathode -> thruster -> plume, one pass, sweep radius `radius` (R = 1) -----------
 * 15 inputs per sample; outputs V_cc, div_angle, T_c always; I_B0, T, invalid optional (NULL);
 * j_ion optional: NULL selects the reduced-QoI mode that never writes the 91-point profile (and takes the two
 * divergence integrals of plume.py:117-123 from tables of the beam width where that is exact to rounding, so its
 * div_angle / T_c agree with the profile mode's to ~1e-13 relative rather than bit for bit).                      */
int pem_coupled_f64_dev(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a,
                        const double* T_e, const double* V_vac, const double* Pstar, const double* P_T,
                        const double* mdot_a, const double* a_1, const double* c0, const double* c1,
                        const double* c2, const double* c3, const double* c4, const double* c5,
                        const double* sigma_cex, double* V_cc, double* I_B0, double* T, double* j_ion,
                        double* div_angle, double* T_c, uint8_t* invalid, pem_stream_t stream);
int pem_coupled_f64(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a,
                    const double* T_e, const double* V_vac, const double* Pstar, const double* P_T,
                    const double* mdot_a, const double* a_1, const double* c0, const double* c1,
                    const double* c2, const double* c3, const double* c4, const double* c5,
                    const double* sigma_cex, double* V_cc, double* I_B0, double* T, double* j_ion,
                    double* div_angle, double* T_c, uint8_t* invalid);

/* The same evaluation with the 15 inputs TILE-INTERLEAVED: x_tiled is [ceil(n / 64)][15][64] doubles -- for every 64-sample
 * tile of the kernel the 15 rows (order of pem_coupled_f64_dev's arguments: P_b V_a T_e V_vac Pstar P_T mdot_a a_1 c0..c5
 * sigma_cex) sit in one contiguous 7680-byte block, so a wave reads ONE block per tile instead of 512 bytes from each of 15
 * arrays.  A second layout of the same `dict of arrays` the reference's callables take (cathode.py:26-31, plume.py:40-49);
 * pem_sample_tiled_f64_dev writes it directly.  Results are bit-identical to pem_coupled_f64_dev.  Device pointers only. */
int pem_coupled_tiled_f64_dev(size_t n, double torr2pa, double radius, const double* x_tiled, double* V_cc, double* I_B0,
                              double* T, double* j_ion, double* div_angle, double* T_c, uint8_t* invalid,
                              pem_stream_t stream);

/* Mixed precision (BASELINE.json configs[4], "fp64 -> fp32 mixed with tolerance check"): identical fp64
 * arithmetic; only the 91-point profile is rounded once to fp32 when it is stored (j_ion_f32: [n][91]
 * floats, 16-byte aligned).  508 instead of 872 algorithmic bytes per evaluation.  Device pointers only.  */
int pem_coupled_mixed_dev(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a,
                          const double* T_e, const double* V_vac, const double* Pstar, const double* P_T,
                          const double* mdot_a, const double* a_1, const double* c0, const double* c1,
                          const double* c2, const double* c3, const double* c4, const double* c5,
                          const double* sigma_cex, double* V_cc, double* I_B0, double* T, float* j_ion_f32,
                          double* div_angle, double* T_c, uint8_t* invalid, pem_stream_t stream);

/* ---- input samplers ------------------------------------------------------------------------------
 * Stand in for `system.sample_inputs(N, ...)` (scripts/gen_data.py:238; scripts/pem_v0/sobol.py:46-66,
 * monte_carlo.py:63-300).  amisc/uqtils are third-party and absent from the reference tree: parity is
 * UNPINNED, the formulas are this library's own (see csrc/pem_sampler.hip).
 * Counter-based (Philox4x32-10): out[d][i] depends only on (seed, stream_id, first_index + i, d), so
 * any sharding / batching of a design yields the same design.  out is SoA: row d at out + d*ld, ld >= n.
 * kind/a/b are HOST arrays of length ndim (<= PEM_SAMPLE_MAX_DIM):
 *   PEM_DIST_UNIFORM     a + (b-a) u            PEM_DIST_LOGUNIFORM  10^(a + (b-a) u)  (a, b = log10 bounds)
 *   PEM_DIST_NORMAL      a + b Phi^-1(u)  (mean a, standard deviation b)
 * swap_dim builds Saltelli blocks: -1 plain (matrix A), -2 every dimension from stream_id+1 (matrix B),
 * d >= 0 matrix A with column d taken from B.
 * pem_sample_lhs_f64_dev: Latin hypercube over n_total strata per dimension (keyed Feistel permutation
 * of the stratum index + jitter); samples first_index .. first_index+n-1 of that design.             */
#define PEM_SAMPLE_MAX_DIM 32
#define PEM_DIST_UNIFORM 0
#define PEM_DIST_LOGUNIFORM 1
#define PEM_DIST_NORMAL 2
int pem_sample_f64_dev(size_t n, uint64_t first_index, uint64_t seed, uint32_t stream_id, int ndim,
                       const int32_t* kind, const double* a, const double* b, int swap_dim, double* out,
                       size_t ld, pem_stream_t stream);
int pem_sample_lhs_f64_dev(size_t n, uint64_t first_index, uint64_t n_total, uint64_t seed, uint32_t stream_id,
                           int ndim, const int32_t* kind, const double* a, const double* b, double* out,
                           size_t ld, pem_stream_t stream);
/* pem_sample_f64_dev's numbers in the tile-interleaved layout of pem_coupled_tiled_f64_dev: out is
 * [ceil(n / 64)][ndim][64] (sample i of the call at out[(i / 64) * ndim * 64 + d * 64 + i % 64]).             */
int pem_sample_tiled_f64_dev(size_t n, uint64_t first_index, uint64_t seed, uint32_t stream_id, int ndim,
                             const int32_t* kind, const double* a, const double* b, int swap_dim, double* out,
                             pem_stream_t stream);

/* Saltelli accumulation for the Sobol' estimators (uq.sobol_sa at scripts/pem_v0/sobol.py:113 -- uqtils, third-party,
 * parity UNPINNED; estimators stated in hallthrusterpem_amd/drivers.py).  fA / fB / fAB: [nq][ld] QoI rows of the
 * A, B and AB_d blocks (fAB NULL for the mean/variance sums).  partial: [n_blocks][nq][2], one deterministic partial
 * sum per workgroup: {sum fA+fB, sum fA^2+fB^2} or {sum fB (fAB-fA), sum (fA-fAB)^2}.  nq <= 8.                    */
int pem_sobol_partial_f64_dev(size_t m, int nq, size_t ld, const double* fA, const double* fB, const double* fAB,
                              double* partial, int n_blocks, pem_stream_t stream);

/* ---- likelihood of measured ion current density (scripts/pem_v0/mcmc.py:57-106, `jion` branch; the mirrored
 * linear interpolation of monte_carlo.py:265-270 / plume.py:142-149).  The scripts are stale and untested in the
 * reference: parity UNPINNED; formula in csrc/pem_likelihood.hip.  Sample i belongs to condition i mod n_cond.
 * kidx/weight/y/inv_std: [n_cond][n_ang] device arrays (k in [0, 89], weight in [0, 1]), at most
 * PEM_LOGLIK_MAX_MEASUREMENTS entries; loglik: [n].                                                          */
#define PEM_LOGLIK_MAX_MEASUREMENTS 4096
int pem_jion_loglik_f64_dev(size_t n, int n_cond, int n_ang, const int32_t* kidx, const double* weight,
                            const double* y, const double* inv_std, const double* j_ion, double* loglik,
                            pem_stream_t stream);

/* pem_coupled_f64_dev + pem_jion_loglik_f64_dev in one launch: the 91-point profile is staged in LDS, reduced against
 * the measurements there and never written (152 bytes per evaluation).  Same per-sample results as the two-launch
 * pipeline up to the summation order of the partial sums.  n_cond * (n_ang | 1) <= PEM_FUSED_LOGLIK_MAX_MEASUREMENTS
 * (LDS table).                                                                                                       */
#define PEM_FUSED_LOGLIK_MAX_MEASUREMENTS 1024
int pem_coupled_loglik_f64_dev(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a,
                               const double* T_e, const double* V_vac, const double* Pstar, const double* P_T,
                               const double* mdot_a, const double* a_1, const double* c0, const double* c1,
                               const double* c2, const double* c3, const double* c4, const double* c5,
                               const double* sigma_cex, int n_cond, int n_ang, const int32_t* kidx,
                               const double* weight, const double* y, const double* inv_std, double* V_cc,
                               double* div_angle, double* T_c, double* loglik, uint8_t* invalid, pem_stream_t stream);

/* pem_coupled_f64_dev + pem_svd_compress_f64_dev in one launch: latent[i][r] = sum_k norm(j_ion[i][k]) basis[k][r]
 * accumulated in the registers of the angle loop, one lane per sample (csrc/pem_latent.hip) -- the profile is neither
 * stored nor staged (120 + 24 + 8 rank bytes per evaluation).  norm: PEM_NORM_NONE or PEM_NORM_LOG10; basis: [91][rank] device array, rank <=
 * PEM_FUSED_LATENT_MAX_RANK; latent: [n][rank].  An invalid sample gets the latents of its 1e-20 profile
 * (plume.py:106), as the two-launch pipeline gives.                                                               */
#define PEM_FUSED_LATENT_MAX_RANK 8
int pem_coupled_latent_f64_dev(size_t n, double torr2pa, double radius, const double* P_b, const double* V_a,
                               const double* T_e, const double* V_vac, const double* Pstar, const double* P_T,
                               const double* mdot_a, const double* a_1, const double* c0, const double* c1,
                               const double* c2, const double* c3, const double* c4, const double* c5,
                               const double* sigma_cex, int rank, int norm, const double* basis, double* latent,
                               double* V_cc, double* div_angle, double* T_c, uint8_t* invalid, pem_stream_t stream);

/* Marginal likelihood over nuisance draws and the prior of the calibration parameters (mcmc.py:100-121; unpinned).
 * loglik: [n_chains][n_draws][n_cond] per-sample sums (pem_jion_loglik / pem_coupled_loglik output).
 * out[k] = logsumexp_m( sum_e loglik[k][m][e] + sum_e -0.5 ((discharge_current - I_d[k][m][e]) / discharge_sigma)^2 ),
 * I_d = (q/m_i) mdot_a / (1 - 2 a_1) of the analytic thruster test double; mdot_a NULL drops the discharge term.
 * log_prior (NULL or [n_chains]): out becomes the log posterior, -inf where the prior is or the likelihood is NaN. */
int pem_loglik_marginal_f64_dev(size_t n_chains, int n_draws, int n_cond, const double* loglik, const double* mdot_a,
                                const double* a_1, double discharge_current, double discharge_sigma,
                                const double* log_prior, double* out, pem_stream_t stream);
/* out[i] = sum_d log pdf_d(theta[i][d]) for the PEM_DIST_* table (kind, a, b: host arrays, ndim <= 32); -inf outside
 * the support of a uniform / log-uniform variable.                                                                 */
int pem_log_prior_f64_dev(size_t n, int ndim, const int32_t* kind, const double* a, const double* b,
                          const double* theta, double* out, pem_stream_t stream);

/* ---- SVD compression / reconstruction of field QoIs (fp64 MFMA) --------------------------------
 * Stand in for amisc `Compression(method='svd')` on `j_ion` (norm log10) and `u_ion` (norm linear(1e-3)):
 * scripts/pem_v0/pem_v0_SPT-100.yml:207-214,273-280, scripts/gen_data.py:261-294.  Third-party in the
 * reference: parity UNPINNED; formulas (csrc/pem_svd.hip):
 *     latent[n][rank] = norm(field[n][dof]) @ basis[dof][rank]        (compress)
 *     field[n][dof]   = denorm(latent[n][rank] @ basis[dof][rank]^T)  (reconstruct)
 * norm: PEM_NORM_NONE x; PEM_NORM_LOG10 log10(x) / 10^y; PEM_NORM_LINEAR x*norm_scale / y/norm_scale.
 * All arrays row-major in device memory; dof <= PEM_SVD_MAX_DOF, rank <= 16.                          */
#define PEM_SVD_MAX_DOF 208
#define PEM_NORM_NONE 0
#define PEM_NORM_LOG10 1
#define PEM_NORM_LINEAR 2
int pem_svd_compress_f64_dev(size_t n, int dof, int rank, int norm, double norm_scale, const double* field,
                             const double* basis, double* latent, pem_stream_t stream);
int pem_svd_reconstruct_f64_dev(size_t n, int dof, int rank, int norm, double norm_scale, const double* latent,
                                const double* basis, double* field, pem_stream_t stream);

/* ---- sparse-grid Lagrange surrogate: batched predict ----------------------------------------------------
 * Stand in for the surrogate evaluations of amisc `System.fit(num_refine=...)` / `System.predict`
 * (scripts/fit_surr.py:101-116; BASELINE.json configs[3]).  Third-party in the reference: parity UNPINNED;
 * formula in csrc/pem_surrogate.hip, tables built by hallthrusterpem_amd/surrogate.py.
 *   out[o][i] = sum_b coef[b] * sum_nodes values[offset_b + node][o] * prod_a basis(level_a, t[dim_a][i])
 * index: [n_beta][2 + 2*PEM_SURR_MAX_ACTIVE] int32 = {n_active, value offset (rows), dims[], levels[]};
 * values: [rows][n_out]; t: [n_dim][ld] normalised coordinates in [-1, 1], n_dim <= PEM_SURR_MAX_DIM;
 * out: [n_out][ld_out]; n_out <= 16.                                                                        */
#define PEM_SURR_MAX_ACTIVE 5
#define PEM_SURR_MAX_LEVEL 4
#define PEM_SURR_MAX_DIM 32
/* max_active / max_level: the largest number of active dimensions / the highest level any multi-index of `index` has (the caller
 * built the table): they size the LDS the kernel keeps the outer dimensions' bases in -- (max_active - 1) (2^max_level + 1) + n_dim
 * doubles per thread of 256 must fit 160 KB.                                                                                          */
int pem_sparse_predict_f64_dev(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef,
                               const double* values, int n_out, const double* t, size_t ld, double* out,
                               size_t ld_out, int max_active, int max_level, pem_stream_t stream);
/* The same tables, every grid on its own: out[b][o][i] = coef[b] * (the interpolant of grid b at point i), out: [n_beta][n_out]
 * [ld_out].  The adaptive refinement scores all its candidate index sets from ONE such launch -- a prediction is linear in the
 * combination coefficients, so each trial is a [n_beta] x [n_beta][n_out n] product of these values (surrogate.py refine). */
int pem_sparse_grid_values_f64_dev(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef,
                                   const double* values, int n_out, const double* t, size_t ld, double* out,
                                   size_t ld_out, int max_active, int max_level, pem_stream_t stream);
/* Prediction + reconstruction of a compressed field QoI in one launch (round 4): outputs lat0 .. lat0 + rank - 1 of the surrogate
 * are the SVD latent coefficients the reference trains on (scripts/pem_v0/pem_v0_SPT-100.yml:273-280 `j_ion`: svd, log10 norm,
 * reconstruction_tol 0.01; scripts/fit_surr.py:101-133), and field[i][k] = denorm(sum_q latent_q(i) basis[k][q]), k < dof, leaves
 * with them -- what amisc's System.predict returns for such a variable, without pem_svd_reconstruct_f64_dev's second pass.
 * norm / norm_scale / basis as pem_svd_reconstruct_f64_dev; field: [n][dof] row-major.                                            */
int pem_sparse_predict_field_f64_dev(size_t n, int n_dim, int n_beta, const int32_t* index, const double* coef, const double* values,
                                     int n_out, const double* t, size_t ld, double* out, size_t ld_out, int max_active, int max_level,
                                     int lat0, int rank, int dof, int norm, double norm_scale, const double* basis, double* field,
                                     pem_stream_t stream);

/* ---- per-column order statistics over the sample axis -------------------------------------------------------------
 * The percentiles of scripts/gen_data.py:125-174 (`np.percentile(arr, 25 | 75, axis=0)`, NaN / interquartile-range masks) and
 * of scripts/pem_v0/monte_carlo.py:363-658 (5 / 50 / 95 % bands) at forward-UQ sizes, by exact radix selection instead of a
 * sort (csrc/pem_quantile.hip).  data: [n][ld] row-major device array of which columns 0..m-1 are used, m <= 256 (ld = m:
 * fully coalesced); for quantile i the caller gives the two ranks
 * numpy's method 'linear' reads (rank_prev[i] <= rank_next[i] < n) and its weight gamma[i] (HOST arrays);
 * out[i][c] = _lerp(x_(rank_prev[i]), x_(rank_next[i]), gamma[i]) of column c, NaN if the column holds a NaN -- equal to
 * np.percentile bit for bit.  nq <= PEM_QUANTILE_MAX_Q per call (PEM_QUANTILE_MAX_Q_WIDE for m > 128).  Allocates its workspace and synchronises the stream.
 * From n * m = 2^25 values on, every 32nd row is examined first and brackets the wanted ranks, which leaves two passes over the
 * data instead of four; a call whose data defeat the brackets (the counts say so) repeats with the four passes -- the result is
 * the same either way.  Environment: PEM_QUANTILE_PILOT = that stride (0: never), PEM_QUANTILE_PILOT_MIN = the smallest n * m it
 * is used for.  pem_quantiles_last_path(): how the last call went (0 four passes, 1 brackets held, 2 brackets failed, four passes).   */
#define PEM_QUANTILE_MAX_Q 6
#define PEM_QUANTILE_MAX_Q_WIDE 3      /* per call for 128 < m <= 256 columns */
int pem_quantiles_f64_dev(size_t n, int m, const double* data, size_t ld, int nq, const uint64_t* rank_prev, const uint64_t* rank_next,
                          const double* gamma, double* out, pem_stream_t stream);
/* The same over a strided view: value (row, c) at data[row * ld + c * cs] -- (ld >= m, cs = 1) is the form above; (ld = 1, cs >= n)
 * takes the m columns from m contiguous arrays of n values cs apart, e.g. the [3][n] reduced-QoI tensor of a batch
 * (V_cc, div_angle, T_c in ONE call instead of three).                                                                            */
int pem_quantiles_strided_f64_dev(size_t n, int m, const double* data, size_t ld, size_t cs, int nq, const uint64_t* rank_prev,
                                  const uint64_t* rank_next, const double* gamma, double* out, pem_stream_t stream);
int pem_quantiles_last_path(void);

/* Fused Monte-Carlo evaluation + percentiles of the profile, counted where the profile is produced (round 4; csrc/pem_qfused.h):
 * `sample_inputs` + `predict` of scripts/gen_data.py:238-239 followed by the `np.percentile(j_ion, ..., axis=0)` of
 * gen_data.py:163-168 and scripts/pem_v0/monte_carlo.py:363-658, without reading the profile back.  The arguments of
 * pem_coupled_mc_f64_dev (plain Monte-Carlo block: swap_dim -1) and of pem_quantiles_f64_dev (HOST arrays rank_prev / rank_next /
 * gamma of nq <= PEM_QUANTILE_MAX_Q quantiles of the n samples; q_out [nq][91] on the device).  Samples 0 .. ceil(n / 32) - 1 are
 * evaluated first and bracket the wanted ranks; ONE launch then evaluates all n samples, counts every profile value against the
 * brackets in LDS and writes the values inside them (4 %) out as records, from which the order statistics are selected.
 * j_ion NULL: the profile is never stored (pilot_rows: room for ceil(n / 32) x 91 doubles then holds the pilot's rows; ignored
 * when j_ion is given).  *fused_ok = 1: q_out equals np.percentile of the profile bit for bit.  *fused_ok = 0: the selection
 * declined -- brackets of one key or overlapping (heavy ties), a rank outside its bracket, a non-finite profile value, record
 * overflow -- q_out is not written; all other outputs are complete either way, and the caller takes pem_quantiles_f64_dev over
 * the stored profile.  n >= PEM_MC_STATS_MIN_N.  Synchronises the stream.
 * The premask (row_certain / row_uncertain [n] bytes on the device and premask_ok non-NULL, nq <= 5): quantiles q25 and q75 of the
 * call are the quartiles of gen_data.py:163-164; the outlier bounds p25 - f iqr, p75 + f iqr are then known to lie in intervals
 * (the quartiles lie in their brackets), and the counting launch also writes, per sample, how many of its 91 values lie outside
 * the bounds FOR CERTAIN and how many are UNCERTAIN (inside one of the two intervals).  certain > int(0.75 * 91) is an outlier,
 * certain + uncertain <= that is not, and the caller settles the rest from the exact bounds: gen_data.py:166-168 without a pass
 * over the profile.  *premask_ok = 0: not produced (bounds zero or not finite, f < 0, nq > 5, or the selection declined).
 * q_scalars (optional; [nq][3] on the device): the same nq percentiles of V_cc, div_angle and T_c -- which must then be rows 0, 1, 2
 * of one array (the [3][n] reduced-QoI tensor of a batch) -- as pem_quantiles_strided_f64_dev gives them, selected on a second
 * thread and a stream of the library's own while the calling thread takes the profile's records through their passes; written
 * whether or not the profile's selection declined.                                                                                */
#define PEM_MC_STATS_MIN_N 4096
int pem_coupled_mc_stats_f64_dev(size_t n, uint64_t first_index, uint64_t seed, uint32_t stream_id, const int32_t* kind, const double* a,
                                 const double* b, double torr2pa, double radius, double* x_out, size_t ld, double* V_cc, double* I_B0,
                                 double* T, double* j_ion, double* pilot_rows, double* div_angle, double* T_c, uint8_t* invalid, int nq,
                                 const uint64_t* rank_prev, const uint64_t* rank_next, const double* gamma, double* q_out, double* q_scalars,
                                 int* fused_ok, int q25, int q75, double iqr_factor, uint8_t* row_certain, uint8_t* row_uncertain,
                                 int* premask_ok, pem_stream_t stream);

/* The per-sample masks of `_filter_outputs` (scripts/gen_data.py:150-168) for one output variable in one pass over it
 * (csrc/pem_masks.hip): data [n][ld] row-major, entries 0..m-1 of a sample; lo / hi: m per-entry bounds each (DEVICE arrays:
 * p25 - f iqr and p75 + f iqr, computed by the caller from pem_quantiles_f64_dev's result as the reference does);
 * nan_out[i] = 1 if sample i holds a NaN (`np.any(np.isnan(arr), axis=rest)`), outside_out[i] = the number of its entries with
 * x < lo or x > hi (`np.sum((arr < lo) | (arr > hi), axis=rest)`; a comparison with a NaN is false, as in numpy) -- the caller
 * compares it with int(0.75 * m).  1 <= m <= PEM_ROW_MASKS_MAX_M.  Asynchronous on `stream`.                                  */
#define PEM_ROW_MASKS_MAX_M 512
int pem_row_masks_f64_dev(size_t n, int m, const double* data, size_t ld, const double* lo, const double* hi, uint8_t* nan_out,
                          int32_t* outside_out, pem_stream_t stream);

/* The masks of a campaign's SCALAR outputs and the verdict of the profile's premask counts in one pass, one thread per sample
 * (drivers.forward_uq_statistics; gen_data.py:150-168 for variables of one entry per sample): vars: HOST array of nvar <= 8 device
 * pointers to n values each; q: the variables' percentiles on the device, rows of nvar values q_ld apart (pem_coupled_mc_stats_f64_dev's
 * q_scalars), of which rows row25 / row75 are the quartiles: the bounds are p25 - f iqr and p75 + f iqr, rounded as numpy rounds them;
 * nan_out / outl_out [nvar][mask_ld] bytes, mask_ld >= n (a multiple of 4 lets the pass write words) (0 / 1: np.isnan(x); (x < lo) | (x > hi)).  With row_certain / row_uncertain (the per-sample counts
 * of pem_coupled_mc_stats_f64_dev's premask; both or neither) the arrays have one row more, the profile's: nan_out[nvar][i] = 0,
 * outl_out[nvar][i] = certain > thresh, and the samples with certain <= thresh < certain + uncertain -- whose verdict the exact bounds
 * must settle -- are appended to open_rows (int64, room for `cap`; unordered) and counted in *open_count (device int32, zeroed by the
 * caller; more than cap: the list is incomplete).  Asynchronous on `stream`.                                                       */
int pem_campaign_masks_f64_dev(size_t n, int nvar, const double* const* vars, const double* q, int q_ld, int row25, int row75,
                               double iqr_factor, uint8_t* nan_out,
                               uint8_t* outl_out, size_t mask_ld, const uint8_t* row_certain, const uint8_t* row_uncertain, int thresh,
                               int64_t* open_rows, int32_t* open_count, int cap, pem_stream_t stream);

/* The multi-rank building blocks of the same selection (samples sharded over GPUs; hallthrusterpem_amd/percentiles.py drives the
 * levels and all-reduces between them): per-column min / max of the order-preserving 64-bit image of the values (sign bit
 * flipped, negative values inverted; kmin > kmax: no finite-or-infinite value) and a NaN flag; and the histogram of the keys
 * inside caller-given ranges, nr = 1, 2, 4 or 6 ranges per column, hist[c][r][bin] (zeroed here), m * nr * bins <= 36864.
 * Bin of key k in [klo, khi]: d = (k - klo) >> shift, shift the smallest with (khi - klo) >> shift < 2^31; bin = d if
 * ((khi - klo) >> shift) < bins, else floor(d * mult / 2^32), mult = min(2^32 - 1, floor(2^32 * bins / (((khi - klo) >> shift) + 1))).
 * All arrays on the device.  pem_key_minmax synchronises the stream; pem_range_hist does not.                                  */
int pem_key_minmax_f64_dev(size_t n, int m, const double* data, size_t ld, uint64_t* kmin, uint64_t* kmax, int32_t* has_nan,
                           pem_stream_t stream);
int pem_range_hist_f64_dev(size_t n, int m, const double* data, size_t ld, int nr, const uint64_t* klo, const uint64_t* khi,
                           int bins, uint32_t* hist, pem_stream_t stream);
/* One level's decision, on the device: for each of n_ranges ranges the bin of hist[range][bins] (the all-reduced counts) that
 * holds resid[range], then klo / khi <- the keys of that bin and resid <- the rank inside it; ranges with klo >= khi are left. */
int pem_range_narrow_dev(int n_ranges, int bins, const uint32_t* hist, uint64_t* klo, uint64_t* khi, int64_t* resid,
                         pem_stream_t stream);

/* The sharded selection on the single-GPU selection's passes (round 3; the percentiles of scripts/gen_data.py:163-168 and
 * scripts/pem_v0/monte_carlo.py:363-658 over samples that live on several GPUs).  Every rank runs the same stages over its own
 * rows; hallthrusterpem_amd/percentiles.py all-reduces kmin / kmax / has_nan (MIN / MAX), hist1 and hist2 (SUM) between them and
 * all-gathers the padded candidate lists, so that every rank takes the same decisions: four streaming passes instead of the
 * eleven levels of pem_range_hist.  All arrays on the device, caller-owned; nt = 2 nq targets per column (the two order
 * statistics of each of nq <= PEM_QUANTILE_MAX_Q quantiles); nothing here synchronises the stream.
 *   pem_qsel_bins      bins1 / bins2 the histograms use for m columns and nt targets (pure arithmetic: LDS-bound powers of two)
 *   pem_qsel_minmax    kmin / kmax [m]: smallest / largest key of each column (kmin > kmax: no value), has_nan [m]
 *   pem_qsel_hist1     hist1[m][bins1] over [kmin, kmax] (the all-reduced ones): bin = floor(d mult / 2^32), d = (k - kmin) >> shift,
 *                      shift the smallest with (kmax - kmin) >> shift < 2^31, mult = min(2^32 - 1, floor(2^32 bins1 / (((kmax - kmin) >> shift) + 1)))
 *   pem_qsel_decide1   per (column, target): bin1 = the bin of the summed hist1 that holds resid (in: the wanted 0-based rank),
 *                      resid <- rank inside that bin; a column with kmin >= kmax is done at once (answer = kmin)
 *   pem_qsel_hist2     hist2[m][nt][bins2]: sub-bins (floor(frac bins2 / 2^32), frac = low word of d mult) of every target's bin1;
 *                      a bin shared by several targets of a column is counted under the first of them
 *   pem_qsel_decide2   bin2 from the summed hist2, resid <- rank inside it, count = values of all ranks in it, count_local = this
 *                      rank's (read from hist2_local, the rank's own counts)
 *   pem_qsel_compact   this rank's keys of every (bin1, bin2) into cand[m nt][list_len], padded with ~0; cursor[m nt] = values
 *                      offered (> list_len: the list overflowed); shared pairs are collected once, under the first target
 *   pem_qsel_select    answer[m nt] = the key of rank resid in the union of the `world` gathered lists (gathered[world][m nt][list_len]) */
int pem_qsel_bins(int m, int nt, int* bins1, int* bins2);
int pem_qsel_minmax_f64_dev(size_t n, int m, const double* data, size_t ld, uint64_t* kmin, uint64_t* kmax, int32_t* has_nan,
                            pem_stream_t stream);
int pem_qsel_hist1_f64_dev(size_t n, int m, const double* data, size_t ld, const uint64_t* kmin, const uint64_t* kmax, int bins1,
                           uint32_t* hist1, pem_stream_t stream);
int pem_qsel_decide1_dev(int m, int nt, const uint64_t* kmin, const uint64_t* kmax, const uint32_t* hist1, int bins1, uint64_t* resid,
                         int32_t* bin1, int32_t* done, uint64_t* answer, pem_stream_t stream);
int pem_qsel_hist2_f64_dev(size_t n, int m, const double* data, size_t ld, const uint64_t* kmin, const uint64_t* kmax, int nt,
                           const int32_t* bin1, int bins1, int bins2, uint32_t* hist2, pem_stream_t stream);
int pem_qsel_decide2_dev(int m, int nt, const uint32_t* hist2, const uint32_t* hist2_local, int bins2, const int32_t* bin1,
                         const int32_t* done, uint64_t* resid, int32_t* bin2, uint64_t* count, uint32_t* count_local, pem_stream_t stream);
int pem_qsel_compact_f64_dev(size_t n, int m, const double* data, size_t ld, const uint64_t* kmin, const uint64_t* kmax, int nt,
                             const int32_t* bin1, const int32_t* bin2, const int32_t* done, int bins1, int bins2, uint32_t list_len,
                             uint64_t* cand, uint32_t* cursor, pem_stream_t stream);
int pem_qsel_select_dev(int m, int nt, int world, uint32_t list_len, const uint64_t* gathered, const int32_t* bin1, const int32_t* bin2,
                        const int32_t* done, const uint64_t* resid, uint64_t* answer, pem_stream_t stream);

/* ---- fused Monte-Carlo evaluation -----------------------------------------------------------------
 * sample_inputs + predict of scripts/gen_data.py:238-239 in ONE launch: the 15 coupled inputs of global samples
 * first_index .. first_index+n-1 are generated in registers from the counter-based design (kind/a/b as for
 * pem_sample_f64_dev, input order P_b V_a T_e V_vac Pstar P_T mdot_a a_1 c0..c5 sigma_cex) and never touch HBM
 * unless x_out ([15][ld], ld >= n) is given.  Results are bit-identical to pem_sample_f64_dev followed by
 * pem_coupled_f64_dev (swap_dim selects the Saltelli block as there).  752 instead of 872 bytes per evaluation. */
int pem_coupled_mc_f64_dev(size_t n, uint64_t first_index, uint64_t seed, uint32_t stream_id, int swap_dim,
                           const int32_t* kind, const double* a, const double* b, double torr2pa, double radius,
                           double* x_out, size_t ld,
                           double* V_cc, double* I_B0, double* T, double* j_ion, double* div_angle, double* T_c,
                           uint8_t* invalid, pem_stream_t stream);

/* ---- single-precision arithmetic: reduced QoIs and the fused Saltelli design (csrc/pem_fp32.hip) ----------------------
 * SURVEY.md section 8b "fp32/mixed entry points optional with a tolerance report", section 8d config 5 (BASELINE
 * configs[4]: Sobol' sensitivity, "fp64 -> fp32 mixed with tolerance check").  The same formulas as pem_coupled_f64_dev
 * (cathode.py:24-38, tests/sim_hallthruster.jl:35-48, plume.py:39-140) evaluated in fp32 from fp32 tables; only the
 * scalar QoIs are produced.  The tolerance report (fp32 against fp64 on identical inputs, per QoI) is
 * hallthrusterpem_amd.fp32.compare_with_fp64 / bench.py --fp32 / tests/test_fp32.py.
 *   x:   [15][ld] floats, rows in the order P_b V_a T_e V_vac Pstar P_T mdot_a a_1 c0..c5 sigma_cex
 *   qoi: [3][ldq] floats: V_cc, div_angle, T_c;  invalid (optional): plume.py:105's flag per sample.                       */
int pem_coupled_f32_dev(size_t n, float torr2pa, float radius, const float* x, size_t ld, float* qoi, size_t ldq,
                        uint8_t* invalid, pem_stream_t stream);
/* The Saltelli design of scripts/pem_v0/sobol.py:46-118 (uqtils sobol_sa, third-party: parity UNPINNED) as ONE launch:
 * for base samples first_index .. first_index+n_base-1 the rows A (stream_id) and B (stream_id + 1) of the counter-based
 * design -- the same numbers as pem_sample_f64_dev, rounded to float -- are generated in registers, the fp32 model is
 * evaluated on A, B and on A with column varied[j] from B for j < n_varied, and the estimator sums are accumulated in
 * fp64: partial[n_blocks][2 + 2 n_varied][3] (rows: sum fA+fB, sum fA^2+fB^2, then per varied input sum fB (fAB-fA) and
 * sum (fA-fAB)^2; columns V_cc, div_angle, T_c), one deterministic partial per workgroup.  flags[n_blocks][2]: counts of
 * non-physical thruster results (thruster.py:490-493: T < 0 or I_B0 < 0) and of invalid plume samples over all
 * n_base (n_varied + 2) evaluations.  Nothing but the partial sums touches HBM.                                         */
int pem_saltelli_f32_dev(size_t n_base, uint64_t first_index, uint64_t seed, uint32_t stream_id, const int32_t* kind,
                         const double* a, const double* b, int n_varied, const int32_t* varied, float torr2pa,
                         float radius, double* partial, uint64_t* flags, int n_blocks, pem_stream_t stream);
/* The same launch around the fp64 model (lane per sample; the scalar stages and tables of the coupled kernel, bit-identical
 * to pem_coupled_f64_dev's reduced-QoI results for samples inside the table range), on the design itself, not rounded.  */
int pem_saltelli_f64_dev(size_t n_base, uint64_t first_index, uint64_t seed, uint32_t stream_id, const int32_t* kind,
                         const double* a, const double* b, int n_varied, const int32_t* varied, double torr2pa,
                         double radius, double* partial, uint64_t* flags, int n_blocks, pem_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PEM_HIP_H */
