"""BASELINE.json's configs at their STATED sizes on one MI355X (the driver-run suite; the smaller cases elsewhere in
tests/ cover the corners, these cover the sizes):

  configs[0]  1e4 Latin-hypercube cathode samples            -> tests/test_gpu_parity.py::test_cathode_config1_lhs
  configs[1]  1e6 Monte-Carlo plume samples, fp64            -> tests/test_gpu_parity.py::test_config2_full_size_properties
  configs[2]  1e7 coupled forward UQ                         -> test_config_1e7_coupled_forward_uq (here: the whole campaign
              in ONE launch on one GPU, held to the oracle shard by shard, and equal to the 8-shard evaluation bit for bit)
  configs[3]  5e5 candidate evaluations + batched predict    -> test_config_5e5_candidates_and_batched_predict
  configs[4]  2e7-evaluation Saltelli design incl. the thruster QoI post-process, fp64 -> fp32 with a tolerance check
                                                             -> test_config_2e7_saltelli_design
"""
import numpy as np
import pytest

from conftest import div_err, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def test_config_1e7_coupled_forward_uq():
    import torch
    from hallthrusterpem_amd import constants
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.distributed import shard_bounds
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    from hallthrusterpem_amd.sampling import Design
    from oracle import oracle_ctypes as oc
    oc.set_threads(16)
    n, world = 10_000_000, 8
    batch = CoupledBatch(n)                                     # 8.7 GB of results + 1.2 GB of inputs, resident
    Design(seed=2).fill(batch.inputs)                           # the counter-based PEM-v0 prior design, global indices 0 .. n-1
    batch.run()
    torch.cuda.synchronize()
    out = batch.outputs()
    shard = CoupledBatch(shard_bounds(n, world, 0)[1])
    worst = {}
    for r in range(world):                                      # the eight contiguous shards of the multi-GPU layout
        lo, hi = shard_bounds(n, world, r)
        x = {k: batch.inputs[i, lo:hi].cpu().numpy() for i, k in enumerate(COUPLED_INPUTS)}
        want = oc.coupled(x, constants.TORR_2_PA)
        got = {k: out[k][lo:hi].cpu().numpy() for k in ('V_cc', 'I_B0', 'T', 'j_ion', 'div_angle', 'T_c', 'invalid')}
        assert np.array_equal(got['invalid'], want['invalid'])
        for k in ('V_cc', 'I_B0', 'T', 'j_ion', 'T_c'):
            worst[k] = max(worst.get(k, 0.0), rel_err(got[k], want[k]))
        worst['div_angle'] = max(worst.get('div_angle', 0.0), div_err(got['div_angle'], want['div_angle']))
        # the same shard evaluated on its own -- what rank r of an 8-GPU run computes -- is the same bits
        shard.inputs.copy_(batch.inputs[:, lo:hi])
        shard.run()
        torch.cuda.synchronize()
        for k, v in shard.outputs().items():
            assert torch.equal(v, out[k][lo:hi]), (r, k)
    assert all(v <= RTOL for v in worst.values()), worst
    # ... and so is a range launch inside the big batch (what the chunked multi-GPU pipeline issues)
    before = batch.j_ion[1_250_048:1_250_048 + 4096].clone()
    batch.j_ion[1_250_048:1_250_048 + 4096].zero_()
    batch.run(first=1_250_048, count=4096)
    torch.cuda.synchronize()
    assert torch.equal(batch.j_ion[1_250_048:1_250_048 + 4096], before)


def test_config_5e5_candidates_and_batched_predict():
    """BASELINE configs[3]: fit_surr.py's adaptive surrogate training + 5e5 candidate evaluations + the batched tensor-interpolant
    predict -- on what the reference trains (round 4): the scalar QoIs AND the latent coefficients of j_ion's SVD map
    (pem_v0_SPT-100.yml:273-280: svd, log10, reconstruction_tol 0.01; gen_data.py:261-294; fit_surr.py:101-133), the profile
    reconstructed inside the predict launch."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.surrogate import SparseGridSurrogate
    from oracle import surrogate_np as snp
    fixed = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6, 'a_1': 0.01, 'sigma_cex': 55e-20, 'c4': 1e20, 'c5': 1e16}
    varied = ('T_e', 'V_vac', 'Pstar', 'P_T', 'c0', 'c1', 'c2', 'c3')
    s = SparseGridSurrogate(varied, fixed, qoi=('V_cc', 'div_angle', 'T_c', 'j_ion'))
    assert s.compression.relative_error <= 0.01 and s.n_out == 3 + s.compression.rank
    hist = s.refine(max_iter=160, num_refine=1000, seed=0)       # fit_surr.py:111's num_refine
    assert len(hist) == 160 and len(s.index_set) == 161
    n = 500_000
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    t = torch.rand((len(varied), n), dtype=torch.float64, device='cuda', generator=g) * 2 - 1
    y = s.predict_fields(t)                                      # the batched interpolant predict + reconstruction, all 5e5 points
    pred = torch.cat([torch.stack([y[k] for k in s.scalars]), y['j_ion_latent'].T])
    # (1) the kernel against the numpy restatement of its formula, on every 250th point -- scalars and latents
    sub = slice(0, n, 250)
    want = snp.predict(s.index_set, s.combination_coefficients(s.index_set), s.values, t[:, sub].cpu().numpy())
    got = pred[:, sub].cpu().numpy()
    assert np.max(np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)) < 1e-11
    # ... and the reconstruction against numpy: j_ion = 10^(latent @ basis^T)
    basis = s.compression.basis.cpu().numpy()
    want_j = 10.0 ** (want[3:].T @ basis.T)
    assert np.allclose(y['j_ion'][sub].cpu().numpy(), want_j, rtol=1e-9, atol=0.0)
    # (2) the 5e5 candidate evaluations through the true model, and the surrogate's error against them: scalars to 1e-3,
    #     the profile to the reconstruction tolerance of its compression, in its norm (log10)
    x = {k: np.full(n, v) for k, v in fixed.items()}
    x.update(s.to_physical(t.cpu().numpy()))
    batch = CoupledBatch(n, profile=True)
    batch.set_inputs(x)
    batch.run()
    torch.cuda.synchronize()
    err = (torch.linalg.norm(pred[:3] - batch.qoi, dim=1) / torch.linalg.norm(batch.qoi, dim=1)).cpu().numpy()
    assert np.all(err < 1e-3), err
    lt = torch.log10(batch.j_ion)
    err_j = float(torch.linalg.norm(torch.log10(y['j_ion']) - lt) / torch.linalg.norm(lt))
    assert err_j <= 0.01, err_j
    assert not bool(batch.invalid.any())


def test_config_2e7_saltelli_design():
    import torch
    from hallthrusterpem_amd import constants, drivers, sampling
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    from hallthrusterpem_amd.models.thruster import check_thruster_outputs
    from oracle import oracle_ctypes as oc
    from oracle import sampler_np as snp
    oc.set_threads(16)
    fixed = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6}                      # operating point held, as sobol.py:104
    n_base = 1_428_572                                                       # x (12 + 2) = 20,000,008 evaluations
    full = drivers.sobol_indices(n_base, seed=1, fixed=fixed)                 # one fused launch, fp64 model
    assert full['fused'] and full['non_physical'] == 0 and full['invalid'] == 0
    assert full['evaluations'] == n_base * 14 >= 20_000_000 and len(full['inputs']) == 12
    for q in ('V_cc', 'div_angle', 'T_c'):
        assert torch.isfinite(full['S1'][q]).all() and torch.isfinite(full['ST'][q]).all()
        assert 0.9 < float(full['S1'][q].sum()) <= 1.05 and float(full['ST'][q].min()) >= 0.0
    idx = {k: j for j, k in enumerate(full['inputs'])}
    assert float(full['ST']['V_cc'][idx['c2']]) == 0.0 and float(full['ST']['div_angle'][idx['T_e']]) == 0.0

    # the thruster QoI post-process (the two filters of thruster.py:490-502) over the A and B blocks of the design
    pri = dict(sampling.PEM_V0_PRIORS)
    for k, v in fixed.items():
        pri[k] = sampling.Prior(sampling.UNIFORM, v, v, 'fixed')
    design = sampling.Design(priors=pri, seed=1)
    b = CoupledBatch(n_base, profile=False)
    for swap in (-1, -2):
        b.run_mc(design, first_index=0, swap_dim=swap, write_inputs=True)
        bad = check_thruster_outputs({'T': b.T, 'I_B0': b.I_B0})
        want_bad = (b.T < 0) | (b.I_B0 < 0)
        assert torch.equal(bad, want_bad) and int(bad.sum()) == 0            # the priors keep V_a > V_cc and mdot_a > 0
        assert not bool(b.invalid.any())

    # the estimator against the numpy restatement (numpy Philox design, oracle as the model) on a 1.12e6-evaluation sub-design
    N = 80_000
    got = drivers.sobol_indices(N, seed=1, fixed=fixed, batch_size=1 << 16, fused=False)     # the block-by-block driver
    fused_sub = drivers.sobol_indices(N, seed=1, fixed=fixed)
    kind = [pri[k].kind for k in COUPLED_INPUTS]
    lo, hi = [pri[k].a for k in COUPLED_INPUTS], [pri[k].b for k in COUPLED_INPUTS]

    def f(swap):
        x = snp.sample(N, 0, 1, 0, kind, lo, hi, swap_dim=swap)
        o = oc.coupled({k: x[i] for i, k in enumerate(COUPLED_INPUTS)}, constants.TORR_2_PA)
        return np.stack([o['V_cc'], o['div_angle'], o['T_c']], axis=1)
    fA, fB = f(-1), f(-2)
    var = np.concatenate([fA, fB]).var(axis=0)
    for j, name in enumerate(got['inputs']):
        fAB = f(COUPLED_INPUTS.index(name))
        s1, st = (fB * (fAB - fA)).mean(0) / var, ((fA - fAB) ** 2).mean(0) / (2 * var)
        for i, q in enumerate(('V_cc', 'div_angle', 'T_c')):
            assert float(got['S1'][q][j]) == pytest.approx(s1[i], abs=2e-9)
            assert float(got['ST'][q][j]) == pytest.approx(st[i], abs=2e-9)
            assert float(fused_sub['S1'][q][j]) == pytest.approx(s1[i], abs=2e-9) and float(fused_sub['ST'][q][j]) == pytest.approx(st[i], abs=2e-9)
            # and the full design agrees with its own first 80 000 base samples to Monte-Carlo accuracy
            assert float(full['ST'][q][j]) == pytest.approx(st[i], abs=0.02)

    # "fp64 -> fp32 ... with tolerance check", arithmetic: the same 2e7-evaluation design through the fp32 model in one
    # fused launch (tests/test_fp32.py holds the model's per-QoI tolerance report and the launch's exactness)
    f32 = drivers.sobol_indices(n_base, seed=1, fixed=fixed, precision='fp32')
    assert f32['evaluations'] == full['evaluations'] and f32['non_physical'] == 0 and f32['invalid'] == 0
    for q in ('V_cc', 'div_angle', 'T_c'):
        assert float((f32['S1'][q] - full['S1'][q]).abs().max()) < 2e-4 and float((f32['ST'][q] - full['ST'][q]).abs().max()) < 2e-4

    # ... and storage: the same fp64 arithmetic, profile stored as fp32, on identical inputs
    n = 2_000_000
    f64, mix = CoupledBatch(n), CoupledBatch(n, mixed=True)
    sampling.Design(seed=2).fill(f64.inputs)
    mix.inputs.copy_(f64.inputs)
    f64.run()
    mix.run()
    torch.cuda.synchronize()
    assert torch.equal(f64.qoi, mix.qoi)
    rel = ((mix.j_ion.double() - f64.j_ion) / f64.j_ion).abs()
    assert float(rel.max()) <= 2.0 ** -24 * (1 + 1e-6)
