"""Sampling-loop drivers.

`filter_outputs` is pinned: tests/golden/filter_outputs.npz holds what the reference's own `_filter_outputs`
(scripts/gen_data.py:125-174) returned for the stored inputs.  The samplers and the Sobol' estimator are third-party
in the reference (parity unpinned): on the GPU they are checked against the CPU oracle evaluated on the very same
device-generated design, with the estimator restated in numpy."""
import numpy as np
import pytest
import torch

from conftest import div_err, load_golden, rel_err
from hallthrusterpem_amd import drivers


def _golden_filter():
    g = load_golden('filter_outputs')
    ins = {k[3:]: g[k] for k in g if k.startswith('in_')}
    return g, ins


@pytest.mark.parametrize('as_torch', [False, True])
def test_filter_outputs_matches_reference(as_torch):
    g, ins = _golden_filter()
    data = {k: (torch.from_numpy(v) if as_torch else v) for k, v in ins.items()}
    for q, tag in ((1.5, '15'), (3.0, '30')):
        nan_idx, out_idx = drivers.filter_outputs(data, iqr_factor=q)
        assert sorted(nan_idx) == sorted(k[6:] for k in g if k.startswith(f'nan{tag}_'))     # coords / errors skipped
        for k in nan_idx:
            assert np.array_equal(np.asarray(nan_idx[k]), g[f'nan{tag}_{k}']), k
            assert np.array_equal(np.asarray(out_idx[k]), g[f'out{tag}_{k}']), k
    nan_idx, out_idx = drivers.filter_outputs(data, iqr_factor=1.5)
    drop = np.asarray(drivers.discard_mask(nan_idx, out_idx))
    assert drop.sum() == 2 and drop[5] and drop[13]                   # only the NaN samples
    drop_all = np.asarray(drivers.discard_mask(nan_idx, out_idx, discard_outliers=True))
    assert drop_all[[3, 77, 10, 12, 20]].all() and not drop_all[11]


@pytest.mark.gpu
def test_forward_uq_matches_oracle_on_device_design():
    from hallthrusterpem_amd import constants
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    from oracle import oracle_ctypes as oc
    n = 50_003
    res = drivers.forward_uq(n, seed=11, batch_size=16_384, keep_profile=True)      # several batches + ragged tail
    x = {k: res['x'][i].cpu().numpy() for i, k in enumerate(COUPLED_INPUTS)}
    want = oc.coupled(x, constants.TORR_2_PA)
    assert rel_err(res['V_cc'].cpu().numpy(), want['V_cc']) <= 1e-10
    assert rel_err(res['j_ion'].cpu().numpy(), want['j_ion']) <= 1e-10
    assert div_err(res['div_angle'].cpu().numpy(), want['div_angle']) <= 1e-10
    assert rel_err(res['T_c'].cpu().numpy(), want['T_c']) <= 1e-10
    assert np.array_equal(res['invalid'].cpu().numpy(), want['invalid'])
    # the launches of a shard may be dealt onto side streams: the same results as on the caller's stream alone (the default)
    one = drivers.forward_uq(n, seed=11, batch_size=16_384, keep_profile=True, streams=2)
    for key in ('V_cc', 'div_angle', 'T_c', 'j_ion', 'x', 'invalid'):
        assert torch.equal(one[key], res[key]), key
    l2 = drivers.forward_uq(20_000, seed=3, method='lhs', batch_size=4096, streams=2)
    l1 = drivers.forward_uq(20_000, seed=3, method='lhs', batch_size=4096)
    assert torch.equal(l1['x'], l2['x']) and torch.equal(l1['T_c'], l2['T_c'])
    # the design does not depend on batching or sharding
    quiet = drivers.forward_uq(n, seed=11, keep_profile=True, keep_inputs=False)       # one launch, the inputs never written
    assert 'x' not in quiet and all(torch.equal(quiet[k], res[k]) for k in quiet)
    whole = drivers.forward_uq(n, seed=11, batch_size=1 << 20)
    other = drivers.forward_uq(n, seed=11, batch_size=7_777)
    assert torch.equal(whole['x'], res['x']) and torch.equal(whole['T_c'], other['T_c'])
    assert torch.equal(whole['div_angle'], other['div_angle'])
    # (without a profile the divergence integrals come from tables: equal to the profile mode's within rounding)
    assert float(((whole['T_c'] - res['T_c']).abs() / res['T_c'].abs()).max()) < 1e-12
    parts = [drivers.forward_uq(n, seed=11, rank=r, world=3) for r in range(3)]
    assert torch.equal(torch.cat([p['V_cc'] for p in parts]), whole['V_cc'])
    lhs = drivers.forward_uq(4096, seed=3, method='lhs')
    va = ((lhs['x'][1] - 200) / 200 * 4096).floor().long().sort().values
    assert torch.equal(va, torch.arange(4096, device=va.device))


@pytest.mark.gpu
def test_generate_data_layout_and_masks():
    d = drivers.generate_data_on_device(20_000, seed=5, description='compression')
    samples, outputs = d['compression']
    assert set(samples) >= {'P_b', 'c0', 'sigma_cex'} and outputs['j_ion'].shape == (20_000, 91)
    assert d['iqr_factor'] == 1.5 and set(d['nan_idx']) == set(outputs) == set(d['outlier_idx'])
    assert not any(bool(v.any()) for v in d['nan_idx'].values())             # the priors never produce NaN
    host = {k: v.cpu().numpy() for k, v in outputs.items()}
    host['j_ion'] = np.log10(host['j_ion'])
    nan_h, out_h = drivers.filter_outputs(host)
    for k in out_h:                                                          # device masks == numpy masks
        assert np.array_equal(d['outlier_idx'][k].cpu().numpy(), out_h[k]), k


@pytest.mark.gpu
def test_sobol_indices_against_numpy_estimator():
    from hallthrusterpem_amd import constants, sampling
    from hallthrusterpem_amd.models.coupled import COUPLED_INPUTS
    from oracle import oracle_ctypes as oc
    from oracle import sampler_np as snp
    N = 20_000
    fixed = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6}                      # operating point held, as sobol.py:104
    got = drivers.sobol_indices(N, seed=9, fixed=fixed, batch_size=8192)
    assert got['inputs'] == [k for k in COUPLED_INPUTS if k not in fixed] and got['evaluations'] == N * 14
    # restate on the host: same design (numpy Philox), oracle as the model, same estimators
    pri = dict(sampling.PEM_V0_PRIORS)
    for k, v in fixed.items():
        pri[k] = sampling.Prior(sampling.UNIFORM, v, v, 'fixed')
    kind = [pri[k].kind for k in COUPLED_INPUTS]
    a, b = [pri[k].a for k in COUPLED_INPUTS], [pri[k].b for k in COUPLED_INPUTS]

    def f(swap):
        x = snp.sample(N, 0, 9, 0, kind, a, b, swap_dim=swap)
        o = oc.coupled({k: x[i] for i, k in enumerate(COUPLED_INPUTS)}, constants.TORR_2_PA)
        return np.stack([o['V_cc'], o['div_angle'], o['T_c']], axis=1)
    fA, fB = f(-1), f(-2)
    pooled = np.concatenate([fA, fB])
    var = pooled.var(axis=0)
    for j, name in enumerate(got['inputs']):
        fAB = f(COUPLED_INPUTS.index(name))
        s1 = (fB * (fAB - fA)).mean(0) / var
        st = ((fA - fAB) ** 2).mean(0) / (2 * var)
        for i, q in enumerate(('V_cc', 'div_angle', 'T_c')):
            assert float(got['S1'][q][j]) == pytest.approx(s1[i], abs=2e-9)
            assert float(got['ST'][q][j]) == pytest.approx(st[i], abs=2e-9)
    # structure of the model: V_cc only sees the cathode inputs; the divergence angle only the plume inputs
    idx = {k: j for j, k in enumerate(got['inputs'])}
    assert float(got['ST']['V_cc'][idx['c2']]) == 0.0 and float(got['ST']['V_cc'][idx['a_1']]) == 0.0
    assert float(got['ST']['div_angle'][idx['T_e']]) == 0.0
    assert float(got['S1']['V_cc'][idx['V_vac']]) > 0.5
    assert 0.9 < float(sum(got['S1']['V_cc'])) < 1.1


@pytest.mark.gpu
def test_fused_monte_carlo_equals_sample_then_evaluate():
    """pem_coupled_mc_f64_dev: inputs generated inside the kernel == Design.fill + run, bit for bit, for any slice."""
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.sampling import Design
    design = Design(seed=321, stream=2)
    for n, first in ((1, 0), (63, 5), (64, 0), (4097, 10 ** 12), (100_003, 777)):
        ref = CoupledBatch(n)
        design.fill(ref.inputs, first_index=first)
        ref.run()
        fused = CoupledBatch(n)
        fused.inputs.zero_()
        fused.run_mc(design, first_index=first, write_inputs=True)
        torch.cuda.synchronize()
        assert torch.equal(fused.inputs, ref.inputs)
        for k in ('V_cc', 'div_angle', 'T_c', 'I_B0', 'T', 'j_ion', 'invalid'):
            assert torch.equal(fused.outputs()[k], ref.outputs()[k]), (n, k)
        quiet = CoupledBatch(n, profile=False)
        quiet.inputs.fill_(-1.0)
        quiet.run_mc(design, first_index=first)                 # reduced QoIs, inputs never written
        torch.cuda.synchronize()
        plain = CoupledBatch(n, profile=False)
        plain.inputs.copy_(ref.inputs)
        plain.run()
        torch.cuda.synchronize()
        assert torch.equal(quiet.qoi, plain.qoi) and bool((quiet.inputs == -1.0).all())
        assert float(((quiet.qoi - ref.qoi).abs() / ref.qoi.abs().clamp_min(1e-300)).max()) < 1e-11


@pytest.mark.gpu
def test_rejection_of_plume_spikes():
    from hallthrusterpem_amd.models.plume import current_density
    n = 50_000
    thr = 20.0      # the PEM-v0 priors rarely reach the reference's 200 A/m^2; a lower bar exercises the redraw loop
    x, rounds = drivers.sample_plume_without_spikes(n, seed=13, threshold=thr)
    assert x.shape == (8, n) and rounds >= 2
    ins = {k: x[i] for i, k in enumerate(drivers.PLUME_INPUTS)}
    ins['I_B0'] = torch.full((n,), 4.0, dtype=torch.float64, device='cuda')
    j = current_density(ins)['j_ion']
    assert float(j.max()) < thr
    raw = drivers.sampling.Design(names=drivers.PLUME_INPUTS, seed=13).sample(n)
    kept = (x == raw).all(dim=0)
    assert 0.5 < float(kept.float().mean()) < 1.0                    # accepted first draws are untouched
    again, _ = drivers.sample_plume_without_spikes(n, seed=13, threshold=thr)
    same, one = drivers.sample_plume_without_spikes(n, seed=13, threshold=1e9)
    assert one == 1 and torch.equal(same, raw)
    assert torch.equal(again, x)                                     # deterministic


@pytest.mark.gpu
def test_config1_cathode_lhs_design_on_device():
    """BASELINE.json configs[0] end to end on the device: a 1e4-sample Latin-hypercube design over the ranges of
    tests/test_cathode.py:19-21, generated by the HIP sampler, through cathode_coupling, against the oracle."""
    from hallthrusterpem_amd import constants
    from hallthrusterpem_amd.models import cathode_coupling
    from hallthrusterpem_amd.sampling import LOGUNIFORM, UNIFORM, Design, Prior
    from oracle import oracle_ctypes as oc
    pri = {'P_b': Prior(LOGUNIFORM, -8.0, -4.0, 'test_cathode.py:19'), 'V_a': Prior(UNIFORM, 200.0, 400.0, ':19'),
           'T_e': Prior(UNIFORM, 1.0, 5.0, ':20'), 'V_vac': Prior(UNIFORM, 0.0, 60.0, ':20'),
           'Pstar': Prior(UNIFORM, 10e-6, 100e-6, ':21'), 'P_T': Prior(UNIFORM, 10e-6, 100e-6, ':21')}
    d = Design(priors=pri, names=tuple(pri), seed=0)
    n = 10_000
    x = d.sample(n, method='lhs', n_total=n)
    for i, k in enumerate(pri):                                   # one sample per stratum in every dimension
        u = (torch.log10(x[i]) - pri[k].a) / (pri[k].b - pri[k].a) if pri[k].kind == LOGUNIFORM else (x[i] - pri[k].a) / (pri[k].b - pri[k].a)
        cells = (u * n).floor().long().clamp_(0, n - 1).sort().values
        assert int((cells != torch.arange(n, device=cells.device)).sum()) <= 2       # log10/10^ round trip at cell edges
    got = cathode_coupling(d.as_dict(x))['V_cc']
    xh = x.cpu().numpy()
    want = oc.cathode(*[xh[i] for i in range(6)], constants.TORR_2_PA)
    assert rel_err(got.cpu().numpy(), want) <= 1e-10
    assert float(got.min()) >= 0 and float(got.max()) <= 100


def _sobol_rank(rank, world, port, n_base, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        res = drivers.sobol_indices(n_base, seed=5, fixed={'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6}, batch_size=4096)
        np.savez(f'{out_dir}/rank{rank}.npz', **{f'{a}_{k}': v.cpu().numpy() for a in ('S1', 'ST') for k, v in res[a].items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sobol_indices_sharded_over_two_ranks(tmp_path):
    """Sobol' needs only per-rank partial sums -> one all-reduce of O(d n_qoi) doubles.  Two ranks
    (gloo collectives, both on this box's one GPU) each evaluate half of the base samples; every rank ends with the
    indices of the single-process run (sums are reassociated across ranks: equal to rounding)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    n_base = 20_001
    mp.spawn(_sobol_rank, args=(2, port, n_base, str(tmp_path)), nprocs=2, join=True)
    one = drivers.sobol_indices(n_base, seed=5, fixed={'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6}, batch_size=4096)
    for r in range(2):
        got = np.load(tmp_path / f'rank{r}.npz')
        for a in ('S1', 'ST'):
            for k, v in one[a].items():
                assert np.allclose(got[f'{a}_{k}'], v.cpu().numpy(), rtol=1e-9, atol=1e-12), (r, a, k)


def _sharded_rank(rank, world, port, n_total, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from hallthrusterpem_amd.distributed import evaluate_sharded
        from hallthrusterpem_amd.models import pem_v0_coupled
        from hallthrusterpem_amd.sampling import Design
        design = Design(seed=99)

        def make_inputs(lo, hi):                      # counter-based: the shard [lo, hi) of the one global design
            return design.as_dict(design.sample(hi - lo, first_index=lo))

        gathered, local = evaluate_sharded(n_total, make_inputs, lambda x: pem_v0_coupled(x, profile=False))
        assert all(v.is_cuda for v in gathered.values())
        np.savez(f'{out_dir}/rank{rank}.npz', **{k: v.cpu().numpy() for k, v in gathered.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_evaluate_sharded_with_the_device_path_on_two_and_three_ranks(tmp_path):
    """`distributed.evaluate_sharded` end to end on the GPU box: every rank draws its shard of one counter-based design on
    the device, evaluates it with the HIP kernels and all-gathers the QoIs (CUDA tensors through gloo here, RCCL on a
    multi-GPU node) -- the gathered arrays equal a single-process evaluation bit for bit, for even and ragged shards."""
    import socket
    import torch.multiprocessing as mp
    from hallthrusterpem_amd.models import pem_v0_coupled
    from hallthrusterpem_amd.sampling import Design
    for world, n_total in ((2, 10_000), (3, 10_001)):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        out = tmp_path / f'w{world}'
        out.mkdir()
        mp.spawn(_sharded_rank, args=(world, port, n_total, str(out)), nprocs=world, join=True)
        design = Design(seed=99)
        one = pem_v0_coupled(design.as_dict(design.sample(n_total)), profile=False)
        for r in range(world):
            got = np.load(out / f'rank{r}.npz')
            for k in ('V_cc', 'div_angle', 'T_c'):
                assert np.array_equal(got[k], one[k].cpu().numpy(), equal_nan=True), (world, r, k)
