"""Sparse-grid surrogate (BASELINE.json configs[3]).  amisc's surrogate is third-party (parity
unpinned): the HIP predict kernel is held to a numpy restatement of its own formula, the index-set logic to the
combination-technique identities, and the fitted surrogate to the true model."""
import itertools

import numpy as np
import pytest

from hallthrusterpem_amd.surrogate import SparseGridSurrogate


def test_combination_coefficients_identities():
    D = 4
    for level in (0, 1, 2, 3):
        I = [b for b in itertools.product(range(level + 1), repeat=D) if sum(b) <= level]
        c = SparseGridSurrogate.combination_coefficients(I)
        assert sum(c.values()) == 1                                  # constants are reproduced
        from math import comb
        for b in I:                                                  # classical Smolyak coefficients
            q = level - sum(b)
            assert c[b] == ((-1) ** q * comb(D - 1, q) if q <= D - 1 else 0)
    assert SparseGridSurrogate.combination_coefficients([(0, 0), (1, 0), (2, 0)]) == {(0, 0): 0, (1, 0): 0, (2, 0): 1}


def test_combination_delta_equals_the_difference_of_coefficient_sets():
    rng = np.random.default_rng(0)
    D = 4
    for _ in range(40):
        I = [(0,) * D]                                    # a random downward-closed set, grown one admissible index at a time
        for _ in range(rng.integers(1, 25)):
            cands = {b[:d] + (b[d] + 1,) + b[d + 1:] for b in I for d in range(D)} - set(I)
            cands = [c for c in cands if all(c[:d] + (c[d] - 1,) + c[d + 1:] in I for d in range(D) if c[d] > 0)]
            cand = cands[rng.integers(len(cands))]
            before = SparseGridSurrogate.combination_coefficients(I)
            after = SparseGridSurrogate.combination_coefficients(I + [cand])
            delta = SparseGridSurrogate.combination_delta(I, cand)
            assert {b: after[b] - before.get(b, 0) for b in after if after[b] - before.get(b, 0) != 0} == delta
            I.append(cand)


FIXED = {'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6, 'a_1': 0.01, 'sigma_cex': 55e-20, 'c4': 1e20, 'c5': 1e16}
VARIED = ('T_e', 'V_vac', 'Pstar', 'P_T', 'c0', 'c1', 'c2', 'c3')


@pytest.mark.gpu
def test_predict_kernel_matches_numpy_restatement():
    import torch
    from oracle import surrogate_np as snp
    s = SparseGridSurrogate(VARIED, FIXED)
    rng = np.random.default_rng(0)
    for beta in [(1, 0, 0, 0, 0, 0, 0, 0), (0, 2, 0, 0, 0, 0, 0, 0), (1, 1, 0, 0, 0, 0, 0, 0), (0, 0, 0, 0, 3, 0, 0, 0),
                 (1, 0, 0, 0, 0, 0, 0, 1), (1, 1, 0, 0, 0, 0, 0, 1), (2, 0, 0, 0, 1, 0, 0, 0), (0, 0, 0, 0, 2, 0, 0, 0),
                 (0, 0, 0, 0, 1, 0, 0, 0), (0, 0, 0, 0, 0, 0, 0, 1), (0, 1, 0, 0, 0, 0, 0, 0), (1, 0, 0, 0, 1, 0, 0, 0),
                 (0, 1, 0, 0, 0, 0, 0, 1)]:
        s._ensure_values(beta)
    I = sorted(s.values)
    I = [b for b in I if all((b[:d] + (b[d] - 1,) + b[d + 1:]) in I for d in range(len(b)) if b[d] > 0)]
    t = rng.uniform(-1, 1, (len(VARIED), 2000))
    t[:, 0] = 0.0                                                   # all nodes of level 0
    t[0, 1], t[1, 2], t[4, 3] = 1.0, -1.0, -np.cos(np.pi * 3 / 8)                 # exact node hits
    got = s.predict(torch.from_numpy(t).cuda(), index_set=I).cpu().numpy()
    want = snp.predict(I, s.combination_coefficients(I), s.values, t)
    assert np.max(np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)) < 1e-12
    # interpolation property: at the nodes of an activated full tensor grid the surrogate returns the model values
    grid = s._grid((1, 1, 0, 0, 0, 0, 0, 1))
    full = [b for b in itertools.product(range(2), repeat=8) if all(b[d] == 0 for d in (2, 3, 4, 5, 6))]
    for b in full:
        s._ensure_values(b)
    at_nodes = s.predict(torch.from_numpy(grid).cuda(), index_set=full).cpu().numpy()
    assert np.max(np.abs(at_nodes.T - s.values[(1, 1, 0, 0, 0, 0, 0, 1)])) < 1e-10


@pytest.mark.gpu
def test_adaptive_refinement_reduces_error_against_true_model():
    import torch
    from hallthrusterpem_amd.models import pem_v0_coupled
    s = SparseGridSurrogate(VARIED, FIXED)
    rng = np.random.default_rng(1)
    t = rng.uniform(-1, 1, (len(VARIED), 5000))
    x = {k: np.full(5000, v) for k, v in FIXED.items()}
    x.update(s.to_physical(t))
    truth = pem_v0_coupled({k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in x.items()}, profile=False)
    truth = torch.stack([truth[k] for k in s.qoi]).cpu().numpy()

    def rel_l2():
        p = s.predict(torch.from_numpy(t).cuda()).cpu().numpy()
        return np.linalg.norm(p - truth, axis=1) / np.linalg.norm(truth, axis=1)
    e0 = rel_l2()
    hist = s.refine(max_iter=12, num_refine=1000, seed=3)
    e1 = rel_l2()
    assert len(hist) == 12 and len(s.index_set) == 13 and s.model_evals > 13
    assert np.all(e1 < 0.2 * e0) and np.all(e1 < 3e-2)
    # the set stayed downward closed and the candidates are admissible forward neighbours
    I = set(s.index_set)
    for b in I:
        for d in range(len(b)):
            if b[d] > 0:
                assert b[:d] + (b[d] - 1,) + b[d + 1:] in I
    assert all(c not in I and s._admissible(c) for c in s.candidates)
    # V_cc does not depend on the plume coefficients: refinement in those dimensions never helps V_cc
    first = [h[0] for h in hist]
    assert any(b[1] > 0 or b[0] > 0 for b in first)


@pytest.mark.gpu
@pytest.mark.parametrize('n_out', [1, 2, 4, 5, 9, 16])
def test_predict_kernel_synthetic_tables_all_output_widths(n_out):
    """pem_sparse_predict_f64_dev through the C ABI on hand-made tables: 15 dimensions (the coordinate staging then needs more
    than 64 KB of LDS), every register width of the kernel (exact 1..4, guarded 8 / 16), grids up to 9 x 9 x 9 nodes, a
    point count that is not a multiple of the workgroup."""
    import ctypes as C
    import torch
    from hallthrusterpem_amd import _lib
    from oracle import surrogate_np as snp
    D, n = 15, 1000
    rng = np.random.default_rng(n_out)
    betas = [(0,) * D]
    for dims, levels in [((0,), (1,)), ((14,), (3,)), ((3, 9), (2, 1)), ((1, 2, 3), (1, 1, 1)), ((5, 6, 14), (3, 3, 3)),
                         ((0, 7, 13), (1, 3, 2)), ((10, 11), (3, 3))]:
        b = [0] * D
        for d, l in zip(dims, levels):
            b[d] = l
        betas.append(tuple(b))
    values = {b: rng.standard_normal((int(np.prod([snp.nodes(l).size for l in b])), n_out)) for b in betas}
    coefs = {b: float(c) for b, c in zip(betas, rng.integers(-3, 4, len(betas)))}
    coefs[betas[0]] = 1.0
    used = [b for b in betas if coefs[b] != 0]
    idx = np.zeros((len(used), 12), dtype=np.int32)                          # {n_active, first row, dims[5], levels[5]} (PEM_SURR_MAX_ACTIVE = 5)
    off = 0
    for i, b in enumerate(used):
        active = [d for d in range(D) if b[d] > 0]
        idx[i, 0], idx[i, 1] = len(active), off
        for a, d in enumerate(active):
            idx[i, 2 + a], idx[i, 7 + a] = d, b[d]
        off += values[b].shape[0]
    t = rng.uniform(-1, 1, (D, n))
    t[:, 0] = 0.0
    t[14, 1], t[5, 2] = 1.0, -np.cos(np.pi / 8)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()                   # noqa: E731
    d_idx, d_coef = dev(idx), dev(np.array([coefs[b] for b in used]))
    d_val, d_t = dev(np.concatenate([values[b] for b in used])), dev(t)
    out = torch.full((n_out, n + 7), np.nan, dtype=torch.float64, device='cuda')
    p = lambda x: C.c_void_p(x.data_ptr())                                               # noqa: E731
    _lib.check(_lib.load().pem_sparse_predict_f64_dev(n, D, len(used), p(d_idx), p(d_coef), p(d_val), n_out, p(d_t), n, p(out), n + 7, 3, 3,
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isnan(got[:, n:]).all()                                    # nothing written past the n points of a row
    want = snp.predict(used, coefs, values, t)
    assert np.max(np.abs(got[:, :n] - want) / np.abs(want).max(axis=1, keepdims=True)) < 1e-12
    assert _lib.load().pem_sparse_predict_f64_dev(n, 33, len(used), p(d_idx), p(d_coef), p(d_val), n_out, p(d_t), n, p(out), n + 7, 3, 3, None) != 0


@pytest.mark.gpu
def test_grid_values_times_coefficients_is_the_prediction():
    """pem_sparse_grid_values_f64_dev: every grid's interpolant on its own; a prediction is the combination-coefficient row times
    those values -- what `refine` builds all its trial index sets from."""
    import torch
    s = SparseGridSurrogate(VARIED, FIXED)
    s.refine(max_iter=8, num_refine=200, seed=5)
    I = list(s.index_set)
    t = torch.from_numpy(np.random.default_rng(2).uniform(-1, 1, (len(VARIED), 777))).cuda()
    gv = s.grid_values(t, I)
    assert gv.shape == (len(I), len(s.qoi), 777)
    c = s.combination_coefficients(I)
    row = torch.tensor([float(c[b]) for b in I], dtype=torch.float64, device='cuda')
    want = s.predict(t)
    got = (row @ gv.reshape(len(I), -1)).reshape(len(s.qoi), -1)
    assert float(((got - want).abs() / want.abs().max(dim=1, keepdim=True).values).max()) < 1e-12
    # the constant grid (beta = 0) returns the model value at the centre of the box for every point
    zero = I.index((0,) * len(VARIED))
    assert torch.equal(gv[zero], torch.from_numpy(s.values[(0,) * len(VARIED)][0]).cuda()[:, None].expand(-1, 777))


@pytest.mark.gpu
def test_grids_of_level_four_and_five_active_dimensions_match_numpy():
    """Round 4: a multi-index may have up to five active dimensions of level four (PEM_SURR_MAX_ACTIVE / _LEVEL; three and three
    before).  Tables that hold such grids against the numpy restatement, with the LDS the launch takes sized from the table."""
    import torch
    from oracle import surrogate_np as snp
    s = SparseGridSurrogate(VARIED, FIXED)
    rng = np.random.default_rng(3)
    want_grids = [(4, 0, 0, 0, 0, 0, 0, 0), (0, 0, 0, 0, 0, 0, 0, 4), (1, 1, 1, 1, 0, 0, 0, 0), (1, 1, 1, 1, 1, 0, 0, 0), (2, 1, 0, 1, 0, 0, 1, 1),
                  (0, 4, 0, 0, 1, 0, 0, 0), (1, 0, 0, 0, 0, 0, 0, 4), (0, 0, 3, 0, 0, 2, 0, 1), (1, 2, 0, 0, 0, 1, 0, 0)]
    for beta in want_grids:
        s._ensure_values(beta)
    I = sorted(s.values)
    t = rng.uniform(-1, 1, (len(VARIED), 600))
    t[:, 0] = 0.0
    t[0, 1], t[7, 2], t[1, 3] = 1.0, -1.0, -np.cos(np.pi * 5 / 16)           # exact hits of level-4 nodes
    coefs = {b: float(rng.integers(-2, 3)) or 1.0 for b in I}                  # any coefficients: the formula is linear in them
    idx, _, vals, nb, na, lv = s._build_tables(I, unit_coefficients=True)
    assert (na, lv) == (5, 4)
    gv = s.grid_values(torch.from_numpy(t).cuda(), I).cpu().numpy()            # [B][n_out][n]
    got = np.tensordot(np.array([coefs[b] for b in I]), gv, axes=1)
    want = snp.predict(I, coefs, s.values, t)
    assert np.max(np.abs(got - want) / np.abs(want).max(axis=1, keepdims=True)) < 1e-11
    # the C entry point refuses what it is not built for
    import ctypes as C
    from hallthrusterpem_amd import _lib
    p = lambda x: C.c_void_p(x.data_ptr())                                                             # noqa: E731
    td = torch.from_numpy(t).cuda()
    out = torch.empty((s.n_out, 600), dtype=torch.float64, device='cuda')
    coef = torch.ones(nb, dtype=torch.float64, device='cuda')
    assert _lib.load().pem_sparse_predict_f64_dev(600, 8, nb, p(idx), p(coef), p(vals), s.n_out, p(td), 600, p(out), 600, 6, 4, None) == 1
    assert _lib.load().pem_sparse_predict_f64_dev(600, 8, nb, p(idx), p(coef), p(vals), s.n_out, p(td), 600, p(out), 600, 5, 5, None) == 1


@pytest.mark.gpu
def test_field_surrogate_interpolates_latents_and_reconstructs_the_profile():
    """qoi with 'j_ion': the surrogate carries the field as the latent coefficients of its SVD map (yml:273-280, gen_data.py:261-294,
    fit_surr.py:101-133).  The node values are the fused evaluate-and-compress launch's; `predict` equals the numpy restatement on
    all 3 + r outputs; `predict_fields` -- one launch -- equals predict followed by the reconstruction kernel, and tracks the
    true model's profile."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    from oracle import surrogate_np as snp
    s = SparseGridSurrogate(VARIED, FIXED, qoi=('V_cc', 'div_angle', 'T_c', 'j_ion'), num_compress=500, compress_seed=1)
    r = s.compression.rank
    assert s.field == 'j_ion' and 1 <= r <= 13 and s.n_out == 3 + r and s.out_names[:3] == ('V_cc', 'div_angle', 'T_c')
    assert s.compression.relative_error <= 0.01                                  # reconstruction_tol of the YAML
    hist = s.refine(max_iter=80, num_refine=500, seed=0)
    assert len(hist) == 80
    rng = np.random.default_rng(5)
    n = 3000
    t = rng.uniform(-1, 1, (len(VARIED), n))
    td = torch.from_numpy(t).cuda()
    pred = s.predict(td)
    want = snp.predict(s.index_set, s.combination_coefficients(s.index_set), s.values, t[:, ::10])
    assert np.max(np.abs(pred[:, ::10].cpu().numpy() - want) / np.abs(want).max(axis=1, keepdims=True)) < 1e-12
    both = s.predict_fields(td)
    assert set(both) == {'V_cc', 'div_angle', 'T_c', 'j_ion', 'j_ion_latent'} and both['j_ion'].shape == (n, 91)
    for i, k in enumerate(('V_cc', 'div_angle', 'T_c')):
        assert torch.equal(both[k], pred[i])
    assert torch.equal(both['j_ion_latent'], pred[3:].T)
    rec = s.compression.reconstruct(pred[3:].T.contiguous())                     # pem_svd_reconstruct_f64_dev (fp64 MFMA) on the same latents
    assert torch.allclose(both['j_ion'], rec, rtol=1e-11, atol=0.0)
    # against the true model: the profile in its norm (log10), and the latents the true model compresses to
    x = {k: np.full(n, v) for k, v in FIXED.items()}
    x.update(s.to_physical(t))
    b = CoupledBatch(n, profile=True)
    b.set_inputs(x)
    b.run()
    torch.cuda.synchronize()
    lt, lp = torch.log10(b.j_ion), torch.log10(both['j_ion'])
    err = float(torch.linalg.norm(lp - lt) / torch.linalg.norm(lt))
    assert err < 0.03, err                                                       # (80 refinements; configs[3]'s 160 reach 0.008: tests/test_baseline_configs.py)
    # interpolation property on the latents: at the nodes of an active grid the surrogate returns the compressed true model
    beta = max(s.index_set, key=lambda b: sum(b))
    at = s.predict(torch.from_numpy(s._grid(beta)).cuda(), index_set=[b for b in s.index_set]).cpu().numpy().T
    full_rows = s.values[beta]
    assert np.max(np.abs(at - full_rows) / (np.abs(full_rows).max(axis=0) + 1e-300)) < 1e-9


@pytest.mark.gpu
def test_field_surrogate_with_more_latents_than_the_fused_launch_keeps():
    """A compression map of rank 10 (> PEM_FUSED_LATENT_MAX_RANK = 8): the node values come from run() + compress() over a stored
    profile instead of the fused launch, and everything downstream (13 outputs per node, predict, reconstruction) is unchanged."""
    import torch
    from hallthrusterpem_amd import _lib
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.compression import SVDCompression
    from oracle import surrogate_np as snp
    rng = np.random.default_rng(2)
    probe = SparseGridSurrogate(VARIED, FIXED, qoi=('V_cc', 'div_angle', 'T_c'))
    t0 = rng.uniform(-1, 1, (len(VARIED), 400))
    x = {k: np.full(400, v) for k, v in FIXED.items()}
    x.update(probe.to_physical(t0))
    b = CoupledBatch(400, profile=True)
    b.set_inputs(x)
    b.run()
    torch.cuda.synchronize()
    comp = SVDCompression(norm='log10', rank=10).fit(b.j_ion)
    assert comp.rank == 10 > _lib.FUSED_LATENT_MAX_RANK
    s = SparseGridSurrogate(VARIED, FIXED, qoi=('V_cc', 'div_angle', 'T_c', 'j_ion'), compression=comp)
    assert s.n_out == 13
    y = s._true_outputs(x, 400)
    assert torch.equal(y[:, 3:], comp.compress(b.j_ion)) and torch.equal(y[:, 0], b.qoi[0])
    s.refine(max_iter=12, num_refine=200, seed=0)
    t = rng.uniform(-1, 1, (len(VARIED), 500))
    td = torch.from_numpy(t).cuda()
    pred = s.predict(td)
    want = snp.predict(s.index_set, s.combination_coefficients(s.index_set), s.values, t)
    assert np.max(np.abs(pred.cpu().numpy() - want) / np.abs(want).max(axis=1, keepdims=True)) < 1e-12
    both = s.predict_fields(td)
    assert torch.allclose(both['j_ion'], comp.reconstruct(pred[3:].T.contiguous()), rtol=1e-11, atol=0.0)
