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
