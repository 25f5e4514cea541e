"""numpy restatement of the sparse-grid Lagrange prediction formula of csrc/pem_surrogate.hip.  TEST INFRASTRUCTURE ONLY
(the reference's surrogate is amisc: parity unpinned; this pins the HIP kernel to its own stated formula)."""
import itertools

import numpy as np


def nodes(level):
    if level == 0:
        return np.zeros(1)
    m = 2 ** level + 1
    x = -np.cos(np.pi * np.arange(m) / (m - 1))
    x[(m - 1) // 2] = 0.0
    return x


def lagrange_basis(level, t):
    """[m][n] values of the Lagrange cardinal polynomials of the level's nodes at points t (product formula)."""
    x = nodes(level)
    out = np.ones((x.size, t.size))
    for j in range(x.size):
        for k in range(x.size):
            if k != j:
                out[j] *= (t - x[k]) / (x[j] - x[k])
    return out


def predict(index_set, coefs, values, t):
    """f[n_out][n] = sum_beta c_beta sum_nodes Y_beta[node] prod_d l_{beta_d, j_d}(t_d); node order = itertools.product."""
    D, n = t.shape
    n_out = next(iter(values.values())).shape[1]
    f = np.zeros((n_out, n))
    for beta in index_set:
        c = coefs[beta]
        if c == 0:
            continue
        bases = [lagrange_basis(beta[d], t[d]) for d in range(D)]
        for node, js in enumerate(itertools.product(*[range(b.shape[0]) for b in bases])):
            w = np.ones(n)
            for d, j in enumerate(js):
                w = w * bases[d][j]
            f += c * values[beta][node][:, None] * w[None, :]
    return f
