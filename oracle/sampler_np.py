"""numpy restatement of csrc/pem_sampler.hip (Philox4x32-10 counter-based MC / Latin-hypercube / Saltelli designs).
TEST INFRASTRUCTURE ONLY.  The reference delegates sampling to amisc/uqtils (absent): PARITY UNPINNED -- this file
restates the library's OWN formulas so the HIP sampler can be checked bit for bit (uniform/linear) or to an ulp
(log-uniform, normal).  Philox itself is pinned by the Random123 known-answer vectors in tests/test_sampler.py."""
import numpy as np
from scipy.special import ndtri

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
UNIFORM, LOGUNIFORM, NORMAL = 0, 1, 2
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c = [np.asarray(x, dtype=np.uint64) & _MASK for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k = [np.uint64(k0) & _MASK, np.uint64(k1) & _MASK]
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k[0]) & _MASK, p1 & _MASK, ((p0 >> np.uint64(32)) ^ c[3] ^ k[1]) & _MASK, p0 & _MASK]
        k = [(k[0] + np.uint64(W0)) & _MASK, (k[1] + np.uint64(W1)) & _MASK]
    return c


def u53(hi, lo):
    return (((hi >> np.uint64(5)) << np.uint64(26)) | (lo >> np.uint64(6))).astype(np.float64) * 2.0 ** -53


def transform(kind, a, b, u):
    if kind == LOGUNIFORM:
        return np.exp(2.302585092994045684 * (a + (b - a) * u))
    if kind == NORMAL:
        return a + b * ndtri(u)
    return a + (b - a) * u


def feistel_permute(i, n, half_bits, k0, k1, dim):
    i = np.asarray(i, dtype=np.uint64).copy()
    mask = np.uint64((1 << half_bits) - 1)
    hb = np.uint64(half_bits)
    todo = np.ones(i.shape, dtype=bool)
    while todo.any():
        x = i[todo]
        l, r = x >> hb, x & mask
        for rnd in range(4):
            f = philox4x32_10(r & _MASK, r >> np.uint64(32), dim, 0x4C485300 + rnd, k0, k1)
            t = l ^ (((f[1] << np.uint64(32)) | f[0]) & mask)
            l, r = r, t
        x = (l << hb) | r
        i[todo] = x
        todo[todo] = x >= np.uint64(n)
    return i


def sample(n, first, seed, stream, kind, a, b, mode='mc', n_total=0, swap_dim=-1):
    """[ndim][n] float64 design, element (d, i) for global sample index first + i."""
    ndim = len(kind)
    g = np.arange(first, first + n, dtype=np.uint64)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    half_bits = 1
    while mode == 'lhs' and (1 << (2 * half_bits)) < n_total:
        half_bits += 1
    out = np.empty((ndim, n))
    for d in range(ndim):
        st = stream + (1 if (swap_dim == -2 or swap_dim == d) else 0)
        r = philox4x32_10(g & _MASK, g >> np.uint64(32), d // 2, st, k0, k1)
        u = u53(r[0], r[1]) if d % 2 == 0 else u53(r[2], r[3])
        if mode == 'lhs':
            cell = feistel_permute(g, n_total, half_bits, k0, k1 ^ st, d)
            u = (cell.astype(np.float64) + u) / float(n_total)
        out[d] = transform(kind[d], a[d], b[d], u)
    return out
