"""Per-sample tolerances for plume results on inputs far OUTSIDE the PEM-v0 priors.  TEST INFRASTRUCTURE.

Inside the priors every quantity is compared at the north_star's plain 1e-10 relative (tests/test_gpu_parity.py).  The
fuzz inputs (tools/fuzz_parity.py: negative amplitudes, negative densities, c0 outside [0, 1]) make two of the
reference's expressions cancel:

  j_ion[m] = X1 g1[m] + X2 g2[m] + j_cex          plume.py:99-102   X = base * A, g = exp(-(alpha_m / a)^2)
  cos_div  = num / den,  num, den = Simpson sums of (X1 g1 + X2 g2) cos(.) [sin(.)]       plume.py:117-124

A sum of terms t_i that are each reproduced to a relative accuracy tau is reproduced to tau * sum|t_i| -- relative to
the result that is tau * cond with cond = sum|t_i| / |sum t_i|.  No implementation (the reference's own included: numpy's
exp and libm's differ in the last bit) can do better than that, so a result is accepted when

  |got - want| <= 1e-10 |want|  +  TAU * sum|t_i|        (j_ion;   + the 1 - exp(-x) floor below)
  cos_div: relative error <= 4 eps + TAU * (cond_den + cond_num - 2) + 1.5 eps (u2_den + u2_num)
           (4 eps = 8 ulp of a number in [0.5, 1); u2 = the |term|-weighted mean of u^2 = (alpha / a)^2 over a Simpson sum: a term
           g = exp(-u^2) is reproduced to 1.5 eps u^2 at best -- first bullet below -- and a sum WITHOUT cancellation inherits the
           weighted mean of its terms' errors, which the TAU x (cond - 1) part does not cover: for a beam one grid step wide the
           terms that carry the sums have u^2 = 1 .. 5 and the two sums differ by 8 eps between correct evaluations -- nothing
           anywhere else, 2e-10 of a divergence angle of 0.003 rad (seed 5160, round 4: arccos at the pole); wide beams: +3 eps)

with the term sizes taken from the oracle (oracle_plume_terms_f64), per entry -- not a blanket tolerance for a class of
samples.  TAU = 3e-13 is the agreement two correct fp64 evaluations of one TERM can be held to here:
  * g = exp(-u^2): the reference rounds u = alpha/a and u^2 before exp(), which moves g by up to 1.5 eps u^2 relative
    (u^2 reaches 745 before exp() returns 0: 2.5e-13); the device path advances g by a recurrence that follows the
    exact value to ~6e-14 (CH^2/2 ulp) instead of repeating that rounding;
  * X = I_B0 exp(-x) A / r^2 with A from a different (equally accurate, 3e-16) normaliser method.
Without cancellation (cond = 1: every sample inside the priors) the TAU term is 300 times smaller than the 1e-10 term
and the rule is the plain relative one.

j_cex = I_B0 (1 - exp(-x)) / (2 pi r^2) is itself a difference: two exp() results one ulp apart move it by
eps * exp(-x) * I_B0 / (2 pi r^2) in absolute terms -- the `8 eps (1 + decay)` floor.

Where the reference's own result is noise the comparison says so instead of widening a tolerance:
  * beam amplitude in the denormal range (0 < |base| < 1e-280): both Simpson sums are a few denormal bits, their
    ratio is 0/0 or arbitrary on either side -> div_angle / T_c not compared (j_ion is);
  * both beams narrower than a quarter grid step (|a| < 0.0044): only the centreline point contributes, cos_div = 1 to
    the last bit and arccos gives 0 or NaN depending on that bit -> div_angle / T_c not compared;
  * |cos_div| within its own error bound of 1: arccos is 0-or-NaN on either side -> the NaN pattern of div_angle is not
    compared there (T_c = T cos_div still is).
"""
import numpy as np

TOL = 1e-10
TAU = 3e-13
EPS = float(np.finfo(np.float64).eps)
NANGLE = 91


def angle_grid():
    a = np.arange(NANGLE) * ((np.pi / 2) / 90.0)
    a[-1] = np.pi / 2
    return a


def _simpson_weights():
    """w with scipy.integrate.simpson(y, x=angle_grid()) == w @ y (the rule is linear in y)"""
    from scipy.integrate import simpson
    a = angle_grid()
    return np.array([simpson(np.eye(NANGLE)[m], x=a) for m in range(NANGLE)])


def plume_bounds(terms: dict, I_B0):
    """terms: oracle_ctypes.plume_terms(...).  Returns per-entry absolute slack for j_ion (n, 91, R), the condition
    numbers, the allowed relative error of cos_div (n, R) and the mask of (sample, radius) pairs whose div_angle / T_c are
    defined by normal-range arithmetic."""
    with np.errstate(all='ignore'):
        al = angle_grid()[None, :]
        g1 = np.exp(-(al / terms['a1'][:, None]) ** 2)[:, :, None]            # (n, 91, 1)
        g2 = np.exp(-(al / terms['a2'][:, None]) ** 2)[:, :, None]
        X1, X2, jc, decay = (terms[k][:, None, :] for k in ('X1', 'X2', 'j_cex', 'decay'))
        rr = terms['radii'][None, None, :]
        beams = np.abs(X1) * g1 + np.abs(X2) * g2                              # (n, 91, R)
        unit = np.abs(np.asarray(I_B0, dtype=np.float64))[:, None, None] / (2 * np.pi * rr ** 2)
        j_slack = TAU * beams + 8 * EPS * (1.0 + np.abs(decay)) * unit
        j_terms = beams + (1.0 + np.abs(decay)) * unit          # j_cex is itself the difference unit * (1 - decay)
        cond_den = terms['den_abs'] / np.abs(terms['den'])
        cond_num = terms['num_abs'] / np.abs(terms['num'])
        # the |term|-weighted mean of u^2 over the two Simpson sums (weights folded as plume.py:117-123 applies them: the profile
        # flipped against cos(alpha) [sin(alpha)], scipy's composite weights w)
        w = _simpson_weights()
        cden = (w * np.cos(angle_grid()))[::-1][None, :, None]
        cnum = (w * np.cos(angle_grid()) * np.sin(angle_grid()))[::-1][None, :, None]
        u1sq = ((al / terms['a1'][:, None]) ** 2)[:, :, None]
        u2sq = ((al / terms['a2'][:, None]) ** 2)[:, :, None]
        t1, t2 = np.abs(X1) * g1, np.abs(X2) * g2
        wsum = lambda c: np.nan_to_num(np.sum(np.abs(c) * (t1 * u1sq + t2 * u2sq), axis=1) /                       # noqa: E731
                                       np.sum(np.abs(c) * (t1 + t2), axis=1), nan=0.0, posinf=0.0)
        cos_rel = 4 * EPS + TAU * np.maximum(cond_den + cond_num - 2.0, 0.0) + 1.5 * EPS * (wsum(cden) + wsum(cnum))
        cos_rel = np.where(np.isfinite(cos_rel), cos_rel, np.inf)              # den == 0 or a NaN amplitude: nothing to hold
        base = np.asarray(I_B0)[:, None] * terms['decay'] / terms['radii'][None, :] ** 2
        denormal = (np.abs(base) < 1e-280) & (base != 0.0)
        unresolved = np.maximum(np.abs(terms['a1']), np.abs(terms['a2'])) < 0.0044
        comparable = ~denormal & ~unresolved[:, None]
    return {'j_slack': j_slack, 'j_terms': j_terms, 'cos_rel': cos_rel, 'cond_cos': np.maximum(cond_den, cond_num),
            'comparable': comparable, 'cos_div': terms['num'] / np.where(terms['den'] == 0, np.nan, terms['den'])}


def _same_special(g, w, what):
    assert np.array_equal(np.isnan(g), np.isnan(w)), f'NaN pattern differs: {what}'
    assert np.array_equal(np.isinf(g), np.isinf(w)) and np.array_equal(np.sign(g[np.isinf(g)]), np.sign(w[np.isinf(w)])), f'inf pattern differs: {what}'


def j_ion_error(got, want, bounds, what='j_ion'):
    """dict(err, cond, n_cancelling, tau_seen, worst): err = worst |got - want| / (|want| + slack / TOL) over the finite
    entries (<= TOL: every entry is inside its bound); cond = the largest condition number among the entries that needed
    more than the plain relative rule, n_cancelling their number, tau_seen the largest |got - want| / sum|t_i| among them
    (what TAU would have had to be); worst = flat index of the entry behind `err`."""
    n, _, R = bounds['j_slack'].shape
    g, w = np.asarray(got, dtype=np.float64).reshape(n, NANGLE, R), np.asarray(want, dtype=np.float64).reshape(n, NANGLE, R)
    _same_special(g, w, what)
    invalid_rows = np.all(w == 1e-20, axis=(1, 2))
    assert np.array_equal(np.all(g == 1e-20, axis=(1, 2)), invalid_rows), f'invalid rows differ: {what}'
    with np.errstate(all='ignore'):
        slack = np.where(invalid_rows[:, None, None], 0.0, bounds['j_slack'])      # the 1e-20 fill is exact
        fin = np.isfinite(w) & np.isfinite(slack)
        d = np.abs(g - w)
        e = np.where(fin, d / (np.abs(w) + slack / TOL + 1e-300), 0.0)
        needs = fin & (d > TOL * np.abs(w))
        cond = np.max(np.where(needs, bounds['j_terms'] / np.abs(w), 0.0), initial=0.0)
        tau_seen = np.max(np.where(needs, d / bounds['j_terms'], 0.0), initial=0.0)
    return {'err': float(np.max(e, initial=0.0)), 'cond': float(cond), 'n_cancelling': int(needs.sum()), 'tau_seen': float(tau_seen),
            'worst': int(np.argmax(e)) if e.size else -1}


def divergence_error(got_div, want_div, got_tc, want_tc, bounds, what='div_angle'):
    """div_angle = arccos(cos_div) and T_c = T cos_div against the bound on cos_div.  An angle passes if it is within TOL
    relative, or if its error maps back -- |cos(got) - cos(want)|, evaluated without cancellation -- into the allowed
    error of cos_div.  dict(err_div, err_tc, cond, n_cancelling, tau_seen): errors scaled so that <= TOL passes; cond =
    the largest condition number among the pairs that needed the bound; tau_seen = the largest relative T_c error per unit
    of (cond_den + cond_num - 2) among them."""
    ok = bounds['comparable']
    shape = ok.shape
    gd, wd = np.asarray(got_div, dtype=np.float64).reshape(shape), np.asarray(want_div, dtype=np.float64).reshape(shape)
    with np.errstate(all='ignore'):
        cosw = bounds['cos_div']
        dcos = np.abs(cosw) * bounds['cos_rel']                        # allowed absolute error of cos_div
        at_pole = np.abs(np.abs(cosw) - 1.0) <= dcos                   # arccos is 0-or-NaN on either side
        cmp_nan = ok & ~at_pole & np.isfinite(dcos)
    assert np.array_equal(np.isnan(gd[cmp_nan]), np.isnan(wd[cmp_nan])), f'NaN pattern differs: {what}'
    out = {'err_tc': 0.0, 'tau_seen': 0.0}
    with np.errstate(all='ignore'):
        d = np.abs(gd - wd)
        fin = ok & np.isfinite(wd) & np.isfinite(gd)
        rel = d / np.where(wd == 0, 1.0, np.abs(wd))
        # |cos(got) - cos(want)| = 2 |sin((got + want) / 2) sin((got - want) / 2)|, exact also at the pole (where it is d^2 / 2)
        in_bound = 2.0 * np.abs(np.sin(0.5 * (gd + wd)) * np.sin(0.5 * (gd - wd))) <= dcos
        excess = np.where(fin & ~in_bound, rel, 0.0)
        needed = fin & in_bound & (rel > TOL) & (bounds['cos_rel'] > 4 * EPS * 1.5)
        out['err_div'] = float(np.max(excess, initial=0.0))
        out['cond'] = float(np.max(np.where(needed, bounds['cond_cos'], 0.0), initial=0.0))
        out['n_cancelling'] = int(needed.sum())
        if got_tc is not None and want_tc is not None:
            gt, wt = np.asarray(got_tc, dtype=np.float64).reshape(shape), np.asarray(want_tc, dtype=np.float64).reshape(shape)
            cmp_t = ok & np.isfinite(bounds['cos_rel'])
            assert np.array_equal(np.isnan(gt[cmp_t]), np.isnan(wt[cmp_t])), f'NaN pattern differs: T_c ({what})'
            fin_t = cmp_t & np.isfinite(wt) & np.isfinite(gt)
            relt = np.abs(gt - wt) / (np.abs(wt) + 1e-300)
            out['err_tc'] = float(np.max(np.where(fin_t, relt / (1.0 + bounds['cos_rel'] / TOL), 0.0), initial=0.0))
            excess_cond = (bounds['cos_rel'] - 4 * EPS) / TAU
            out['tau_seen'] = float(np.max(np.where(fin_t & (excess_cond > 1.0), relt / excess_cond, 0.0), initial=0.0))
    return out
