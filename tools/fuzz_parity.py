#!/usr/bin/env python3
"""Differential fuzzing of the device path against the CPU oracle (test infrastructure) on inputs far outside the
priors: signs, zeros, huge / tiny magnitudes, NaN / inf, beams around every table boundary.  For each seed the coupled
evaluation runs in full, reduced and mixed mode, fused with the likelihood and with the SVD compression (against their
two-launch pipelines), the tile-interleaved input layout, and the plume alone with one, two, three, five, nine, 17 and 40 radii.

NaN / inf / invalid patterns must agree exactly.  Finite values are held to the north_star's 1e-10 relative, and where a
result is a cancelling sum (negative amplitudes or densities, c0 outside [0, 1] -- none of which the priors reach) to
the PER-ENTRY bound of tests/parity_rules.py: 1e-10 of the result plus TAU = 3e-13 of the sum of the magnitudes of its
terms, the terms taken from the oracle's decomposition of that very sample.  The report lists, per quantity, how many
entries needed the second part, the largest condition number among them and the largest error per unit of term size
that was seen (`tau seen`: what TAU would have had to be).  V_cc = V_vac + T_e ln(1 + PB/PT) - T_e PB / (PT + P*) is held
the same way against |V_vac| + |T_e ln(.)| + |T_e PB / (PT + P*)|.  The regimes where the reference's own result is noise
(denormal amplitudes, beams narrower than a quarter grid step, |cos_div| = 1 to rounding) are listed in parity_rules.py.

    python tools/fuzz_parity.py [--seeds 40] [--seed-list 867 940 1100] [--n 20000] [--dump worst.npz]
"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))


def wild(rng, n):
    def mix(base, *alts):
        out = base.copy()
        for frac, vals in alts:
            m = rng.random(n) < frac
            out[m] = vals[m] if isinstance(vals, np.ndarray) else vals
        return out
    u = rng.random((15, n))
    sgn = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    x = {'P_b': mix(10 ** (u[0] * 6 - 9), (0.02, 0.0), (0.02, -1e-5), (0.01, np.inf), (0.01, 1e3)),
         'V_a': mix(u[1] * 400 + 50, (0.02, 0.0), (0.02, -100.0)),
         'T_e': mix(u[2] * 8, (0.02, 0.0), (0.01, -2.0)),
         'V_vac': mix(u[3] * 120 - 30, (0.05, 0.0)),
         'Pstar': mix(10 ** (u[4] * 4 - 7), (0.02, 0.0), (0.01, -1e-5)),
         'P_T': mix(10 ** (u[5] * 4 - 7), (0.02, 0.0), (0.01, -1e-5)),
         'mdot_a': mix(u[6] * 1e-5, (0.02, 0.0), (0.01, -1e-6)),
         'a_1': mix(10 ** (u[7] * 3 - 3), (0.02, 0.5), (0.02, 0.2)),                      # eta_c = 1 - 2 a_1 = 0
         'c0': mix(u[8] * 1.4 - 0.2, (0.03, 0.0), (0.03, 1.0)),
         'c1': mix(u[9] * 1.2 - 0.1, (0.02, 0.0), (0.02, 1e-3), (0.01, -0.5)),
         'c2': mix((u[10] * 2 - 1) * 10 ** (rng.random(n) * 4 - 1), (0.1, 0.0)),
         # alpha1 around every switch of the kernel: 0, QA_MIN = 0.03, 0.25, pi/2 clip, and alpha2 = alpha1 / c1 up to the
         # erfi-overflow bound 53.28
         'c3': mix(10 ** (u[11] * 3.5 - 2.5) * np.where(rng.random(n) < 0.05, -1, 1), (0.02, 0.0), (0.03, 0.03), (0.03, 0.25),
                   (0.03, 0.0299999), (0.03, 0.2500001), (0.02, 1.5707963267948966), (0.02, 53.28349511409265 * 1e-3)),
         'c4': mix(10 ** (u[12] * 8 + 16), (0.02, 0.0), (0.01, -1e19)),
         'c5': mix(10 ** (u[13] * 8 + 10), (0.02, 0.0), (0.01, -1e15)),
         'sigma_cex': mix(u[14] * 1e-18, (0.02, 0.0), (0.01, -5e-19))}
    for k in x:
        m = rng.random(n) < 0.002
        x[k][m] = np.nan
    x['c3'] *= np.where(rng.random(n) < 0.02, sgn, 1.0)
    return x


def prior_campaign(batches, oc, constants, pem_v0_coupled, div_err, rel_err, n=1_250_000):
    """Many shards of BASELINE configs[2] (1.25e6 samples from the PEM-v0 priors each), full and reduced mode against the
    oracle at the plain 1e-10 tolerance: table boundaries, interval edges and rare corners of the prior box."""
    from _inputs import coupled_inputs
    worst = {}
    for b in range(batches):
        x = coupled_inputs(n, seed=5000 + b)
        want = oc.coupled(x, constants.TORR_2_PA)
        for name, got in (('full', pem_v0_coupled(x)), ('reduced', pem_v0_coupled(x, profile=False))):
            assert np.array_equal(got['invalid'], want['invalid'])
            for key in ('V_cc', 'I_B0', 'T', 'T_c', 'div_angle') + (('j_ion',) if name == 'full' else ()):
                g, w = np.asarray(got[key]).reshape(-1), np.asarray(want[key]).reshape(-1)
                e = div_err(g, w) if key == 'div_angle' else rel_err(g, w)
                worst[f'{name}.{key}'] = max(worst.get(f'{name}.{key}', 0.0), e)
    print(f'{batches} x {n} samples from the priors: worst relative errors vs the oracle')
    for key, v in sorted(worst.items()):
        print(f'  {key:20s} {v:.2e}')
        assert v <= 1e-10, key


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=40, help='seeds FIRST .. SEEDS-1')
    ap.add_argument('--first-seed', type=int, default=0, help='first seed of the range (campaigns longer than one GPU call are run in pieces)')
    ap.add_argument('--seed-list', type=int, nargs='*', default=[], help='further seeds (e.g. the ones earlier campaigns failed on)')
    ap.add_argument('--n', type=int, default=20_000)
    ap.add_argument('--time-budget', type=float, default=0.0, metavar='SECONDS',
                    help='stop starting new seeds after this long and summarise the seeds that ran (a GPU call has a hard limit; 0 = none)')
    ap.add_argument('--many-radii-samples', type=int, default=3000, help='samples per seed that also go through 9 / 17 / 40 sweep radii')
    ap.add_argument('--dump', default='', help='write the worst sample of every quantity (inputs, got, want) to this .npz')
    ap.add_argument('--priors', type=int, default=0, metavar='BATCHES',
                    help='instead: BATCHES x 1.25e6 samples drawn from the PEM-v0 priors, full and reduced mode, strict 1e-10')
    args = ap.parse_args()
    import torch
    import parity_rules as pr
    from conftest import div_err, rel_err
    from oracle import oracle_ctypes as oc
    from hallthrusterpem_amd import constants
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.models import current_density, pem_v0_coupled
    oc.set_threads(16)
    if args.priors:
        return prior_campaign(args.priors, oc, constants, pem_v0_coupled, div_err, rel_err)
    from hallthrusterpem_amd.compression import SVDCompression
    from hallthrusterpem_amd.likelihood import JionLikelihood
    frng = np.random.default_rng(7)
    alpha = np.sort(frng.uniform(-np.pi / 2, np.pi / 2, (7, 29)), axis=1)
    alpha[0, 0], alpha[0, -1] = -np.pi / 2, np.pi / 2
    yv = frng.lognormal(0.0, 2.0, (7, 29))
    lik = JionLikelihood(alpha, yv, 0.3 * yv + 0.01)
    comp = SVDCompression(norm='log10', rank=6)
    comp.basis = torch.from_numpy(np.ascontiguousarray(np.linalg.qr(frng.standard_normal((91, 6)))[0])).cuda()
    k = constants.TORR_2_PA

    worst, where, extra, dump = {}, {}, {}, {}
    seed_now = [0]

    def note(key, val, **info):
        assert val == val, f'{key}: the error metric itself is NaN'
        if float(val) > worst.get(key, -1.0):
            worst[key], where[key] = float(val), seed_now[0]
        e = extra.setdefault(key, {'cond': 0.0, 'n_cancelling': 0, 'tau_seen': 0.0})
        e['cond'] = max(e['cond'], info.get('cond', 0.0))
        e['tau_seen'] = max(e['tau_seen'], info.get('tau_seen', 0.0))
        e['n_cancelling'] += info.get('n_cancelling', 0)

    def same_pattern(a, b, what):
        assert np.array_equal(np.isnan(a), np.isnan(b)), f'NaN pattern differs: {what}'
        assert np.array_equal(np.isinf(a), np.isinf(b)) and np.array_equal(np.sign(a[np.isinf(a)]), np.sign(b[np.isinf(b)])), f'inf pattern differs: {what}'

    def plume_check(tag, got, want, bounds, seed, with_j=True, inputs=None):
        """one plume result (coupled or stand-alone) against the per-entry bounds"""
        if with_j:
            r = pr.j_ion_error(got['j_ion'], want['j_ion'], bounds, f'{tag} j_ion seed {seed}')
            before = worst.get(f'{tag}.j_ion', -1.0)
            note(f'{tag}.j_ion', r['err'], **r)
            if args.dump and r['err'] > before and inputs is not None:
                R = bounds['j_slack'].shape[2]
                smp = r['worst'] // (91 * R)
                dump[f'{tag}.j_ion'] = {'seed': seed, 'sample': smp, 'got': np.asarray(got['j_ion']).reshape(-1, 91, R)[smp],
                                        'want': np.asarray(want['j_ion']).reshape(-1, 91, R)[smp], **{q: v[smp] for q, v in inputs.items()}}
        d = pr.divergence_error(got['div_angle'], want['div_angle'], got.get('T_c'), want.get('T_c'), bounds, f'{tag} seed {seed}')
        note(f'{tag}.div_angle', d['err_div'], **d)
        note(f'{tag}.T_c', d['err_tc'], **d)

    import time
    t_start = time.perf_counter()
    seeds = [s_ for s_ in args.seed_list if not args.first_seed <= s_ < args.seeds] + list(range(args.first_seed, args.seeds))
    ran = []
    for it, seed in enumerate(seeds):
        if args.time_budget and time.perf_counter() - t_start > args.time_budget:
            print(f'time budget of {args.time_budget:.0f} s reached before seed {seed}: {len(seeds) - it} seeds not run', flush=True)
            break
        ran.append(seed)
        seed_now[0] = seed
        if it % 25 == 0:
            print(f'seed {seed} ({it} / {len(seeds)})', flush=True)      # a long campaign must not look hung
        rng = np.random.default_rng(1000 + seed)
        x = wild(rng, args.n)
        with np.errstate(all='ignore'):
            want = oc.coupled(x, k)
            pin = {q: x[q] for q in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')}
            terms = oc.plume_terms(*[pin[q] for q in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')], want['I_B0'], k)
            bounds = pr.plume_bounds(terms, want['I_B0'])
            lg = np.log(1.0 + x['P_b'] * k / (x['P_T'] * k))
            v_scale = np.abs(x['V_vac']) + np.abs(x['T_e'] * lg) + np.abs(x['T_e'] / ((x['P_T'] + x['Pstar']) * k) * (x['P_b'] * k))
        full = pem_v0_coupled(x)
        red = pem_v0_coupled(x, profile=False)
        for name, got in (('full', full), ('reduced', red)):
            assert np.array_equal(got['invalid'], want['invalid']), f'invalid flags differ ({name}, seed {seed})'
            for key in ('V_cc', 'I_B0', 'T'):
                g, w = np.asarray(got[key]).reshape(-1), np.asarray(want[key]).reshape(-1)
                same_pattern(g, w, f'{key} ({name}, seed {seed})')
                if key == 'V_cc':
                    fin = np.isfinite(w) & np.isfinite(v_scale)
                    note(f'{name}.{key}', np.max(np.abs(g[fin] - w[fin]) / (np.abs(w[fin]) + (pr.TAU / pr.TOL) * v_scale[fin] + 1e-300), initial=0.0))
                else:
                    note(f'{name}.{key}', rel_err(g, w))
            plume_check(name, got, want, bounds, seed, with_j=(name == 'full'), inputs={**x, 'I_B0': want['I_B0']})
        # the tile-interleaved input layout (pem_coupled_tiled_f64_dev) is the same evaluation bit for bit, wild inputs included
        tb = CoupledBatch(args.n, layout='tile')
        tb.set_inputs(x)
        tb.run()
        torch.cuda.synchronize()
        for key, val in tb.outputs().items():
            assert np.array_equal(val.cpu().numpy().reshape(-1), np.asarray(full[key]).reshape(-1), equal_nan=True), f'tile layout differs: {key} (seed {seed})'
        del tb
        b = CoupledBatch(args.n, mixed=True)
        b.set_inputs(x)
        b.run()
        torch.cuda.synchronize()
        j32 = b.j_ion.cpu().numpy().astype(np.float64).reshape(-1)
        j64 = np.asarray(full['j_ion']).reshape(-1).astype(np.float32).astype(np.float64)
        assert np.array_equal(j32, j64, equal_nan=True), f'mixed profile is not the rounded fp64 profile (seed {seed})'
        # the fused modes against their two-launch pipelines on the same wild inputs: likelihood and SVD compression
        ref = CoupledBatch(args.n, profile=True, thruster_qoi=False)
        ref.set_inputs(x)
        ref.run()
        fused = CoupledBatch(args.n, profile=False, thruster_qoi=False)
        fused.inputs.copy_(ref.inputs)
        want_ll = lik.per_sample(ref.j_ion).cpu().numpy()
        got_ll = fused.run_loglik(lik).cpu().numpy()
        assert torch.equal(fused.invalid, ref.invalid), f'invalid flags differ (fused likelihood, seed {seed})'
        same_pattern(got_ll, want_ll, f'fused loglik (seed {seed})')
        fin = np.isfinite(want_ll)
        note('fused.loglik', np.max(np.abs(got_ll[fin] - want_ll[fin]) / np.maximum(np.abs(want_ll[fin]), 1.0), initial=0.0))
        want_z = comp.compress(ref.j_ion).cpu().numpy()
        got_z = fused.run_latent(comp).cpu().numpy()
        assert torch.equal(fused.invalid, ref.invalid), f'invalid flags differ (fused compression, seed {seed})'
        same_pattern(got_z, want_z, f'fused latents (seed {seed})')
        with np.errstate(all='ignore'):
            scale = np.nansum(np.abs(np.log10(np.abs(ref.j_ion.cpu().numpy()))), axis=1, keepdims=True)   # sum_k |log10 j_k|
        fin = np.isfinite(want_z) & np.isfinite(scale)
        note('fused.latent', np.max((np.abs(got_z - want_z) / np.maximum(scale, 1.0))[fin], initial=0.0))
        # plume alone: one radius (fast path), two / three / five radii (few-radii recurrence kernel, even and odd R)
        p = dict(pin)
        p['I_B0'], p['T'] = full['I_B0'], full['T']
        for radii in ((1.0,), (0.7, 1.3), (0.5, 1.0, 2.5), (0.5, 0.8, 1.0, 1.7, 2.5)):     # R = 2: the 16-byte-store form of the few-radii kernel
            with np.errstate(all='ignore'):
                w = oc.plume(p['P_b'], p['c0'], p['c1'], p['c2'], p['c3'], p['c4'], p['c5'], p['sigma_cex'], p['I_B0'], k, T=p['T'], radii=radii)
                tR = oc.plume_terms(*[pin[q] for q in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')], p['I_B0'], k, radii=radii)
                bR = pr.plume_bounds(tR, p['I_B0'])
            g = current_density(p, sweep_radius=radii[0] if len(radii) == 1 else np.array(radii))
            plume_check(f'plume[R={len(radii)}]', g, w, bR, seed, inputs=p)
        # many radii, on the first samples of the seed (the blocks are large): 9 = the wave-per-sample kernel, 17 / 40 = the staged
        # kernel with two / one samples in flight (odd and even counts: runs that start on odd and even doubles)
        m = min(args.n, args.many_radii_samples)
        ps = {q: np.ascontiguousarray(np.asarray(v)[:m]) for q, v in p.items()}
        for R in (9, 17, 40):
            radii = tuple(np.linspace(0.45, 2.6, R))
            with np.errstate(all='ignore'):
                w = oc.plume(ps['P_b'], ps['c0'], ps['c1'], ps['c2'], ps['c3'], ps['c4'], ps['c5'], ps['sigma_cex'], ps['I_B0'], k, T=ps['T'], radii=radii)
                tR = oc.plume_terms(*[ps[q] for q in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')], ps['I_B0'], k, radii=radii)
                bR = pr.plume_bounds(tR, ps['I_B0'])
            g = current_density(ps, sweep_radius=np.array(radii))
            plume_check(f'plume[R={R}]', g, w, bR, seed, inputs=ps)
    in_range = [s_ for s_ in ran if args.first_seed <= s_ < args.seeds]
    print(f'{len(ran)} seeds x {args.n} wild samples (seeds {in_range[0] if in_range else "-"}..{in_range[-1] if in_range else "-"}'
          + (f' + {[s_ for s_ in ran if s_ not in in_range]}' if len(in_range) < len(ran) else '') + '):')
    print('NaN / inf / invalid patterns identical everywhere; worst errors (1e-10 = at the bound) and the cancellation they met:')
    for key, v in sorted(worst.items()):
        e = extra[key]
        tail = f'   cancelling entries {e["n_cancelling"]}, largest cond {e["cond"]:.1e}, tau seen {e["tau_seen"]:.1e}' if e['n_cancelling'] else ''
        print(f'  {key:22s} {v:.2e}   (seed {where[key]}){tail}')
    if args.dump:
        flat = {}
        for key, rec in dump.items():
            for q, v in rec.items():
                flat[f'{key}|{q}'] = np.asarray(v)
        np.savez(args.dump, **flat)
    # (the fused modes are compared with their two-launch pipelines, whose sums run in another order: 1e-9)
    for key, v in worst.items():
        tol = 1e-9 if key.startswith('fused.') else pr.TOL
        assert v <= tol, f'tolerance exceeded: {key} {v:.2e}'


if __name__ == '__main__':
    main()
