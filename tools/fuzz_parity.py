#!/usr/bin/env python3
"""Differential fuzzing of the device path against the CPU oracle (test infrastructure) on inputs far outside the
priors: signs, zeros, huge / tiny magnitudes, NaN / inf, beams around every table boundary.  For each seed the coupled
evaluation runs in full, reduced and mixed mode, fused with the likelihood and with the SVD compression (against their
two-launch pipelines), and the plume alone with several radii; NaN / inf / invalid patterns
must agree exactly, finite values to 1e-10 (div_angle by conftest.div_err's rule) wherever the quantity is defined by
normal-range arithmetic:
  * div_angle / T_c are ratios of two Simpson sums of the beam terms; when the beam amplitude I_B0 exp(-r n sigma) / r^2
    is itself a denormal number (0 < |.| < 1e-280) both sums are a few denormal bits in the reference and here, and
    their ratio is noise on both sides (0/0 = NaN or an arbitrary angle) -- div_angle / T_c of such samples are not
    compared, their j_ion is; when both beams are
    narrower than a quarter of the 1-degree grid only the centreline point contributes, cos_div = 1 to the last bit and
    arccos returns 0 or NaN depending on that bit -- div_angle of such samples is not compared; with c0 outside [0, 1]
    one beam amplitude is negative and the two sums cancel -- those samples are held to 1e-6;
  * V_cc = V_vac + T_e ln(1 + PB/PT) - T_e PB / (PT + P*) cancels for wild pressure ratios, j_cex carries the
    rounding of 1 - exp(-x), and with a negative amplitude or a negative j_cex (c0 outside [0, 1], negative density or
    cross-section) j_ion = j_beam + j_scat + j_cex cancels too: the tolerance is 1e-10 of the value plus 1e-13 of the
    largest term (for j_ion: of the largest entry of that sample's profile and of |j_cex|), not 1e-10 of the cancelled
    result.
Prints the worst errors.

    python tools/fuzz_parity.py [--seeds 40] [--n 20000]
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
    ap.add_argument('--seeds', type=int, default=40)
    ap.add_argument('--n', type=int, default=20_000)
    ap.add_argument('--priors', type=int, default=0, metavar='BATCHES',
                    help='instead: BATCHES x 1.25e6 samples drawn from the PEM-v0 priors, full and reduced mode, strict 1e-10')
    args = ap.parse_args()
    import torch
    from conftest import div_err, rel_err
    from oracle import oracle_ctypes as oc
    from hallthrusterpem_amd import constants
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.models import current_density, pem_v0_coupled
    oc.set_threads(16)
    worst = {}
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

    where, current_seed = {}, [0]

    def note(key, val):
        assert val == val, f'{key}: the error metric itself is NaN'
        if float(val) > worst.get(key, 0.0):
            worst[key], where[key] = float(val), current_seed[0]

    def same_pattern(a, b, what):
        assert np.array_equal(np.isnan(a), np.isnan(b)), f'NaN pattern differs: {what}'
        assert np.array_equal(np.isinf(a), np.isinf(b)) and np.array_equal(np.sign(a[np.isinf(a)]), np.sign(b[np.isinf(b)])), f'inf pattern differs: {what}'

    for seed in range(args.seeds):
        current_seed[0] = seed
        if seed % 25 == 0:
            print(f'seed {seed} / {args.seeds}', flush=True)      # a long campaign must not look hung
        rng = np.random.default_rng(1000 + seed)
        x = wild(rng, args.n)
        with np.errstate(all='ignore'):
            want = oc.coupled(x, constants.TORR_2_PA)
        full = pem_v0_coupled(x)
        red = pem_v0_coupled(x, profile=False)
        with np.errstate(all='ignore'):
            k = constants.TORR_2_PA
            base = want['I_B0'] * np.exp(-(x['c4'] * (x['P_b'] * k) + x['c5']) * x['sigma_cex'])      # radius 1 m
            beams_normal = ~((np.abs(base) < 1e-280) & (base != 0.0))
            same_sign = (x['c0'] >= 0.0) & (x['c0'] <= 1.0)
            a1w = np.minimum(x['c2'] * (x['P_b'] * k) + x['c3'], np.pi / 2)
            resolved = ~(np.maximum(np.abs(a1w), np.abs(a1w / x['c1'])) < 0.0044)
            beams_normal &= resolved
            lg = np.log(1.0 + x['P_b'] * k / (x['P_T'] * k))
            v_scale = np.abs(x['V_vac']) + np.abs(x['T_e'] * lg) + np.abs(x['T_e'] / ((x['P_T'] + x['Pstar']) * k) * (x['P_b'] * k))
            decay1 = np.exp(-(x['c4'] * (x['P_b'] * k) + x['c5']) * x['sigma_cex'])
            # rounding of 1 - decay, and of the sum when a large negative j_cex (negative density) cancels the beams
            j_floor = np.abs(want['I_B0']) / (2 * np.pi) * (8 * np.finfo(float).eps + 1e-13 * np.abs(1.0 - decay1))
        for name, got in (('full', full), ('reduced', red)):
            assert np.array_equal(got['invalid'], want['invalid']), f'invalid flags differ ({name}, seed {seed})'
            for key in ('V_cc', 'I_B0', 'T', 'T_c', 'div_angle') + (('j_ion',) if name == 'full' else ()):
                g, w = np.asarray(got[key]).reshape(-1), np.asarray(want[key]).reshape(-1)
                if key in ('div_angle', 'T_c'):         # beams_normal already excludes the unresolved beams
                    same_pattern(g[beams_normal], w[beams_normal], f'{key} ({name}, seed {seed})')
                else:
                    same_pattern(g, w, f'{key} ({name}, seed {seed})')
                if key in ('div_angle', 'T_c'):
                    for tag, m in (('', beams_normal & same_sign), (' (beams of opposite sign)', beams_normal & ~same_sign)):
                        note(f'{name}.{key}{tag}', div_err(g[m], w[m]) if key == 'div_angle' else rel_err(g[m], w[m]))
                elif key == 'V_cc':
                    fin = np.isfinite(w) & np.isfinite(v_scale)
                    note(f'{name}.{key}', np.max(np.abs(g[fin] - w[fin]) / np.maximum(np.maximum(np.abs(w[fin]), v_scale[fin]), 1e-300), initial=0.0))
                elif key == 'j_ion':
                    with np.errstate(all='ignore'):
                        peak = np.max(np.where(np.isfinite(w), np.abs(w), 0.0).reshape(-1, 91), axis=1)
                    fl = np.repeat(j_floor + 1e-13 * np.nan_to_num(peak), 91)
                    fin = np.isfinite(w) & np.isfinite(fl)
                    note(f'{name}.{key}', np.max(np.abs(g[fin] - w[fin]) / (np.abs(w[fin]) + 1e10 * fl[fin] + 1e-300), initial=0.0))
                else:
                    note(f'{name}.{key}', rel_err(g, w))
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
        # plume alone, several radii (generic kernel) and one radius (fast path)
        p = {k: x[k] for k in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')}
        p['I_B0'], p['T'] = full['I_B0'], full['T']
        for radii in ((1.0,), (0.5, 1.0, 2.5), (0.5, 0.8, 1.0, 1.7, 2.5)):   # fast path, lane-per-sample, wave-per-sample kernels
            with np.errstate(all='ignore'):
                w = oc.plume(p['P_b'], p['c0'], p['c1'], p['c2'], p['c3'], p['c4'], p['c5'], p['sigma_cex'], p['I_B0'],
                             constants.TORR_2_PA, T=p['T'], radii=radii)
            g = current_density(p, sweep_radius=radii[0] if len(radii) == 1 else np.array(radii))
            with np.errstate(all='ignore'):
                rr = np.asarray(radii)
                nsig = (x['c4'] * (x['P_b'] * constants.TORR_2_PA) + x['c5']) * x['sigma_cex']
                bR = p['I_B0'][:, None] * np.exp(-rr[None, :] * nsig[:, None]) / rr[None, :] ** 2
                okR = ~((np.abs(bR) < 1e-280) & (bR != 0.0)) & resolved[:, None]                     # (n, R)
                flR = np.abs(p['I_B0'])[:, None] / (2 * np.pi * rr[None, :] ** 2) * (8 * np.finfo(float).eps + 1e-13 * np.abs(1.0 - np.exp(-rr[None, :] * nsig[:, None])))
            for key in ('j_ion', 'div_angle', 'T_c'):
                gg, ww = np.asarray(g[key]).reshape(-1), np.asarray(w[key]).reshape(-1)
                if key in ('div_angle', 'T_c'):
                    rm = okR.reshape(-1)
                    same_pattern(gg[rm], ww[rm], f'plume {key} R={len(radii)} seed {seed}')
                else:
                    same_pattern(gg, ww, f'plume {key} R={len(radii)} seed {seed}')
                if key == 'j_ion':
                    with np.errstate(all='ignore'):
                        w3 = np.where(np.isfinite(ww), np.abs(ww), 0.0).reshape(args.n, 91, len(radii))
                        peak = np.max(w3, axis=1)                                  # (n, R)
                    fl = np.broadcast_to((flR + 1e-13 * peak)[:, None, :], (args.n, 91, len(radii))).reshape(-1)
                    fin = np.isfinite(ww) & np.isfinite(fl)
                    note(f'plume[R={len(radii)}].{key}', np.max(np.abs(gg[fin] - ww[fin]) / (np.abs(ww[fin]) + 1e10 * fl[fin] + 1e-300), initial=0.0))
                else:
                    for tag, ss in (('', same_sign), (' (beams of opposite sign)', ~same_sign)):
                        m = (okR & ss[:, None]).reshape(-1)
                        note(f'plume[R={len(radii)}].{key}{tag}', div_err(gg[m], ww[m]) if key == 'div_angle' else rel_err(gg[m], ww[m]))
    print(f'{args.seeds} seeds x {args.n} wild samples: NaN / inf / invalid patterns identical everywhere; worst errors:')
    for k, v in sorted(worst.items()):
        print(f'  {k:28s} {v:.2e}   (seed {where[k]})')
    # (the fused modes are compared with their two-launch pipelines, whose sums run in another order: 1e-9)
    for key, v in worst.items():
        tol = 1e-6 if 'opposite sign' in key else (1e-9 if key.startswith('fused.') else 1e-10)
        assert v <= tol, f'tolerance exceeded: {key} {v:.2e}'


if __name__ == '__main__':
    main()
