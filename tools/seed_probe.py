#!/usr/bin/env python3
"""One fuzz seed of tools/fuzz_parity.py looked at closely: the worst div_angle sample, its inputs, cos(got) - cos(want) and the bound's
terms (how seed 5160 -- an arccos pole, DESIGN section 2 -- was diagnosed).  python tools/seed_probe.py [seed]"""
from pathlib import Path
import sys
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests')); sys.path.insert(0, str(ROOT / 'tools'))
import fuzz_parity as fp
import parity_rules as pr
from oracle import oracle_ctypes as oc
from hallthrusterpem_amd.models import pem_v0_coupled
from hallthrusterpem_amd import constants
k = constants.TORR_2_PA
rng = np.random.default_rng(1000 + 5160)
x = fp.wild(rng, 20000)
with np.errstate(all='ignore'):
    want = oc.coupled(x, k)
    pin = [x[q] for q in ('P_b', 'c0', 'c1', 'c2', 'c3', 'c4', 'c5', 'sigma_cex')]
    terms = oc.plume_terms(*pin, want['I_B0'], k)
    bounds = pr.plume_bounds(terms, want['I_B0'])
full = pem_v0_coupled(x)
gd, wd = np.asarray(full['div_angle']).reshape(-1), np.asarray(want['div_angle']).reshape(-1)
with np.errstate(all='ignore'):
    cosw = bounds['cos_div'].reshape(-1); dcos = np.abs(cosw) * bounds['cos_rel'].reshape(-1)
    d = np.abs(gd - wd); rel = d / np.where(wd == 0, 1.0, np.abs(wd))
    inb = 2.0 * np.abs(np.sin(0.5 * (gd + wd)) * np.sin(0.5 * (gd - wd))) <= dcos
    fin = bounds['comparable'].reshape(-1) & np.isfinite(wd) & np.isfinite(gd)
    ex = np.where(fin & ~inb, rel, 0.0)
i = int(np.argmax(ex))
print('sample', i, 'excess', ex[i], 'got', repr(gd[i]), 'want', repr(wd[i]), 'cos_div', repr(cosw[i]), 'cos_rel', bounds['cos_rel'].reshape(-1)[i], 'cond', bounds['cond_cos'].reshape(-1)[i])
print({q: repr(float(np.asarray(v)[i])) for q, v in x.items()})
print('I_B0', repr(want['I_B0'][i]), 'T_c got/want', repr(np.asarray(full['T_c']).reshape(-1)[i]), repr(want['T_c'].reshape(-1)[i]))
print('cos(got) - cos(want)', np.cos(gd[i]) - np.cos(wd[i]), 'rel of cos', (np.cos(gd[i]) - np.cos(wd[i]))/np.cos(wd[i]))
