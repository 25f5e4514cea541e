#!/usr/bin/env python3
"""bench.py -- coupled PEM-v0 (cathode -> thruster test double -> plume) evaluations per second on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic Monte-Carlo samples already resident in HBM: this
rank's shard of BASELINE.json configs[2] (1e7 coupled samples sharded over 8 GPUs = 1.25e6 samples per GPU; the same
per-GPU shard at every N, so scaling is weak) evaluated by `pem_coupled_tiled_f64_dev` (inputs tile-interleaved; `--layout soa`:
`pem_coupled_f64_dev`, 15 arrays -- 2-4 % slower, profiles/grid_modes_r03.txt), followed at N > 1 by the path's only
exchange, the RCCL all-gather of the reduced QoIs (V_cc, div_angle, T_c).  Consecutive launches are dealt onto `--streams`
(default 2) HIP streams: every step is a complete launch over its own batch, but launch i+1 may start on the wave slots the
tail of launch i leaves idle (3-7 % per step); `config.single_stream` carries the one-stream rate of the same steps, and the
`roofline` object quotes the duration of ISOLATED launches, as before.  Steps rotate over `--batches` (default 8)
batches with their own inputs and result buffers -- 8.7 GB, the footprint of the whole config -- because re-evaluating ONE
batch every step is helped by the 256 MB Infinity Cache (its 150 MB of inputs never leave it: measured 191 against 209-225
us per launch, tools/mall_probe.py); the cache-assisted rate of a single re-evaluated batch, the number rounds 1 and early
round 2 reported, is carried as `config.single_batch_rerun` for comparison and is not `value`.  At N > 1 the shard is cut into `--chunks`
(default 2, equal, on tile boundaries) pieces and the all-gather of piece k runs beside the evaluation of piece k+1 (hallthrusterpem_amd.distributed.
ChunkedGather), so a single campaign overlaps its own exchange; `--gather once` is the one-collective-per-campaign
schedule, `--gather none` skips the exchange, `--gather full` moves the 91-point profiles.
Before the W untimed warm-up steps the same steps run, untimed, for `--spin-up-ms` (default 30) milliseconds: after an idle
period the GPU needs about 15 ms of load before a launch takes its steady duration (193 us instead of 210-220,
tools/warmup_probe.py, profiles/warmup_r03.txt), and a run of W + K = 25 launches would otherwise time that ramp
(`config.spin_up` says how many steps it took; `--spin-up-ms 0` turns it off).  The timed region is exactly K steps.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (python -m torch.distributed.run,
as a child process, before anything touches the GPU); under torch.distributed.run it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* as usual.  After the timed region every rank checks the gathered QoIs of every other rank bit for
bit against a local re-evaluation of that rank's seeded inputs and fails loudly on a mismatch.

Prints ONE JSON line on rank 0 with the fields the driver expects plus
  "roofline":     the coupled kernel against the HBM roofline -- achieved = 872 algorithmic bytes per evaluation
                   x samples per launch / the kernel's mean duration, measured here with HIP
                  events on the stream the kernel runs on (torch's current stream); `traffic` is NOT measured in this
                  run: it is the committed rocprofv3 --pmc result for the same launch size (`traffic_source` says so);
  "cpu_baseline": the CPU oracle (a C restatement of the reference's NumPy path, OpenMP) timed on this host on a
                  bounded sample of the same workload -- reported for context, not a target -- with the reference's own
                  NumPy rate from BASELINE.md (measured in the build container, not on this host) beside it;
  config.full_config (N = 1): the whole 1e7-sample configs[2] campaign as ONE launch on this GPU, measured in the
                  same run (8.7 GB of algorithmic traffic per pass).
  config.campaign (N = 1): the sampling loop around the hot path at configs[2] size -- sample + evaluate, NaN / IQR masks,
                  5 / 50 / 95 % bands -- as one fused call (total_ms; no_profile_total_ms) and as separate calls (outside the
                  timed region).
  --fp32 adds the config-5 report: the fp32-arithmetic reduced-QoI kernel against the fp64 one on identical inputs.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
SAMPLES_PER_GPU = 1_250_000    # BASELINE.json configs[2]: 1e7 coupled samples / 8 GPUs
FULL_CONFIG_SAMPLES = 10_000_000


def synth_inputs(batch, seed, rank, which=0):
    """Fill the batch with draws from the PEM-v0 priors (pem_v0_SPT-100.yml) on device; `which`
    numbers the batches a rank rotates over."""
    import torch
    g = torch.Generator(device=batch.device)
    g.manual_seed((seed * 1000 + rank) * 64 + which)
    chunk = 1 << 21                                 # bounded temporaries for the 1e7-sample campaign
    for lo in range(0, batch.n, chunk):
        hi = min(batch.n, lo + chunk)
        u = torch.rand((15, hi - lo), dtype=torch.float64, device=batch.device, generator=g)
        x = torch.empty_like(u)
        x[0] = 10 ** (u[0] * 4 - 8)                  # P_b      Torr, log-uniform over the domain (yml:9-17)
        x[1] = u[1] * 200 + 200                      # V_a      V
        x[2] = u[2] * 4 + 1                          # T_e      eV
        x[3] = u[3] * 60                             # V_vac    V
        x[4] = u[4] * 90e-6 + 10e-6                  # Pstar    Torr
        x[5] = u[5] * 90e-6 + 10e-6                  # P_T      Torr
        x[6] = u[6] * 5e-6 + 2e-6                    # mdot_a   kg/s
        x[7] = 10 ** (u[7] * 1.5 - 2.5)              # a_1      LogUniform(0.00316, 0.1)
        x[8] = u[8]                                  # c0
        x[9] = u[9] * 0.8 + 0.1                      # c1
        x[10] = u[10] * 30 - 15                      # c2
        x[11] = u[11] * (1.570796 - 0.2) + 0.2       # c3
        x[12] = 10 ** (u[12] * 4 + 18)               # c4
        x[13] = 10 ** (u[13] * 4 + 14)               # c5
        x[14] = u[14] * 7e-20 + 51e-20               # sigma_cex
        batch.load_soa(x, lo)                        # rows -> the batch's own input layout ('soa' or 'tile')
        del u, x


def reference_numpy_baseline():
    """The reference's own NumPy path as timed in the build container by tests/golden/time_reference.py (BASELINE.md section
    3.1) -- replayed from the committed record, because the reference cannot travel to the GPU box."""
    f = ROOT / 'profiles' / 'reference_numpy_baseline.json'
    try:
        rec = json.loads(f.read_text())
        return {'value': float(rec['value']), 'unit': rec['unit'], 'cores': int(rec['cores']),
                'source': f'replayed from profiles/{f.name} (written by {rec["generator"]} in the build container: {rec["what"]}; '
                          f'{rec["samples"]} samples, best {rec["best_s"]:.2f} s on {rec["host"]["cpu"]})'}
    except Exception as exc:
        return {'value': None, 'unit': 'evals/s', 'cores': 1, 'source': f'none: profiles/{f.name} is missing or unreadable ({exc})'}


def host_cpu_share():
    """CPUs this process may really use: affinity mask, cgroup quota, and the GPU box's 16-per-GPU share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = Path('/sys/fs/cgroup/cpu.max').read_text().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get('PEM_CPU_THREADS', '16'))))


def cpu_baseline(target_seconds=2.5):
    """Time the CPU oracle (OpenMP) on a bounded sample of the same coupled workload: best of 3 runs of about
    `target_seconds` wall each on `host_cpu_share()` threads (~20-30 s of CPU work in total)."""
    import numpy as np
    from oracle import oracle_ctypes as oc
    sys.path.insert(0, str(ROOT / 'tests'))
    from _inputs import coupled_inputs
    threads = oc.set_threads(host_cpu_share())
    probe = coupled_inputs(50_000, seed=9)
    oc.coupled(probe, 133.322)                       # warm-up: tables, thread pool, page faults
    t0 = time.perf_counter()
    oc.coupled(probe, 133.322)
    rate = 50_000 / (time.perf_counter() - t0)
    n = int(min(max(rate * target_seconds, 50_000), 8_000_000))
    x = coupled_inputs(n, seed=10)
    best = float('inf')
    for _ in range(3):
        t0 = time.perf_counter()
        out = oc.coupled(x, 133.322)
        best = min(best, time.perf_counter() - t0)
    assert np.isfinite(out['V_cc']).all()
    return {'value': n / best, 'unit': 'evals/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n} coupled samples (same priors, fp64, 91 angles, full profile), best of 3 runs: '
                      f'{best:.2f} s wall on {threads} OpenMP threads',
            'reference_numpy': reference_numpy_baseline()}


def kernel_source_hash():
    """Digest of the translation unit the coupled kernel is compiled from and of the headers it includes: what a committed
    counter measurement has to carry (`kernel_srchash`) to be replayed as this run's `roofline.traffic`."""
    import hashlib
    h = hashlib.sha256()
    csrc = ROOT / 'hallthrusterpem_amd' / 'csrc'
    for f in [csrc / 'pem_kernels.hip'] + sorted(csrc.glob('*.h')) + [ROOT / 'include' / 'pem_hip.h']:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def read_committed_traffic(n, layout='soa'):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic_r*.json) -- only a record taken at
    this launch size, for this input layout, FROM THE KERNEL SOURCES IN THE TREE (`kernel_srchash`): a kernel edit after the
    counters were collected drops the replay (traffic: null) instead of quoting the old ratio.
    Returns (bytes, file name, why-not)."""
    want = kernel_source_hash()
    stale = None
    for f in sorted((ROOT / 'profiles').glob('traffic_r*.json'), reverse=True):
        try:
            rec = json.loads(f.read_text())
            if int(rec.get('samples_per_launch', -1)) != int(n) or rec.get('layout', 'soa') != layout:
                continue
            if rec.get('kernel_srchash') != want:
                stale = stale or f'{f.name} was measured on other kernel sources (kernel_srchash {rec.get("kernel_srchash")}, tree {want})'
                continue
            return float(rec['hbm_bytes_per_launch']), f.name, None
        except Exception:
            continue
    return None, None, stale or 'no committed counter measurement of this launch size'


def read_committed_trace(n, layout='soa'):
    """Traced mean duration (us) of the coupled kernel from the newest committed rocprofv3 --kernel-trace summary of this launch
    size and layout that was taken ON THE KERNEL SOURCES IN THE TREE (profiles/traffic_r*.json: `kernel_mean_us`, `kernel_srchash`):
    (us, file name) or (None, why-not)."""
    want = kernel_source_hash()
    stale = None
    for f in sorted((ROOT / 'profiles').glob('traffic_r*.json'), reverse=True):
        try:
            rec = json.loads(f.read_text())
            if int(rec.get('samples_per_launch', -1)) != int(n) or rec.get('layout', 'soa') != layout:
                continue
            if rec.get('kernel_srchash') != want:
                stale = stale or f'{f.name} was traced on other kernel sources (kernel_srchash {rec.get("kernel_srchash")}, tree {want})'
                continue
            return float(rec['kernel_mean_us']), f.name
        except Exception:
            continue
    return None, stale or 'no committed trace of this launch size'


def event_times(fn, reps):
    """mean / min duration (ms) of `fn` between HIP events on torch's current stream"""
    import torch
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return sum(ms) / len(ms), ms[0]


def full_config_pass(n, seed, lanes, layout='soa'):
    """configs[2] whole: n = 1e7 coupled samples as one launch on this GPU (N = 1 only)."""
    import torch
    from hallthrusterpem_amd.batch import CoupledBatch
    b = CoupledBatch(n, profile=True, thruster_qoi=False, layout=layout)
    synth_inputs(b, seed, 0)
    for _ in range(2):
        b.run()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        b.run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    mean_ms, min_ms = event_times(b.run, reps)
    rec = {'samples': n, 'passes_timed': reps, 'ms_per_pass': 1e3 * wall, 'value': n / wall, 'unit': 'evals/s',
           'kernel_ms_mean': mean_ms, 'kernel_ms_min': min_ms, 'bytes_per_launch': b.bytes_per_eval * n,
           'achieved_GBs': b.bytes_per_eval * n / (mean_ms * 1e-3) / 1e9,
           'frac_of_peak': b.bytes_per_eval * n / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           'invalid_fraction': float(b.invalid.float().mean().item())}
    del b
    torch.cuda.empty_cache()
    return rec


def campaign_report(n, seed):
    """One forward-UQ campaign of `n` samples on this GPU (N = 1 only, outside the timed region): the sampling loop around the
    hot path -- sample + evaluate, the NaN / IQR masks of gen_data.py:125-174 and the 5 / 50 / 95 % bands of
    monte_carlo.py:363-658 -- as one fused driver call (drivers.forward_uq_statistics) with and without a stored profile, and as
    round 3's separate calls (drivers.forward_uq, filter_outputs, percentile_bands).  Best wall time of three calls after 30 ms
    of the same calls, results on the device.  Never lets the line down: an exception is reported in its place."""
    import torch
    try:
        from hallthrusterpem_amd import drivers

        def best(fn, reps=3, warm_ms=30.0):
            t_warm = time.perf_counter()                  # (allocating the stage's buffers idles the GPU: clocks ramp again, see --spin-up-ms)
            while True:
                r = fn()
                torch.cuda.synchronize()
                del r
                if time.perf_counter() - t_warm >= 1e-3 * warm_ms:
                    break
            t_best = 1e9
            for _ in range(reps):
                t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                t_best = min(t_best, time.perf_counter() - t0)
                del r
            return t_best

        # the campaign as ONE driver call (round 4): the profile's percentiles counted inside the evaluation launch, the outlier
        # counts with them (drivers.forward_uq_statistics / pem_coupled_mc_stats_f64_dev) -- with the profile kept, and without
        t_fused = best(lambda: drivers.forward_uq_statistics(n, seed=seed, keep_profile=True))
        r = drivers.forward_uq_statistics(n, seed=seed, keep_profile=True)
        flags = {'fused': bool(r['fused']), 'premasked': bool(r['premasked'])}
        del r
        torch.cuda.empty_cache()
        t_fused_np = best(lambda: drivers.forward_uq_statistics(n, seed=seed, keep_profile=False))
        torch.cuda.empty_cache()
        # the same campaign as three calls over a stored profile (round 3's form; five quantiles now share one selection)
        t_model = best(lambda: drivers.forward_uq(n, seed=seed, keep_profile=True))
        out = drivers.forward_uq(n, seed=seed, keep_profile=True)
        keep = {k: out[k] for k in ('V_cc', 'div_angle', 'T_c', 'j_ion')}
        t_masks = best(lambda: drivers.filter_outputs(keep))
        t_bands = best(lambda: drivers.percentile_bands(out))
        t_stats = best(lambda: drivers.campaign_statistics(keep))
        rec = {'samples': n, 'total_ms': 1e3 * t_fused, 'samples_per_s': n / t_fused,
               'no_profile_total_ms': 1e3 * t_fused_np, 'no_profile_samples_per_s': n / t_fused_np, **flags,
               'separate_calls': {'forward_uq_ms': 1e3 * t_model, 'filter_outputs_ms': 1e3 * t_masks, 'percentile_bands_ms': 1e3 * t_bands,
                                  'total_ms': 1e3 * (t_model + t_masks + t_bands),
                                  'campaign_statistics_ms': 1e3 * t_stats, 'total_one_selection_ms': 1e3 * (t_model + t_stats)},
               'note': 'one forward-UQ campaign: sample + evaluate, NaN / IQR masks and 5/50/95 % bands of every output, wall time, best of '
                       'three, every percentile equal to numpy bit for bit.  total_ms: drivers.forward_uq_statistics with the profile kept '
                       '(percentiles and outlier counts of the profile taken inside the evaluation launch); no_profile_total_ms: the same, '
                       'profile never written; separate_calls: forward_uq + filter_outputs + percentile_bands over the stored profile '
                       '(round 3), and with the five quantiles of a variable in one selection (campaign_statistics)'}
        del out, keep
        torch.cuda.empty_cache()
        return rec
    except Exception as exc:                              # (reported, not raised: the headline line must not depend on this leg)
        return {'samples': n, 'error': f'{type(exc).__name__}: {exc}'}


XGMI_LINK_GBS_PER_DIRECTION = 76.5   # MI355X_MICROARCH.md: 7 links x ~153 GB/s bidirectional per GPU


def gather_explanation(world, n, bytes_per_sample, ms_eval, value, value_nogather, rate_1gpu_equiv):
    """What DESIGN.md section 5 predicts for this run's exchange, beside what was measured: per step every rank receives
    (world - 1) shards of `bytes_per_sample` x n bytes, spread at best over its 7 inbound xGMI links."""
    recv = (world - 1) * n * bytes_per_sample
    links = min(7, max(1, world - 1))
    t_link_ms = recv / (links * XGMI_LINK_GBS_PER_DIRECTION * 1e9) * 1e3
    t_step_pred = max(ms_eval, t_link_ms)                          # gather fully overlapped with the evaluation of the next piece
    return {'bytes_received_per_rank_per_step': recv, 'bytes_sent_per_rank_per_step': n * bytes_per_sample,
            'inbound_links_assumed': links, 'link_GBs_per_direction_assumed': XGMI_LINK_GBS_PER_DIRECTION,
            'link_time_ms_predicted': t_link_ms, 'evaluation_ms_per_step': ms_eval,
            'predicted_ms_per_step': t_step_pred, 'predicted_value': world * n / (t_step_pred * 1e-3),
            'predicted_bound': 'xGMI links' if t_link_ms > ms_eval else 'evaluation kernel',
            'measured_value': value, 'measured_value_without_gather': value_nogather,
            'implied_inbound_GBs_per_rank': (recv / ((world * n / value)) / 1e9) if value else None,
            'note': 'prediction of DESIGN.md section 5: a step cannot be shorter than the evaluation of the shard, nor than its inbound '
                    'bytes over the links; near-linear scaling is expected of value_without_gather and of the reductions-only campaign, '
                    'not of a per-sample gather once (world - 1) x 30 MB per step exceed what the links carry in one evaluation'}


def reductions_campaign(n_total, seed, rank, world, group=None):
    """The campaign that exchanges SUMS, not samples (the mode that can scale linearly): every rank
    evaluates its shard of a forward-UQ campaign with the profile kept, the 5 / 50 / 95 % bands of every output are those
    of ALL ranks' samples (drivers.percentile_bands(sharded=True): histograms all-reduced, a few hundred candidates per rank
    gathered), and first-order / total Sobol' indices come from per-rank partial sums (one all-reduce of O(d n_qoi) doubles).
    No per-sample all-gather.  Wall time of the slowest rank, outside the timed region."""
    import torch
    import torch.distributed as dist
    from hallthrusterpem_amd import drivers
    try:
        def wall(fn):
            if world > 1:
                dist.barrier(group)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device='cuda')
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                dt = float(t.item())
            return dt, r
        wall(lambda: drivers.forward_uq(n_total, seed=seed, keep_profile=True, rank=rank, world=world, keep_inputs=False))   # warm
        t_eval, out = wall(lambda: drivers.forward_uq(n_total, seed=seed, keep_profile=True, rank=rank, world=world, keep_inputs=False))
        t_bands, bands = wall(lambda: drivers.percentile_bands(out, sharded=world > 1, group=group))
        n_base = max(1024, n_total // 14)
        t_sobol, sob = wall(lambda: drivers.sobol_indices(n_base, seed=seed, fixed={'P_b': 1e-5, 'V_a': 300.0, 'mdot_a': 5e-6}, group=group))
        total = t_eval + t_bands + t_sobol
        return {'samples': n_total, 'ranks': world, 'forward_uq_ms': 1e3 * t_eval, 'sharded_bands_ms': 1e3 * t_bands,
                'sobol_ms': 1e3 * t_sobol, 'sobol_evaluations': sob['evaluations'], 'total_ms': 1e3 * total,
                'samples_per_s': n_total / (t_eval + t_bands), 'sobol_evaluations_per_s': sob['evaluations'] / t_sobol,
                'exchange': 'all-reduced histograms and min / max, all-gathered candidate lists (about 1 MB per rank), one all-reduce of the '
                            'Sobol\' sums: no per-sample gather',
                'median_band_of_T_c': float(bands['T_c'][1])}
    except Exception as exc:
        return {'samples': n_total, 'error': f'{type(exc).__name__}: {exc}'}


def collective_library(backend):
    """'RCCL x.y.z' (torch's nccl backend on ROCm) or the backend's name"""
    try:
        import torch
        if backend == 'nccl':
            v = torch.cuda.nccl.version()
            return 'RCCL ' + '.'.join(str(x) for x in (v if isinstance(v, (tuple, list)) else (v,))) + f' (torch {torch.__version__}, backend nccl)'
        return f'{backend} (torch {torch.__version__})'
    except Exception as exc:
        return f'{backend} (version unavailable: {exc})'


def fp32_report(n, seed):
    """config 5's tolerance check: the fp32-arithmetic reduced-QoI kernel against the fp64 one on identical inputs."""
    from hallthrusterpem_amd.fp32 import compare_with_fp64
    return compare_with_fp64(n, lambda batch: synth_inputs(batch, seed, 0))


def spawn_ranks(args_list, n):
    """`bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child process (nothing in THIS process
    has touched the GPU) and pass its stdout -- the one JSON line of rank 0 -- and exit code through."""
    from hallthrusterpem_amd import build as hip_build
    if 'PEM_HIP_LIB' not in os.environ and hip_build.needs_build() and hip_build.have_hipcc():
        hip_build.build()                            # once, here: N ranks would otherwise each find the stale stamp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(Path(__file__).resolve()), *args_list]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--samples-per-gpu', type=int, default=SAMPLES_PER_GPU)
    ap.add_argument('--batches', type=int, default=8,
                    help='batches (own inputs and result buffers) the steps rotate over, so that no step finds its inputs in the '
                         '256 MB Infinity Cache; 1 = re-evaluate one batch every step (cache-assisted, as rounds 1-2a measured)')
    ap.add_argument('--lanes', type=int, default=0, help='lanes per sample of the kernel (0 = library default)')
    ap.add_argument('--gather', choices=['qoi', 'once', 'full', 'none'], default='qoi',
                    help='N>1 exchange: qoi = reduced QoIs (24 B/sample) in --chunks overlapped pieces; once = the same in one '
                         'collective per campaign; full = the 91-point profiles; none = no exchange')
    ap.add_argument('--chunk-align', choices=['round', 'tile'], default='tile',
                    help='tile: equal pieces on 64-sample boundaries (default: measured as fast as one launch, 312 512-sample pieces '
                         'run at 214-217 us per step against 213-218); round: every piece a whole number of rounds of the resident '
                         'grid (measured 4 %% slower on one stream, equal on two: profiles/schedule_r03.txt)')
    ap.add_argument('--chunks', type=int, default=2, help='pieces a shard is cut into at N>1 so that the all-gather of piece k '
                                                          'overlaps the evaluation of piece k+1')
    ap.add_argument('--layout', choices=['soa', 'tile'], default='tile',
                    help="input layout: soa = 15 arrays (pem_coupled_f64_dev); tile = [tiles][15][64] blocks (pem_coupled_tiled_f64_dev)")
    ap.add_argument('--streams', type=int, default=None, choices=[1, 2, 4],
                    help='consecutive launches (steps; at N > 1 the chunk launches of the pipeline) are dealt onto this many HIP '
                         'streams, so that launch i+1 fills the wave slots the tail of launch i leaves idle; 1 = every launch waits for '
                         'the one before (rounds 1-2).  Default: 2 when the fp64 profile is written (the benchmark), 1 with --no-profile / --mixed (the reduced-QoI '
                         'kernel is VALU-bound and two of its launches side by side take twice as long each: '
                         'profiles/launch_amortisation_r03.txt).  The roofline object always quotes the duration of ISOLATED '
                         'launches; the one-stream rate of the same run is carried as config.single_stream')
    ap.add_argument('--spin-up-ms', type=float, default=30.0,
                    help='untimed steps run BEFORE the W warm-up steps until this much time has passed: after an idle period the '
                         'GPU needs about 15 ms of load before a launch takes its steady 193 us instead of 210-220 '
                         '(profiles/warmup_r03.txt: the same ramp after 50 ms of idling, so a clock ramp, not first touch); a run '
                         'of W + K = 25 launches would otherwise time that ramp.  Reported as config.spin_up; 0 = off')
    ap.add_argument('--no-single-batch', action='store_true', help='skip the cache-assisted single-batch comparison run (profiling: every launch is then a rotating one)')
    ap.add_argument('--no-profile', action='store_true', help='reduced-QoI mode: never write j_ion (144 B/eval)')
    ap.add_argument('--mixed', action='store_true', help='fp64 arithmetic, fp32 storage of the profile (508 B/eval)')
    ap.add_argument('--fp32', action='store_true', help='add the fp32-arithmetic reduced-QoI kernel vs fp64 report (config 5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--full-config-samples', type=int, default=FULL_CONFIG_SAMPLES,
                    help='N=1: also run the whole configs[2] campaign as one launch of this many samples (0 = skip)')
    ap.add_argument('--campaign-samples', type=int, default=FULL_CONFIG_SAMPLES,
                    help='N=1: also time one forward-UQ campaign of this many samples stage by stage -- sample + evaluate, NaN / IQR '
                         'masks, 5/50/95 %% bands (config.campaign; 0 = skip)')
    ap.add_argument('--reductions-samples', type=int, default=-1,
                    help='N>1: also time a campaign that exchanges sums instead of samples -- sharded percentile bands + Sobol\' indices, '
                         'this many samples over all ranks (config.reductions_campaign; default: samples-per-gpu x N; 0 = skip)')
    ap.add_argument('--seed', type=int, default=2)
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL; default) or gloo (rehearsal on one GPU)')
    ap.add_argument('--oversubscribe', action='store_true',
                    help='rehearsal only: allow more ranks than GPUs (ranks then share devices; never a measurement)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(sys.argv[1:], args.gpus))

    # stdout carries exactly one JSON line: everything libraries print meanwhile (RCCL writes its version banner to
    # stdout under NCCL_DEBUG=VERSION) is sent to stderr by pointing fd 1 at fd 2 until the line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from hallthrusterpem_amd import _lib
    from hallthrusterpem_amd.batch import CoupledBatch
    from hallthrusterpem_amd.distributed import ChunkedGather, launch_rounds

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit('bench.py: no HIP device visible (there is no CPU path)')
    if world > ndev and not args.oversubscribe:
        raise SystemExit(f'bench.py: {world} ranks but {ndev} GPU(s) visible: one process per GPU (--oversubscribe is for rehearsals)')
    local_dev = local_rank % ndev          # == local_rank unless --oversubscribe
    torch.cuda.set_device(local_dev)
    # PEM_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, overlapped all-gather) with a single rank,
    # to rehearse the RCCL calls on a one-GPU box
    multi = world > 1 or os.environ.get('PEM_BENCH_FORCE_DIST') == '1'
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.dist_backend == 'nccl':
            # RCCL's kernels on a high-priority stream: the all-gather of piece k has to get onto the chip beside the
            # persistent evaluation kernel of piece k + 1 (which also leaves a few workgroup slots free: balanced rounds)
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_dev), pg_options=opts)
            except (AttributeError, TypeError):
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_dev))
        else:
            dist.init_process_group(args.dist_backend)

    lib = _lib.load()
    _lib.require_device()
    if args.mixed:
        args.layout = 'soa'                      # the fp32-profile entry point takes the 15 arrays only
    if args.streams is None:
        args.streams = 1 if (args.no_profile or args.mixed) else 2
    lanes = lib.pem_set_lanes_per_sample(args.lanes)
    n = args.samples_per_gpu
    # outputs per evaluation exactly as SURVEY section 8d counts them: V_cc, j_ion[91], div_angle, T_c (+ invalid flag)
    nb = max(1, args.batches)
    batches = []
    for k in range(nb):
        b = CoupledBatch(n, profile=not args.no_profile, mixed=args.mixed, thruster_qoi=False, layout=args.layout)
        synth_inputs(b, args.seed, rank, k)
        batches.append(b)
    batch = batches[0]
    counter = [0]

    def next_batch():
        b = batches[counter[0] % nb]
        counter[0] += 1
        return b

    # N > 1: the shard in `chunks` pieces, each piece's QoIs in its own send buffer, its all-gather beside the next piece
    gather_on = multi and args.gather != 'none'
    chunks = 1 if (not gather_on or args.gather in ('once', 'full')) else max(1, args.chunks)
    # ... cut on whole rounds of the persistent grid this launch takes (a round that is partly filled costs a whole one)
    mode = 0 if args.no_profile else (2 if args.mixed else 1)
    cus, wg_per_cu = _lib.coupled_occupancy(mode)
    round_samples, shard_rounds = launch_rounds(n, cus, wg_per_cu, memory_bound=mode != 0)
    if args.chunk_align == 'tile':
        round_samples = None                        # round 2's cut: equal pieces on 64-sample tile boundaries
    pipe = ChunkedGather(n, 3, chunks, batch.device, gather=gather_on and args.gather != 'full',
                         round_samples=round_samples) if multi else None
    full_recv = torch.empty((world * n, batch.j_ion.shape[1]), dtype=batch.j_ion.dtype, device=batch.device) \
        if (gather_on and args.gather == 'full' and batch.profile) else None
    pending_full = [None]
    use_gather = [True]

    # --streams S: launches are dealt onto S streams in turn.  A given (batch, chunk) always lands on the same stream (the
    # batches in rotation and hence the launches per cycle are a multiple of S), so stream order alone keeps a launch from
    # overwriting results an earlier one is still producing: no events between the streams, which would serialise them.
    if args.streams > 1 and nb % args.streams:
        raise SystemExit('bench.py: --batches must be a multiple of --streams (a batch is then always evaluated on the same stream)')
    side = [torch.cuda.Stream(device=batch.device) for _ in range(args.streams)] if args.streams > 1 else None
    launches = [0]

    def launch(cur, **kw):
        if side is None:
            cur.run(**kw)
            return
        st = side[launches[0] % len(side)]
        launches[0] += 1
        cur.run(stream=st, **kw)

    def step():
        cur = next_batch()
        if pipe is None:
            launch(cur)
            return
        if not use_gather[0]:
            for first, count in pipe.bounds:
                launch(cur, first=first, count=count)
            return
        if full_recv is not None:
            if pending_full[0] is not None:
                pending_full[0].wait()
            cur.run()
            pending_full[0] = dist.all_gather_into_tensor(full_recv, cur.j_ion, async_op=True)
        else:
            pipe.step(lambda first, count, out_rows: cur.run(first=first, count=count, qoi_out=out_rows), streams=side)

    def drain():
        if pipe is not None:
            pipe.drain()
        if pending_full[0] is not None:
            pending_full[0].wait()
            pending_full[0] = None

    def fence():
        drain()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nsteps):
        fence()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if multi:
            t = torch.tensor([dt], dtype=torch.float64, device=batch.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # The driver's arguments taken literally, first: W warm-up steps and K timed steps from a cold start (no spin-up: the process
    # has done nothing on the GPU but fill its inputs, and those fills ended a while ago) -- reported as config.value_no_spin_up
    # beside `value`, which is the same K steps after the clock spin-up below.
    elapsed_cold = None
    if args.spin_up_ms > 0:
        torch.cuda.synchronize()
        time.sleep(0.05)                                  # (an idle period, as before a process's first launch)
        for _ in range(args.warmup):
            step()
        elapsed_cold = timed(args.steps)
    # clock spin-up (see --spin-up-ms): plain steps, no fence between them and the warm-up steps that follow
    spin_steps = 0
    if args.spin_up_ms > 0:
        torch.cuda.synchronize()
        t_spin = time.perf_counter()
        while True:
            for _ in range(8):
                step()
            spin_steps += 8
            if pipe is None:
                # (the host runs ahead of the GPU: wait for the launches so far, minus a few that keep the queue fed)
                torch.cuda.current_stream().synchronize() if side is None else side[0].synchronize()
            done = time.perf_counter() - t_spin >= 1e-3 * args.spin_up_ms
            if multi:
                flag = torch.tensor([1 if done else 0], dtype=torch.int64, device=batch.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank takes the same number of steps
                done = bool(flag.item())
            if done or spin_steps >= 4096:
                break
    for _ in range(args.warmup):
        step()
    elapsed = timed(args.steps)
    # the same steps with every launch waiting for the one before (what rounds 1-2 timed), for the record
    elapsed_one_stream = None
    if side is not None and not multi:
        keep, side = side, None
        elapsed_one_stream = timed(args.steps)
        side = keep
    # N > 1: also the gather-free rate of the same shards (reported beside `value`, SURVEY.md section 8e "report both")
    elapsed_nogather = None
    if gather_on:
        use_gather[0] = False
        elapsed_nogather = timed(args.steps)
        use_gather[0] = True

    # N > 1: every rank's own gather-free rate (the MAX over ranks above is the slowest one's): min / max over the ranks
    per_rank_nogather = None
    if multi and elapsed_nogather is not None:
        fence()
        t0 = time.perf_counter()
        use_gather[0] = False
        for _ in range(args.steps):
            step()
        drain()
        torch.cuda.synchronize()
        mine = n * args.steps / (time.perf_counter() - t0)
        use_gather[0] = True
        allr = [torch.zeros(1, dtype=torch.float64, device=batch.device) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, torch.tensor([mine], dtype=torch.float64, device=batch.device))
        rates = [float(t.item()) for t in allr]
        per_rank_nogather = {'min': min(rates), 'max': max(rates), 'ranks': len(rates),
                             'note': 'evaluations/s of each rank alone over the same steps without the exchange (no barrier between ranks)'}

    # N > 1: what arrived is what was sent -- every rank re-evaluates every rank's seeded shard and compares bit for bit
    verified = None
    if gather_on and full_recv is None:
        which = counter[0] % nb                                 # the batch the next step takes
        step()
        torch.cuda.synchronize()                                # (the chunks may have run on side streams)
        got = pipe.assemble()                                   # [3][world * n], global order
        ok = True
        scratch = batches[which]
        for q in range(pipe.world):
            synth_inputs(scratch, args.seed, q, which)
            scratch.run()
            torch.cuda.synchronize()
            ok = ok and torch.equal(got[:, q * n:(q + 1) * n].view(torch.int64), scratch.qoi.view(torch.int64))
        synth_inputs(scratch, args.seed, rank, which)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=batch.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())
        if not verified:
            raise SystemExit(f'bench.py: rank {rank}: gathered QoIs differ from a local re-evaluation of the ranks\' shards')

    # kernel-only duration: HIP events on the launch stream around each launch (outside the timed region)
    kern_mean_ms, kern_min_ms = event_times(lambda: next_batch().run(), max(min(args.steps, 50), 2 * nb))
    frac_invalid = float(batch.invalid.float().mean().item())
    # the cache-assisted rate of ONE batch re-evaluated every step (what rounds 1 and 2a reported), for comparison only
    single = None
    if nb > 1 and not multi and not args.no_single_batch:
        for _ in range(5):
            batch.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            batch.run()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 100
        sm, _ = event_times(batch.run, 30)
        single = {'value': n / wall, 'ms_per_step': 1e3 * wall, 'kernel_ms_mean': sm,
                  'achieved_GBs': batch.bytes_per_eval * n / (sm * 1e-3) / 1e9,
                  'frac_of_peak': batch.bytes_per_eval * n / (sm * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  'note': 'one batch re-evaluated every step: its 150 MB of inputs stay in the 256 MB Infinity Cache; NOT the headline'}

    # what the memory system gives its BEST-CASE write stream of the shard's output size on this box (torch fill_ over the
    # batches' profile buffers in rotation -- they are no longer needed; an address-ordered one-shot fill, 6.4-6.9 TB/s, where a
    # store-only kernel with this kernel's 46-KB-per-wave pattern reaches 5.3-5.5: DESIGN.md section 6)
    write_stream = None
    if not multi and nb > 1 and not (args.no_profile or args.mixed) and not args.no_single_batch:
        k = [0]

        def fill():
            batches[k[0] % nb].j_ion.fill_(0.0)
            k[0] += 1
        fm, _ = event_times(fill, 40)
        write_stream = batches[0].j_ion.numel() * 8 / (fm * 1e-3) / 1e9

    red = None
    if multi and args.reductions_samples != 0 and not (args.no_profile or args.mixed):
        n_red = args.reductions_samples if args.reductions_samples > 0 else world * n
        red = reductions_campaign(n_red, args.seed, rank, world)           # (a collective: every rank takes part)

    if rank == 0:
        bytes_per_launch = batch.bytes_per_eval * n
        achieved = bytes_per_launch / (kern_mean_ms * 1e-3) / 1e9
        traffic, traffic_file, traffic_why = read_committed_traffic(n, args.layout) if not (args.no_profile or args.mixed) \
            else (None, None, 'counter passes are kept for the fp64-profile launch only')
        gather_desc = {'qoi': f'reduced QoIs (24 B/sample), {len(pipe.bounds) if pipe else 1} chunks per step, each all-gather overlapped with the next chunk\'s evaluation',
                       'once': 'reduced QoIs (24 B/sample), one all-gather per step, overlapped with the next step',
                       'full': '91-point profiles, one all-gather per step, overlapped with the next step'}.get(args.gather, 'none')
        traced_us, traced_src = read_committed_trace(n, args.layout) if not (args.no_profile or args.mixed) else (None, 'kept for the fp64-profile launch only')
        line = {
            'metric': 'coupled PEM-v0 model evals/sec', 'value': world * n * args.steps / elapsed, 'unit': 'evals/s',
            'value_methodology': (f'{args.steps} complete launches between two fences, dealt onto {args.streams} HIP stream(s)' +
                                  (f', after {args.warmup} warm-up steps and an untimed clock spin-up of {args.spin_up_ms:g} ms; the same steps '
                                   f'without the spin-up: config.value_no_spin_up; on one stream: config.single_stream' if args.spin_up_ms > 0 else
                                   f', after {args.warmup} warm-up steps, no clock spin-up') +
                                  '; rounds 1-2 reported the one-stream, no-spin-up figure'),
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64' if not args.mixed else 'f64 (profile stored as f32)', 'data': 'synthetic',
            'config': {'workload': 'coupled cathode->thruster(analytic test double)->plume forward UQ, '
                                   'BASELINE configs[2] shard (1e7 samples / 8 GPUs), 91 angles, R=1 at 1.0 m',
                       'samples_per_gpu': n, 'global_samples_per_step': world * n, 'seed': args.seed,
                       'TORR_2_PA': 133.322, 'lanes_per_sample': lanes, 'profile_written': not args.no_profile,
                       'batches_rotated': nb, 'input_layout': args.layout, 'streams': args.streams,
                       'spin_up': {'steps': spin_steps, 'ms': args.spin_up_ms,
                                   'note': 'untimed steps before the W warm-up steps: the GPU reaches its steady clocks after about 15 ms '
                                           'of load (profiles/warmup_r03.txt); --spin-up-ms 0 times the ramp instead'},
                       'value_no_spin_up': ({'value': world * n * args.steps / elapsed_cold, 'ms_per_step': 1e3 * elapsed_cold / args.steps,
                                             'note': f'the same {args.warmup} warm-up + {args.steps} timed steps from a cold start, before the spin-up: what the '
                                                     f'arguments mean taken literally (a run of a few dozen launches times the clock ramp, profiles/warmup_r03.txt)'}
                                            if elapsed_cold else None),
                       'single_stream': ({'ms_per_step': 1e3 * elapsed_one_stream / args.steps, 'value': world * n * args.steps / elapsed_one_stream,
                                          'note': 'the same steps on ONE stream, every launch waiting for the one before'}
                                         if elapsed_one_stream else None),
                       'single_batch_rerun': single,
                       'launch_rounds': {'samples_per_round': round_samples or launch_rounds(n, cus, wg_per_cu, mode != 0)[0],
                                         'rounds_per_shard': shard_rounds, 'compute_units': cus, 'workgroups_per_cu': wg_per_cu,
                                         'chunks': [list(b) for b in pipe.bounds] if pipe else [[0, n]],
                                         'chunk_align': args.chunk_align},
                       'gather': gather_desc if gather_on else 'none',
                       'gathered_qoi_verified': verified,
                       'value_without_gather': (world * n * args.steps / elapsed_nogather) if elapsed_nogather else None,
                       'value_without_gather_per_rank': per_rank_nogather,
                       'ranks_seen': (dist.get_world_size() if multi else 1),
                       'collective_library': collective_library(args.dist_backend) if multi else None,
                       'gather_explained': (gather_explanation(world, n, 24 if args.gather != 'full' else 8 * 91,
                                                               1e3 * (elapsed_nogather or elapsed) / args.steps, world * n * args.steps / elapsed,
                                                               (world * n * args.steps / elapsed_nogather) if elapsed_nogather else None, None)
                                            if gather_on else None),
                       'parallelism': f'sample-shard x{world}',
                       'invalid_fraction': frac_invalid},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         'frac_traced': (bytes_per_launch / (traced_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traced_us else None,
                         'frac_traced_source': (f'rocprofv3 --kernel-trace mean of this kernel, {traced_us:.1f} us per launch, from profiles/{traced_src} '
                                                f'(same kernel sources: kernel_srchash {kernel_source_hash()}); not measured in this run')
                                               if traced_us else f'none: {traced_src}',
                         'traffic': traffic,
                         'traffic_source': (f'replayed from profiles/{traffic_file} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this '
                                            f'launch size, gfx950 corrections applied; same kernel sources, kernel_srchash {kernel_source_hash()}); '
                                            f'not measured in this run') if traffic else f'none: {traffic_why}',
                         'kernel': 'plume_r1_kernel<L,COUPLED,JMODE>', 'kernel_ms_mean': kern_mean_ms,
                         'kernel_ms_note': ('ISOLATED launches, one at a time on one stream, HIP events around each (what rocprofv3 --kernel-trace '
                                            'reports for `bench.py --streams 1`: profiles/r03s1_summary.md).  With --streams 2 consecutive launches '
                                            'overlap at their ends: ms_per_step is then SMALLER than this, and per-dispatch durations in a trace of '
                                            'that command are longer than a step (profiles/r03s2_raw/)') if side is not None else
                                           'launches one at a time on one stream, HIP events around each',
                         'kernel_ms_min': kern_min_ms, 'bytes_per_eval': batch.bytes_per_eval,
                         'bytes_per_launch': bytes_per_launch,
                         'steps_overlapped': ({'streams': args.streams, 'ms_per_step': 1e3 * elapsed / args.steps,
                                               'achieved': bytes_per_launch * args.steps / elapsed / 1e9,
                                               'frac': bytes_per_launch * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                                               'note': 'algorithmic bytes per step / wall time per step of the timed region: launches on '
                                                       'alternating streams overlap at their ends, so a step costs less than an isolated launch'}
                                              if (side is not None and not multi) else None),
                         'write_stream_GBs': write_stream,
                         'frac_of_write_stream': (achieved / write_stream) if write_stream else None},
        }
        if world == 1 and not multi and args.full_config_samples > 0 and not (args.no_profile or args.mixed):
            del batch, batches, b
            torch.cuda.empty_cache()
            line['config']['full_config'] = full_config_pass(args.full_config_samples, args.seed, lanes, args.layout)
        if world == 1 and not multi and args.campaign_samples > 0 and not (args.no_profile or args.mixed):
            line['config']['campaign'] = campaign_report(args.campaign_samples, args.seed)
        if red is not None:
            line['config']['reductions_campaign'] = red
        if args.fp32 and world == 1:
            line['fp32'] = fp32_report(n, args.seed)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        else:
            line['cpu_baseline'] = None
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
