#!/usr/bin/env python3
"""bench.py -- coupled PEM-v0 (cathode -> thruster test double -> plume) evaluations per second on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic Monte-Carlo samples already resident in HBM:
a single `pem_coupled_f64_dev` launch over this rank's shard (BASELINE.json configs[2]: 1e7 coupled samples
sharded over 8 GPUs = 1.25e6 samples per GPU; the same per-GPU shard is used at every N, so scaling is weak),
followed at N > 1 by one RCCL all-gather of the reduced QoIs (V_cc, div_angle, T_c), the path's only exchange.

Prints ONE JSON line on rank 0 with the fields the driver expects plus
  "roofline":     the coupled kernel against the HBM roofline -- achieved = 872 algorithmic bytes per
                  evaluation x samples per launch / the kernel's mean duration, measured
                  here with HIP events on the stream the kernel runs on (torch's current stream);
  "cpu_baseline": the CPU oracle (a C restatement of the reference's NumPy path, OpenMP) timed on this host on
                  a bounded sample of the same workload -- reported for context, not a target.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
SAMPLES_PER_GPU = 1_250_000    # BASELINE.json configs[2]: 1e7 coupled samples / 8 GPUs


def synth_inputs(batch, seed, rank):
    """Fill the batch with draws from the PEM-v0 priors (pem_v0_SPT-100.yml) on device."""
    import torch
    g = torch.Generator(device=batch.device)
    g.manual_seed(seed * 1000 + rank)
    u = torch.rand((15, batch.n), dtype=torch.float64, device=batch.device, generator=g)
    x = batch.inputs
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
    del u


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
                      f'{best:.2f} s wall on {threads} OpenMP threads'}


def read_committed_traffic(n):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), if they were taken at this n."""
    for f in sorted((ROOT / 'profiles').glob('traffic_r*.json'), reverse=True):
        try:
            rec = json.loads(f.read_text())
            if int(rec.get('samples_per_launch', -1)) == int(n):
                return float(rec['hbm_bytes_per_launch'])
        except Exception:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--samples-per-gpu', type=int, default=SAMPLES_PER_GPU)
    ap.add_argument('--lanes', type=int, default=0, help='lanes per sample of the kernel (0 = library default)')
    ap.add_argument('--gather', choices=['qoi', 'full', 'none'], default='qoi',
                    help='what the N>1 all-gather moves: reduced QoIs (24 B/sample), full profiles, or nothing')
    ap.add_argument('--no-profile', action='store_true', help='reduced-QoI mode: never write j_ion (144 B/eval)')
    ap.add_argument('--mixed', action='store_true', help='fp64 arithmetic, fp32 storage of the profile (508 B/eval)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--seed', type=int, default=2)
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL; default) or gloo (rehearsal on one GPU)')
    args = ap.parse_args()

    # stdout carries exactly one JSON line: everything libraries print meanwhile (RCCL writes its version banner to
    # stdout under NCCL_DEBUG=VERSION) is sent to stderr by pointing fd 1 at fd 2 until the line is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from hallthrusterpem_amd import _lib
    from hallthrusterpem_amd.batch import CoupledBatch

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    local_dev = local_rank % max(1, torch.cuda.device_count())   # == local_rank on a real multi-GPU node
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
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_dev))
        else:
            dist.init_process_group(args.dist_backend)

    lib = _lib.load()
    _lib.require_device()
    lanes = lib.pem_set_lanes_per_sample(args.lanes)
    n = args.samples_per_gpu
    # outputs per evaluation exactly as SURVEY section 8d counts them: V_cc, j_ion[91], div_angle, T_c (+ invalid flag)
    batch = CoupledBatch(n, profile=not args.no_profile, mixed=args.mixed, thruster_qoi=False)
    synth_inputs(batch, args.seed, rank)
    # N > 1: the all-gather of step i runs beside the evaluation of step i+1 (RCCL works on its own stream):
    # two batches alternate so a QoI buffer is not rewritten while it is still being sent.
    nbuf = 2 if (multi and args.gather != 'none') else 1
    batches = [batch]
    for _ in range(nbuf - 1):
        b2 = CoupledBatch(n, profile=not args.no_profile, mixed=args.mixed, thruster_qoi=False)
        b2.inputs.copy_(batch.inputs)
        batches.append(b2)
    gathered, pending = [], [None] * nbuf
    for b in batches:
        if multi and args.gather == 'qoi':
            gathered.append(torch.empty((world * b.qoi.shape[0], n), dtype=torch.float64, device=b.device))
        elif multi and args.gather == 'full':
            gathered.append(torch.empty((world * n, b.j_ion.shape[1]), dtype=b.j_ion.dtype, device=b.device))
    counter = [0]

    def step():
        i = counter[0] % nbuf
        counter[0] += 1
        if pending[i] is not None:
            pending[i].wait()               # stream-level: this buffer's previous gather must have read it
            pending[i] = None
        batches[i].run()
        if gathered:
            src = batches[i].qoi if args.gather == 'qoi' else batches[i].j_ion
            pending[i] = dist.all_gather_into_tensor(gathered[i], src, async_op=True)

    def drain():
        for i, w in enumerate(pending):
            if w is not None:
                w.wait()
                pending[i] = None

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

    for _ in range(args.warmup):
        step()
    elapsed = timed(args.steps)
    # N > 1: also the gather-free rate of the same shards (reported beside `value`, SURVEY.md section 8e "report both")
    elapsed_nogather = None
    if gathered:
        saved, gathered = gathered, []
        elapsed_nogather = timed(args.steps)
        gathered = saved

    # kernel-only duration: HIP events on the launch stream around each launch (outside the timed region)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 50))]
    for a, b in evs:
        a.record()
        batch.run()
        b.record()
    torch.cuda.synchronize()
    kern_ms = sorted(a.elapsed_time(b) for a, b in evs)
    kern_mean_ms = sum(kern_ms) / len(kern_ms)
    frac_invalid = float(batch.invalid.float().mean().item())

    if rank == 0:
        bytes_per_launch = batch.bytes_per_eval * n
        achieved = bytes_per_launch / (kern_mean_ms * 1e-3) / 1e9
        traffic = read_committed_traffic(n) if not (args.no_profile or args.mixed) else None
        line = {
            'metric': 'coupled PEM-v0 model evals/sec', 'value': world * n * args.steps / elapsed, 'unit': 'evals/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64' if not args.mixed else 'f64 (profile stored as f32)', 'data': 'synthetic',
            'config': {'workload': 'coupled cathode->thruster(analytic test double)->plume forward UQ, '
                                   'BASELINE configs[2] shard (1e7 samples / 8 GPUs), 91 angles, R=1 at 1.0 m',
                       'samples_per_gpu': n, 'global_samples_per_step': world * n, 'seed': args.seed,
                       'TORR_2_PA': 133.322, 'lanes_per_sample': lanes, 'profile_written': not args.no_profile,
                       'gather': (args.gather + ', overlapped with the next step') if (multi and args.gather != 'none') else 'none',
                       'value_without_gather': (world * n * args.steps / elapsed_nogather) if elapsed_nogather else None,
                       'parallelism': f'sample-shard x{world}',
                       'invalid_fraction': frac_invalid},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'plume_r1_kernel<L,COUPLED,JMODE>', 'kernel_ms_mean': kern_mean_ms,
                         'kernel_ms_min': kern_ms[0], 'bytes_per_eval': batch.bytes_per_eval,
                         'bytes_per_launch': bytes_per_launch},
        }
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
