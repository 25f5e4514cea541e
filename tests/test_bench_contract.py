"""bench.py's contract with the driver: one JSON line on stdout with the agreed keys, at N = 1 and -- launched exactly
as the driver launches it, through torch.distributed.run -- at N = 2 (gloo collectives here, because both ranks share
this box's one GPU and RCCL refuses two ranks on one device; the code path is otherwise the RCCL one)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline', 'cpu_baseline'}
SMALL = ['--steps', '4', '--warmup', '2', '--samples-per-gpu', '131072']


def _one_json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout                      # nothing but the JSON line goes to stdout
    return json.loads(lines[0])


def test_single_gpu_line():
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *SMALL, '--no-cpu-baseline'], capture_output=True, text=True,
                         cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _one_json_line(out.stdout)
    assert KEYS <= set(line) and line['n_gpus'] == 1 and line['steps'] == 4 and line['warmup'] == 2
    assert line['metric'] == 'coupled PEM-v0 model evals/sec' and line['unit'] == 'evals/s' and line['vs_baseline'] is None
    assert line['scaling'] == 'weak' and line['higher_is_better'] is True and line['dtype'] == 'f64'
    r = line['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert r['bytes_per_eval'] == 872 and line['config']['samples_per_gpu'] == 131072
    assert abs(line['value'] - 131072 * 4 / (line['ms_per_step'] * 4e-3)) / line['value'] < 1e-9


def test_two_ranks_through_torch_distributed_run():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(ROOT / 'bench.py'), '--gpus', '2', *SMALL, '--dist-backend', 'gloo']
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _one_json_line(out.stdout)
    assert line['n_gpus'] == 2 and line['config']['global_samples_per_step'] == 2 * 131072
    assert line['config']['gather'].startswith('qoi') and line['config']['value_without_gather'] >= line['value'] * 0.5
    assert line['cpu_baseline'] is None and line['config']['parallelism'] == 'sample-shard x2'


def test_rccl_code_path_with_one_rank():
    """PEM_BENCH_FORCE_DIST=1 takes the N > 1 branch (process group on the `nccl` = RCCL backend, overlapped
    all_gather_into_tensor, barrier, max-over-ranks all_reduce) with a single rank -- the RCCL calls themselves, which two
    ranks on one GPU cannot exercise."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PEM_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
               LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *SMALL, '--no-cpu-baseline'], capture_output=True, text=True,
                         cwd=ROOT, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _one_json_line(out.stdout)
    assert line['n_gpus'] == 1 and line['config']['gather'].startswith('qoi') and line['config']['value_without_gather'] > 0
