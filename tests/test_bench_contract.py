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
SMALL = ['--steps', '4', '--warmup', '2', '--samples-per-gpu', '131072', '--full-config-samples', '0', '--campaign-samples', '0']


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
    assert 'full_config' not in line['config'] and line['config']['gather'] == 'none'
    # the untimed clock spin-up in front of the warm-up steps is disclosed, and can be turned off
    assert line['config']['spin_up']['steps'] >= 8 and line['config']['spin_up']['ms'] == 30.0
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *SMALL, '--no-cpu-baseline', '--spin-up-ms', '0'], capture_output=True,
                         text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert _one_json_line(out.stdout)['config']['spin_up']['steps'] == 0
    # steps rotate over 8 batches (nothing is re-read from the Infinity Cache); the one-batch rate is reported beside, not as value
    sb = line['config']['single_batch_rerun']
    assert line['config']['batches_rotated'] == 8 and sb['value'] > 0 and 'NOT the headline' in sb['note']
    if r['traffic'] is not None:
        assert 'not measured in this run' in r['traffic_source']
    # launches are dealt onto two streams by default; the one-stream rate of the same steps and the geometry of the launch
    # are carried beside, and the roofline object keeps the duration of isolated launches
    c = line['config']
    assert c['streams'] == 2 and c['single_stream']['value'] > 0 and c['input_layout'] == 'tile'
    assert r['steps_overlapped']['streams'] == 2 and abs(r['steps_overlapped']['ms_per_step'] - line['ms_per_step']) < 1e-12
    assert c['launch_rounds']['chunks'] == [[0, 131072]] and c['launch_rounds']['samples_per_round'] > 0


def test_single_stream_and_soa_inputs_are_options():
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *SMALL, '--no-cpu-baseline', '--streams', '1', '--layout', 'soa'],
                         capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _one_json_line(out.stdout)
    assert line['config']['streams'] == 1 and line['config']['single_stream'] is None and line['roofline']['steps_overlapped'] is None
    assert line['config']['input_layout'] == 'soa' and line['value'] > 0


def test_single_gpu_line_carries_the_whole_config():
    """configs[2] is 1e7 coupled samples: at N = 1 the line also carries that campaign as one launch, measured in the same
    run (here at a reduced size so that the test stays short; the driver's run uses the default 1e7)."""
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '4', '--warmup', '2', '--full-config-samples', '2000000',
                          '--campaign-samples', '300000'], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _one_json_line(out.stdout)
    camp = line['config']['campaign']                       # the sampling loop around the hot path, stage by stage
    assert 'error' not in camp and camp['samples'] == 300_000 and camp['samples_per_s'] > 1e6
    sep = camp['separate_calls']                            # round 3's three calls over a stored profile, and with one selection per variable
    assert abs(sep['total_ms'] - (sep['forward_uq_ms'] + sep['filter_outputs_ms'] + sep['percentile_bands_ms'])) < 1e-9
    assert sep['campaign_statistics_ms'] > 0 and sep['total_one_selection_ms'] > 0
    # round 4: the campaign as ONE call, percentiles and outlier counts taken inside the evaluation launch; with and without a profile
    assert camp['total_ms'] > 0 and camp['no_profile_total_ms'] > 0 and camp['fused'] is True and camp['premasked'] is True
    fc = line['config']['full_config']
    assert fc['samples'] == 2_000_000 and fc['bytes_per_launch'] == 872 * 2_000_000 and fc['value'] > 1e9
    assert abs(fc['frac_of_peak'] - fc['achieved_GBs'] / 8000.0) < 1e-12 and 0.3 < fc['frac_of_peak'] < 1.0
    # roofline.traffic is replayed from a committed counter measurement only while that measurement was taken on the kernel
    # sources in the tree (kernel_srchash); after a kernel edit it is dropped, not quoted
    src = line['roofline']['traffic_source']
    assert line['config']['samples_per_gpu'] == 1_250_000
    if line['roofline']['traffic'] is None:
        assert src.startswith('none: ')
    else:
        assert src.startswith('replayed from profiles/') and 'kernel_srchash' in src
        assert 0.9 * 872 * 1_250_000 < line['roofline']['traffic'] < 1.3 * 872 * 1_250_000
    cb = line['cpu_baseline']
    # the reference's own NumPy rate is replayed from the committed record of tests/golden/time_reference.py, not a literal
    import json as _json
    rec = _json.loads((ROOT / 'profiles' / 'reference_numpy_baseline.json').read_text())
    assert cb['kind'] == 'port' and cb['reference_numpy']['value'] == rec['value'] and cb['reference_numpy']['cores'] == 1
    assert 'replayed from profiles/reference_numpy_baseline.json' in cb['reference_numpy']['source'] and rec['generator'] == 'tests/golden/time_reference.py'
    # the line is self-sufficient under the driver's literal arguments (round 4): the cold figure, the methodology, the traced fraction
    cold = line['config']['value_no_spin_up']
    assert cold['value'] > 0 and cold['ms_per_step'] > 0 and 'spin-up' in line['value_methodology']
    assert 'frac_traced' in line['roofline'] and line['roofline']['frac_traced_source']
    if line['roofline']['frac_traced'] is not None:
        assert 0.3 < line['roofline']['frac_traced'] < 1.0 and 'kernel_srchash' in line['roofline']['frac_traced_source']


def test_two_ranks_through_torch_distributed_run():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(ROOT / 'bench.py'), '--gpus', '2', *SMALL, '--dist-backend', 'gloo', '--oversubscribe']
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _one_json_line(out.stdout)
    assert line['n_gpus'] == 2 and line['config']['global_samples_per_step'] == 2 * 131072
    assert line['config']['gather'].startswith('reduced QoIs (24 B/sample), 2 chunks') and line['config']['value_without_gather'] >= line['value'] * 0.5
    assert line['cpu_baseline'] is None and line['config']['parallelism'] == 'sample-shard x2'
    # after the overlapped loop every rank compared what it received from every rank with a local re-evaluation
    assert line['config']['gathered_qoi_verified'] is True
    # the line explains an N > 1 run by itself (round 4): who took part, what was exchanged, what the links allow, each rank's own rate
    cfg = line['config']
    assert cfg['ranks_seen'] == 2 and 'gloo' in cfg['collective_library']
    pr = cfg['value_without_gather_per_rank']
    assert pr['ranks'] == 2 and 0 < pr['min'] <= pr['max']
    ge = cfg['gather_explained']
    assert ge['bytes_sent_per_rank_per_step'] == 24 * 131072 and ge['bytes_received_per_rank_per_step'] == 24 * 131072
    assert ge['predicted_bound'] in ('xGMI links', 'evaluation kernel') and ge['predicted_value'] > 0 and ge['measured_value'] == line['value']
    assert abs(ge['predicted_ms_per_step'] - max(ge['link_time_ms_predicted'], ge['evaluation_ms_per_step'])) < 1e-12
    # ... and carries the campaign that exchanges sums, not samples: sharded bands + Sobol' indices, no per-sample gather
    red = cfg['reductions_campaign']
    assert 'error' not in red, red
    assert red['ranks'] == 2 and red['samples'] == 2 * 131072 and red['samples_per_s'] > 0 and red['sobol_evaluations'] >= 14 * 1024
    assert 'no per-sample gather' in red['exchange'] and 0.0 < red['median_band_of_T_c'] < 1.0


@pytest.mark.parametrize('gather', ['once', 'qoi'])
def test_gpus_flag_alone_starts_the_ranks(gather):
    """`python bench.py --gpus 2` the way the driver calls `--gpus 1`: no torch.distributed.run around it -- bench.py
    starts the ranks itself, as a child process, before it touches the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', *SMALL, '--dist-backend', 'gloo', '--oversubscribe',
                          '--gather', gather, '--chunks', '3'], capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _one_json_line(out.stdout)
    assert line['n_gpus'] == 2 and line['config']['gathered_qoi_verified'] is True
    assert ('3 chunks' in line['config']['gather']) == (gather == 'qoi')


def test_more_ranks_than_gpus_is_refused():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', *SMALL, '--dist-backend', 'gloo'],
                         capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert out.returncode != 0 and 'one process per GPU' in out.stderr and not out.stdout.strip()


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
    assert line['n_gpus'] == 1 and line['config']['gather'].startswith('reduced QoIs') and line['config']['value_without_gather'] > 0
    assert line['config']['gathered_qoi_verified'] is True
