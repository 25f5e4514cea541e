"""The N > 1 path on CPU: world_size-2 (and 3, ragged shards) process groups over gloo.

The evaluator injected here is the CPU oracle (tests may use it; the product's distributed module takes the
evaluator as an argument and knows nothing about it).  Checked: shards tile the batch, the single all-gather
restores global order, and the gathered QoIs equal a one-process evaluation bit for bit (samples are independent)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'tests'))

from hallthrusterpem_amd.distributed import (ChunkedGather, all_gather_rows, chunk_bounds, evaluate_sharded, launch_rounds,  # noqa: E402
                                             max_shard, shard_bounds)


def test_shard_bounds_tile_the_batch():
    for n in (0, 1, 7, 64, 1000, 1_250_000 * 8 + 3):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(n, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == max_shard(n, world)
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from _inputs import coupled_inputs
        from oracle import oracle_ctypes as oc
        oc.set_threads(1)
        x_all = coupled_inputs(n_total, seed=77)          # stands in for a counter-based sampler keyed by global index

        def make_inputs(lo, hi):
            return {k: v[lo:hi] for k, v in x_all.items()}

        def evaluate(x):
            return {k: torch.from_numpy(np.asarray(v)) for k, v in oc.coupled(x, 133.322).items()}

        gathered, local = evaluate_sharded(n_total, make_inputs, evaluate)
        lo, hi = shard_bounds(n_total, world, rank)
        assert local['V_cc'].numel() == hi - lo
        # a second collective with a different row count, straight through all_gather_rows
        rows = torch.arange(lo, hi, dtype=torch.float64).repeat(2, 1)
        idx = all_gather_rows(rows, n_total)
        assert torch.equal(idx[0], torch.arange(n_total, dtype=torch.float64))
        np.savez(Path(out_dir) / f'rank{rank}.npz', **{k: v.numpy() for k, v in gathered.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n_total', [(2, 1000), (3, 1001)])
def test_sharded_forward_uq_matches_single_process(tmp_path, world, n_total):
    from _inputs import coupled_inputs
    from oracle import oracle_ctypes as oc
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    want = oc.coupled(coupled_inputs(n_total, seed=77), 133.322)
    for r in range(world):
        got = np.load(tmp_path / f'rank{r}.npz')
        for k in ('V_cc', 'div_angle', 'T_c'):
            assert np.array_equal(got[k], want[k], equal_nan=True), (r, k)


def test_chunk_bounds_tile_a_shard_on_tile_boundaries():
    for n in (1, 63, 64, 65, 1000, 131072, 1_250_000):
        for k in (1, 3, 4, 8, 10 ** 6):
            b = chunk_bounds(n, k)
            assert b[0][0] == 0 and sum(c for _, c in b) == n and all(first % 64 == 0 and c > 0 for first, c in b)
            assert all(x[0] + x[1] == y[0] for x, y in zip(b, b[1:])) and len(b) <= max(1, k)
            assert max(c for _, c in b) - min(c for _, c in b[:-1] or b) <= 64
    assert chunk_bounds(0, 4) == []


@pytest.mark.parametrize('cus,wg_per_cu', [(256, 2), (256, 1), (304, 2), (64, 3)])
def test_every_chunk_of_the_pipeline_is_a_whole_number_of_rounds(cus, wg_per_cu):
    """The N > 1 schedule cuts a shard into range launches.  A persistent launch costs whole rounds of
    its grid -- a partly filled round takes as long as a full one -- so every piece but the last must be a whole number of
    the rounds the LIBRARY's grid arithmetic gives that very piece (pem_persistent_grid: no GPU needed), the last piece
    carries the shard's own tail and nothing else, and the step as a whole needs exactly the rounds of the single launch
    (round 2 cut on 64-sample boundaries: 4 x 3 rounds against 10 for the BASELINE shard)."""
    from hallthrusterpem_amd import _lib
    for n in (1_250_000, 1_250_001, 10_000_000, 131_072, 700_000, 65, 64 * 4 * cus * wg_per_cu * 3):
        per_round, rounds = launch_rounds(n, cus, wg_per_cu)
        assert per_round % 256 == 0 and (rounds - 1) * per_round < n <= rounds * per_round
        for k in (1, 2, 3, 4, 8, 64):
            b = chunk_bounds(n, k, round_samples=per_round)
            assert b[0][0] == 0 and sum(c for _, c in b) == n and all(x[0] + x[1] == y[0] for x, y in zip(b, b[1:]))
            assert len(b) == min(k, rounds) and all(first % 64 == 0 and c > 0 for first, c in b)
            total_rounds = 0
            for i, (first, count) in enumerate(b):
                grid, spr = _lib.persistent_grid(count, cus, wg_per_cu)      # what the launch of this piece will take
                r = -(-count // spr)
                total_rounds += r
                if i < len(b) - 1:
                    assert spr == per_round and count == r * per_round, (n, k, i, count, spr, per_round)
                else:                                                          # the tail piece carries the shard's ragged end
                    assert (r - 1) * spr < count <= r * spr
            assert total_rounds == rounds, (n, k, b)
            sizes = [-(-c // per_round) for _, c in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    # round 2's cut of the BASELINE shard, for the record: three rounds per piece, the last one 38 % full
    old = chunk_bounds(1_250_000, 4)
    assert sum(-(-c // 131_072) for _, c in old) == 12 and launch_rounds(1_250_000, 256, 2)[1] == 10


def _pipeline_worker(rank, world, port, n_local, chunks, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pipe = ChunkedGather(n_local, 3, chunks, 'cpu')
        state = {'step': 0}

        def evaluate(first, count, out_rows):        # a stand-in model: row i of global sample g at step s
            g = torch.arange(rank * n_local + first, rank * n_local + first + count, dtype=torch.float64)
            for i in range(3):
                out_rows[i, :count] = g * (i + 1) + 1000.0 * state['step']
        for s in range(3):                           # repeated campaigns reuse the chunk buffers: the waits must hold
            state['step'] = s
            pipe.step(evaluate)
        got = pipe.assemble()
        g = torch.arange(world * n_local, dtype=torch.float64)
        want = torch.stack([g * (i + 1) + 2000.0 for i in range(3)])
        assert torch.equal(got, want), (rank, chunks)
        assert len(pipe.bounds) == min(chunks, (n_local + 63) // 64)
        Path(out_dir, f'ok{rank}').write_text('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,n_local,chunks', [(2, 1000, 1), (2, 1000, 4), (3, 777, 3), (2, 50, 8)])
def test_chunked_gather_pipeline_restores_global_order(tmp_path, world, n_local, chunks):
    """hallthrusterpem_amd.distributed.ChunkedGather: K chunk all-gathers issued while the next chunk is evaluated, over
    repeated steps; every rank ends up with every rank's rows of the LAST step in global order."""
    mp.spawn(_pipeline_worker, args=(world, _free_port(), n_local, chunks, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f'ok{r}').exists() for r in range(world))


def test_chunked_gather_without_a_process_group_is_the_local_result():
    pipe = ChunkedGather(300, 2, 3, 'cpu')

    def evaluate(first, count, out_rows):
        out_rows[0, :count] = torch.arange(first, first + count, dtype=torch.float64)
        out_rows[1, :count] = -torch.arange(first, first + count, dtype=torch.float64)
    pipe.step(evaluate)
    got = pipe.assemble()
    assert torch.equal(got[0], torch.arange(300, dtype=torch.float64)) and torch.equal(got[1], -got[0])


def _percentile_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from hallthrusterpem_amd import percentiles as P
        rng = np.random.default_rng(123)                         # every rank builds the same global array, keeps its rows
        n, m = 30_011, 7
        a = rng.lognormal(0.0, 3.0, (n, m)) * rng.choice([-1.0, 1.0], (n, m))
        a[rng.random(n) < 0.3, 1] = 1e-20                        # 30 % ties at one value
        a[:, 2] = 4.5                                            # a constant column
        a[::11, 3] = np.inf
        a[17, 4] = np.nan                                        # NaN on ONE rank only: the column is NaN everywhere
        edges = [0, 17_000, n] if world == 2 else [0, 0, 12_345, n]          # ragged shards; world 3: rank 0 holds NO rows
        mine = a[edges[rank]:edges[rank + 1]]
        for pcts in ([25.0, 75.0], [5.0, 50.0, 95.0], 50.0, [0.0, 100.0, 33.3, 99.99, 1e-3]):
            want = np.percentile(a, pcts, axis=0)
            for method in ('select', 'levels'):
                got = P.column_percentiles_numpy(mine, pcts, method=method)
                assert got.shape == np.shape(want) and np.array_equal(got, want, equal_nan=True), (rank, pcts, method)
        # ties heavier than a candidate list may be long: the selection hands the pass to the level loop and still equals numpy
        keep, P.LIST_CAP = P.LIST_CAP, 64
        try:
            got = P.column_percentiles_numpy(mine, [5.0, 50.0, 95.0])
            assert np.array_equal(got, np.percentile(a, [5.0, 50.0, 95.0], axis=0), equal_nan=True), rank
        finally:
            P.LIST_CAP = keep
        Path(out_dir, f'ok{rank}').write_text('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_percentiles_equal_numpy_on_the_union_of_the_shards(tmp_path, world):
    """hallthrusterpem_amd.percentiles.sharded_percentiles on ragged shards (one rank without rows, a NaN on one rank only, 30 %
    ties, a constant column): the four-pass selection -- min / max and two histograms all-reduced, padded candidate lists
    all-gathered -- and the level loop it falls back to, both equal to np.percentile of ALL rows bit for bit on every rank.
    Every stage runs on its numpy restatement here (the device kernels are held to it stage by stage in test_quantiles.py)."""
    mp.spawn(_percentile_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f'ok{r}').exists() for r in range(world))


def _filter_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from hallthrusterpem_amd.drivers import filter_outputs
        rng = np.random.default_rng(321)                         # every rank builds the same campaign and keeps its samples
        n = 20_003
        out = {'T_c': rng.normal(0.08, 0.01, n), 'j_ion': rng.lognormal(0.0, 1.0, (n, 91)), 'j_ion_coords': np.zeros((n, 91))}
        out['T_c'][::500] = 5.0
        out['j_ion'][5::2000] *= 1e4
        out['j_ion'][7, 3] = np.nan
        nan_all, outl_all = filter_outputs(out)                                 # the reference's masks of the whole data set (the default, process group or not)
        edges = [0, 12_000, n] if world == 2 else [0, 0, 9_000, n]
        lo, hi = edges[rank], edges[rank + 1]
        for mine in ({k: v[lo:hi] for k, v in out.items()}, {k: torch.from_numpy(v[lo:hi]) for k, v in out.items()}):
            nan_r, outl_r = filter_outputs(mine, sharded=True)                     # (opt-in: the default treats a rank's data as a whole data set)
            for k in nan_all:
                assert np.array_equal(np.asarray(nan_r[k]), nan_all[k][lo:hi]) and np.array_equal(np.asarray(outl_r[k]), outl_all[k][lo:hi]), (rank, k)
        assert outl_all['T_c'].sum() >= n // 500 and outl_all['j_ion'].sum() >= 5
        # a sharded call is a collective: ranks that bring different variables are told so (on every rank) instead of hanging
        odd = {'T_c': out['T_c'][lo:hi]} if rank == 0 else {'T_c': out['T_c'][lo:hi], 'j_ion': out['j_ion'][lo:hi]}
        try:
            filter_outputs(odd, sharded=True)
            raise AssertionError('mismatched sharded call was accepted')
        except ValueError as exc:
            assert 'do not bring the same variables' in str(exc)
        Path(out_dir, f'ok{rank}').write_text('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_filter_outputs_gives_the_masks_of_the_whole_data_set(tmp_path, world):
    """drivers.filter_outputs on one rank's samples of a campaign (numpy arrays and CPU tensors here; one rank without samples at
    world 3): p25 / p75 are those of all ranks' samples, so the local masks are the slices of the masks gen_data.py:125-174
    computes on the whole data set."""
    mp.spawn(_filter_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f'ok{r}').exists() for r in range(world))


def test_percentiles_of_more_columns_than_one_kernel_call_takes():
    """More than 256 columns go through the kernels 256 at a time, so the bins per level are bounded by the CHUNK's width: sized
    from the whole width (round 2) they came out as 1 for m > 3072 with three percentiles, the ranges never narrowed and the
    level loop gave up after 80 passes.  No process group: one rank."""
    from hallthrusterpem_amd import percentiles as P
    assert P.level_bins(256, 6) == 16 and P.level_bins(91, 6) == 64 and P.level_bins(256, 2) == 64
    rng = np.random.default_rng(4)
    a = rng.standard_normal((50, 3100))
    want = np.percentile(a, [5.0, 50.0, 95.0], axis=0)
    cols = P.NumpyColumns(a)
    kmin, kmax, _ = cols.minmax()
    rp, rn, _ = P.linear_ranks(50, [5.0, 50.0, 95.0])
    ranks = np.stack([rp, rn], axis=1).reshape(-1)
    # the host-level loop exactly as DeviceColumns.levels_pass drives it for wide arrays: bins from the chunk width
    keys = P.narrow_by_levels(lambda lo, hi, b: P.local_hist_numpy(a, lo, hi, b), ranks, kmin, kmax, chunk_columns=256)
    assert np.array_equal(np.sort(a, axis=0)[ranks].T, P.value_of(keys))
    for method in ('select', 'levels'):
        assert np.array_equal(P.column_percentiles_numpy(a, [5.0, 50.0, 95.0], method=method), want)
