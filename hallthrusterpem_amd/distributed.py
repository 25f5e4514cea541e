"""Sample-sharded multi-GPU evaluation: one process per GPU, contiguous shards, one all-gather.

Samples are independent, so the only collective on the path is the final
`all_gather_into_tensor` of per-sample QoIs over RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests).
Nothing here touches the arithmetic: the evaluator is passed in.
"""
from typing import Callable, Dict, Tuple


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: lo = rank*n//world, so shards differ by at most one sample and
    concatenating them in rank order restores the global sample order."""
    if not (0 <= rank < world):
        raise ValueError(f'rank {rank} outside world of {world}')
    return (rank * n) // world, ((rank + 1) * n) // world


def max_shard(n: int, world: int) -> int:
    return max(shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world))


def all_gather_rows(local, n_total: int, group=None):
    """Gather a [k, n_local] tensor of per-sample rows from every rank into [k, n_total] (global sample order).

    One collective (`all_gather_into_tensor`).  Shards are padded to the longest one so the buffer is regular;
    the pad is dropped when the pieces are put back in rank order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    k = local.shape[0]
    width = max_shard(n_total, world)
    lo, hi = shard_bounds(n_total, world, rank)
    if local.shape[1] != hi - lo:
        raise ValueError(f'rank {rank} holds {local.shape[1]} samples, its shard has {hi - lo}')
    send = local if local.shape[1] == width else torch.nn.functional.pad(local, (0, width - local.shape[1]))
    # concatenation layout (world*k rows): accepted by both the RCCL and the gloo backends
    recv = torch.empty((world * k, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    recv = recv.view(world, k, width)
    out = torch.empty((k, n_total), dtype=local.dtype, device=local.device)
    for r in range(world):
        a, b = shard_bounds(n_total, world, r)
        out[:, a:b] = recv[r, :, : b - a]
    return out


def evaluate_sharded(n_total: int, make_inputs: Callable[[int, int], Dict], evaluate: Callable[[Dict], Dict],
                     qoi=('V_cc', 'div_angle', 'T_c'), group=None):
    """Forward UQ over `n_total` samples on all ranks of `group`.

    `make_inputs(lo, hi)` returns this rank's inputs for GLOBAL sample indices [lo, hi) (a counter-based sampler
    keyed by the global index makes the result independent of the number of ranks); `evaluate(inputs)` is the
    model (e.g. `models.pem_v0_coupled`).  Returns ({qoi: [n_total] tensor} gathered on every rank, local outputs).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(n_total, world, rank)
    local = evaluate(make_inputs(lo, hi))
    rows = torch.stack([torch.as_tensor(local[k]).reshape(-1) for k in qoi])
    full = rows if world == 1 else all_gather_rows(rows, n_total, group)
    return {k: full[i] for i, k in enumerate(qoi)}, local
