"""Multi-GPU sharding of independent work items (patches, sessions x reps).

The reference processes its work items one after another in a single process
(mono_cal_target/run_sr.py:358-360, mono_barcodes/run_sr.py:301); items never interact, so
the path shards with NO data-path collective: item i goes to rank i mod G (the order the
reference would reach it), every rank reconstructs its own items on its own GPU, and the
uint8 / float results are gathered on the host of rank 0.  torch.distributed (RCCL = "nccl"
on ROCm, "gloo" on CPU) is used only for that final gather and for barriers.
"""
import numpy as np


def shard_indices(n_items, rank, world):
    """Round-robin ownership: the items rank `rank` of `world` reconstructs."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_items, world))


def owner_of(item, world):
    return item % world


def _group_rank_world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def map_sharded(fn, n_items, group=None, dst=0):
    """fn(i) for the items this rank owns; on rank `dst` -> [fn(0), ..., fn(n_items - 1)] in item order (small host objects:
    one gather_object at the END of the job, the only communication of the sharded mode), None on the other ranks.
    Without an initialised process group it just loops."""
    import torch.distributed as dist
    rank, world = _group_rank_world(group)
    mine = {i: fn(i) for i in shard_indices(n_items, rank, world)}
    if world == 1:
        return [mine[i] for i in range(n_items)]
    parts = [None] * world if rank == dst else None
    dist.gather_object(mine, parts, dst=dst, group=group)
    if rank != dst:
        return None
    merged = {}
    for part in parts:
        merged.update(part)
    if sorted(merged) != list(range(n_items)):
        raise RuntimeError("sharding lost or duplicated items")
    return [merged[i] for i in range(n_items)]


def default_compute(lr, shifts_yx, kernel, factor, n_iter, step):
    """One item on this rank's GPU through libsrx: SAA then IBP -> (hr float64 numpy, errors list)."""
    from . import api
    saa = api.shift_and_add(lr, shifts_yx, factor)
    return api.ibp(lr, shifts_yx, kernel, saa, factor, n_iter, step, verbose=False)


def reconstruct_sharded(items, shifts_yx, kernel, factor=2, n_iter=80, step=0.5, compute=None, group=None, dst=0):
    """items: list of LR stacks [N, h, w] (every rank passes the same list; only owned ones are touched).
    Returns, on rank `dst`, a list with one (hr, errors) per item in the original order; None elsewhere."""
    compute = compute or default_compute

    def one(i):
        hr, errs = compute(items[i], shifts_yx, kernel, factor, n_iter, step)
        return np.asarray(hr), [float(e) for e in errs]

    return map_sharded(one, len(items), group=group, dst=dst)
