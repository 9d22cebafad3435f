"""Thin torch.distributed helpers: one process per GPU, RCCL ("nccl" backend) over xGMI on the
GPU box, gloo in the CPU tests.  Replaces sidekit/sidekit_mpi.py's Reduce/bcast pattern
(sidekit_mpi.py:455-481, 186-214): every rank all-reduces ONE packed float64 buffer and then runs
the identical deterministic M-step, so no broadcast is needed."""
import numpy as np


def _td():
    import torch.distributed as td
    return td


def is_dist():
    try:
        td = _td()
        return td.is_available() and td.is_initialized() and td.get_world_size() > 1
    except Exception:
        return False


def rank():
    return _td().get_rank() if is_dist() else 0


def world_size():
    return _td().get_world_size() if is_dist() else 1


def shard_bounds(n, k=None, r=None):
    """[start, stop) of part r when n items are cut like numpy.array_split(n, k)
    (the reference's partition: sidekit_mpi.py:408-410, sidekit_wrappers.py:168)."""
    k = world_size() if k is None else k
    r = rank() if r is None else r
    base, extra = divmod(int(n), int(k))
    start = r * base + min(r, extra)
    return start, start + base + (1 if r < extra else 0)


def allreduce_sum_(t):
    """In-place sum over ranks of a torch tensor (device tensor -> RCCL, CPU tensor -> gloo)."""
    if is_dist():
        _td().all_reduce(t)
    return t


def allgather_rows(t, counts):
    """Concatenate per-rank row blocks (rank r contributes counts[r] rows) on every rank."""
    if not is_dist():
        return t
    import torch
    td = _td()
    parts = [torch.empty((int(c),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for c in counts]
    td.all_gather(parts, t.contiguous())
    return torch.cat(parts, dim=0)


def barrier():
    if is_dist():
        _td().barrier()
