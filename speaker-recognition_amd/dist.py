"""Thin collective helpers.  Two ways to be a rank: one process per GPU under torch.distributed (RCCL = the "nccl"
backend over xGMI on the GPU box, gloo in the CPU tests), or one THREAD per GPU of a single process (`team.py`:
SKM_DEVICES, grouped RCCL calls through libskm).  Replaces sidekit/sidekit_mpi.py's Reduce/bcast pattern
(sidekit_mpi.py:455-481, 186-214): every rank all-reduces ONE packed float64 buffer and then runs
the identical deterministic M-step, so no broadcast is needed."""
import os

import numpy as np

from . import team as _team


def _td():
    import torch.distributed as td
    return td


def is_dist():
    """True when torch.distributed is initialised with more than one rank.  SKM_FORCE_DIST=1 also takes the
    collective code paths in a world of ONE rank (tests/test_gpu_rccl.py: the RCCL calls, buffer layouts and
    the component-sharded M-step run for real on a single-GPU box)."""
    tm, _ = _team.current()
    if tm is not None:
        return tm.world > 1 or os.environ.get("SKM_FORCE_DIST") == "1" or os.environ.get("SKM_TEAM_FORCE") == "1"
    try:
        td = _td()
        if not (td.is_available() and td.is_initialized()):
            return False
        return td.get_world_size() > 1 or os.environ.get("SKM_FORCE_DIST") == "1"
    except Exception:
        return False


def rank():
    tm, r = _team.current()
    if tm is not None:
        return r
    return _td().get_rank() if is_dist() else 0


def world_size():
    tm, _ = _team.current()
    if tm is not None:
        return tm.world
    return _td().get_world_size() if is_dist() else 1


def in_team():
    """True in a rank THREAD of a single process (the ranks then share the caller's host objects)."""
    return _team.current()[0] is not None


def shard_bounds(n, k=None, r=None):
    """[start, stop) of part r when n items are cut like numpy.array_split(n, k)
    (the reference's partition: sidekit_mpi.py:408-410, sidekit_wrappers.py:168)."""
    k = world_size() if k is None else k
    r = rank() if r is None else r
    base, extra = divmod(int(n), int(k))
    start = r * base + min(r, extra)
    return start, start + base + (1 if r < extra else 0)


# Bytes this rank handed to each kind of collective since the last reset (payload = the rank's own contribution).
# The CPU tests read them to pin the data plane: per-session rows travel as the owner's rows only.
PAYLOAD = {"all_reduce": 0, "all_gather": 0, "reduce_scatter": 0}


def reset_payload():
    for k in PAYLOAD:
        PAYLOAD[k] = 0


def _nbytes(t):
    return int(t.numel()) * int(t.element_size())


def _gloo_on_device(t):
    """A device tensor under the gloo backend: only bench.py's one-GPU rehearsal (SKM_BENCH_REHEARSE=1) gets here.
    gloo's own handling of device tensors hung on this ROCm build, so the collective runs on a host copy."""
    return t.is_cuda and _td().get_backend() == "gloo"


def allreduce_sum_(t):
    """In-place sum over ranks of a torch tensor (device tensor -> RCCL, CPU tensor -> gloo)."""
    tm, r = _team.current()
    if tm is not None:
        if is_dist():
            PAYLOAD["all_reduce"] += _nbytes(t)
            tm.allreduce_(r, t, "sum")
        return t
    if is_dist():
        PAYLOAD["all_reduce"] += _nbytes(t)
        if _gloo_on_device(t):
            h = t.cpu()
            _td().all_reduce(h)
            t.copy_(h)
        else:
            _td().all_reduce(t)
    return t


def allreduce_max_(t):
    """In-place maximum over ranks (a scalar decision every rank must take alike)."""
    tm, r = _team.current()
    if tm is not None:
        if is_dist():
            tm.allreduce_(r, t, "max")
        return t
    if is_dist():
        op = _td().ReduceOp.MAX
        if _gloo_on_device(t):
            h = t.cpu()
            _td().all_reduce(h, op=op)
            t.copy_(h)
        else:
            _td().all_reduce(t, op=op)
    return t


def allgather_rows(t, counts):
    """Concatenate per-rank row blocks (rank r contributes counts[r] rows) on every rank: the replacement of the
    reference's Gatherv (sidekit_mpi.py:313-314).  Each rank sends ITS rows only; blocks are padded to the largest
    count (numpy.array_split shards differ by at most one row) so that one all_gather_into_tensor does it."""
    if not is_dist():
        return t
    import torch
    tm, tr = _team.current()
    td = _td() if tm is None else None
    counts = [int(c) for c in counts]
    assert t.shape[0] == counts[rank()], (t.shape, counts, rank())
    w, m = len(counts), max(counts)
    tail = tuple(t.shape[1:])
    if m == 0:
        return t
    send = t.contiguous()
    if t.shape[0] < m:
        send = torch.zeros((m,) + tail, dtype=t.dtype, device=t.device)
        send[:t.shape[0]] = t
    PAYLOAD["all_gather"] += _nbytes(send)
    if tm is not None:
        out = tm.allgather(tr, send)
        if all(c == m for c in counts):
            return out
        return torch.cat([out[r * m:r * m + counts[r]] for r in range(w)], dim=0)
    out = torch.empty((w * m,) + tail, dtype=t.dtype, device=t.device)
    if td.get_backend() == "nccl":
        td.all_gather_into_tensor(out, send)
    elif _gloo_on_device(t):
        h = torch.empty((w * m,) + tail, dtype=t.dtype)
        td.all_gather(list(h.view((w, m) + tail).unbind(0)), send.cpu())
        out.copy_(h)
    else:
        td.all_gather(list(out.view((w, m) + tail).unbind(0)), send)
    if all(c == m for c in counts):
        return out
    return torch.cat([out[r * m:r * m + counts[r]] for r in range(w)], dim=0)


def reduce_scatter_rows(t):
    """Sum over ranks of a [world * n, ...] tensor, rank r keeps rows [r*n, (r+1)*n).  RCCL: one
    reduce-scatter (each byte crosses xGMI once instead of twice for an all-reduce); gloo has no
    reduce-scatter, the CPU tests take all-reduce + slice."""
    if not is_dist():
        return t
    import torch
    w, r = world_size(), rank()
    n = t.shape[0] // w
    assert n * w == t.shape[0]
    t = t.contiguous()
    PAYLOAD["reduce_scatter"] += _nbytes(t)
    tm, tr = _team.current()
    if tm is not None:
        return tm.reduce_scatter(tr, t)
    td = _td()
    if td.get_backend() == "nccl":
        out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        td.reduce_scatter_tensor(out, t)
        return out
    if _gloo_on_device(t):
        h = t.cpu()
        td.all_reduce(h)
        return h[r * n:(r + 1) * n].to(t.device)
    td.all_reduce(t)
    return t[r * n:(r + 1) * n].clone()


def barrier():
    tm, _ = _team.current()
    if tm is not None:
        tm.barrier()
    elif is_dist():
        _td().barrier()
