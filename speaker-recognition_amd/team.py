"""One plain Python process, several GPUs.

The reference's experiment scripts call `ubm.EM_split(..., num_thread=cpu_count())`, `stat_server.accumulate_stat(...)`,
`fa.total_variability(...)` from ONE process (ubm.py:44, i-vector.py:133 of the reference repository) and fan out over
CPU cores inside (sidekit/mixture.py:570-588, sidekit/sidekit_wrappers.py:141-214).  The same unmodified caller reaches
several GPUs here when the environment names them:

    SKM_DEVICES=0,1,2,3,4,5,6,7 python ubm.py          (or SKM_DEVICES=all)

The fan-out methods (`fanout` below) then run their body once per GPU in a host thread of its own -- rank r = the r-th
listed device, its own HipEngine / SkmCtx, its own shallow copy of the object -- and `dist`'s collectives, which the
multi-process path takes to torch.distributed, rendezvous the threads and issue ONE grouped RCCL call over all devices
through libskm (skm_comm_init_all, skm_allreduce_f64, skm_reduce_scatter_f64, skm_allgather_bytes: include/skm.h).  The
per-rank code is byte for byte the code a `torch.distributed.run` rank executes.  Rank 0 works on the caller's own
object, so the trained model / the filled arrays are where the reference leaves them.

`backend="host"` (tests, and rehearsals of several ranks on one GPU) sums / concatenates through host memory instead
of RCCL; SKM_TEAM_BACKEND=host selects it.
"""
import copy
import ctypes
import functools
import os
import threading

_tls = threading.local()
_process_team = None
_process_team_key = None
_lock = threading.Lock()


def current():
    """(team, rank) of the calling thread, or (None, 0) outside a fan-out."""
    return getattr(_tls, "team", None), getattr(_tls, "rank", 0)


def engine_of_thread():
    return getattr(_tls, "engine", None)


class TeamError(RuntimeError):
    pass


class Team(object):
    def __init__(self, devices, backend="rccl", engine_factory=None):
        self.devices = [int(d) for d in devices]
        self.world = len(self.devices)
        assert self.world >= 1
        self.backend = backend
        self._engine_factory = engine_factory
        self._engines = [None] * self.world
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world
        self._result = None
        self._comm = None
        self._lib = None

    # ------------------------------------------------------------------ engines, threads
    def engine(self, rank):
        if self._engines[rank] is None:
            if self._engine_factory is not None:
                self._engines[rank] = self._engine_factory(rank, self.devices[rank])
            else:
                from .engine import HipEngine
                self._engines[rank] = HipEngine(self.devices[rank])
        return self._engines[rank]

    def run(self, body):
        """body(rank) in one thread per device; returns the list of results (rank order).  The first exception of any
        rank is re-raised after every thread has ended (the barrier is broken so that no rank waits for a dead one)."""
        results, errors = [None] * self.world, [None] * self.world
        self._barrier.reset()

        def work(rank):
            _tls.team, _tls.rank = self, rank
            try:
                eng = self.engine(rank)
                _tls.engine = eng
                torch = getattr(eng, "torch", None)
                if torch is not None and str(getattr(eng, "device", "cpu")).startswith("cuda"):
                    torch.cuda.set_device(eng.device)         # the current device is a per-thread setting
                results[rank] = body(rank)
            except BaseException as e:      # noqa: B902 -- re-raised in the caller's thread
                errors[rank] = e
                self._barrier.abort()
            finally:
                _tls.team, _tls.rank, _tls.engine = None, 0, None

        threads = [threading.Thread(target=work, args=(r,), name="skm-rank-%d" % r) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        first = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if first:
            raise first[0]
        broken = [e for e in errors if e is not None]
        if broken:
            raise TeamError("a rank left a collective early") from broken[0]
        return results

    # ------------------------------------------------------------------ collectives (called by every rank thread)
    def barrier(self):
        self._barrier.wait()

    def _meet(self, rank, item, leader_fn):
        """Every rank deposits `item`; rank 0 runs leader_fn(list of items) once all are there; every rank gets its
        result after it has run."""
        self._slots[rank] = item
        self._barrier.wait()
        if rank == 0:
            try:
                self._result = leader_fn(list(self._slots))
            except BaseException:
                self._barrier.abort()
                raise
        self._barrier.wait()
        out = self._result
        self._barrier.wait()            # nobody deposits for the next collective before everyone has read this one
        return out

    def _rccl(self):
        if self._comm is None:
            from .engine import load_library
            self._lib = load_library()
            comm = ctypes.c_void_p()
            devs = (ctypes.c_int * self.world)(*self.devices)
            rc = self._lib.skm_comm_init_all(self.world, devs, ctypes.byref(comm))
            if rc != 0:
                raise TeamError("skm_comm_init_all: " + self._lib.skm_last_error().decode("utf-8", "replace"))
            self._comm = comm
        return self._lib, self._comm

    @staticmethod
    def _ptrs(values):
        return (ctypes.c_void_p * len(values))(*[ctypes.c_void_p(int(v)) for v in values])

    def _stream_of(self, t):
        import torch
        return torch.cuda.current_stream(t.device).cuda_stream

    def _check(self, lib, rc, what):
        if rc != 0:
            raise TeamError("%s: %s" % (what, lib.skm_last_error().decode("utf-8", "replace")))

    def allreduce_(self, rank, t, op="sum"):
        """In-place sum / maximum of a float64 tensor over the ranks."""
        import torch
        assert t.dtype == torch.float64 and t.is_contiguous(), "team all-reduce: contiguous float64 buffers"
        if self.backend == "rccl" and t.is_cuda:
            def lead(items):
                lib, comm = self._rccl()
                self._check(lib, lib.skm_allreduce_f64(comm, self._ptrs([i[0] for i in items]), items[0][2],
                                                       1 if op == "max" else 0, self._ptrs([i[1] for i in items])),
                            "skm_allreduce_f64")
            self._meet(rank, (t.data_ptr(), self._stream_of(t), t.numel()), lead)
            return t

        def lead(items):
            if items[0].is_cuda:
                for i in items:
                    torch.cuda.current_stream(i.device).synchronize()
            host = [i.detach().cpu() for i in items]
            tot = host[0].clone()
            for h in host[1:]:
                tot = torch.maximum(tot, h) if op == "max" else tot + h
            for i in items:
                i.copy_(tot)
            if items[0].is_cuda:
                for i in items:
                    torch.cuda.current_stream(i.device).synchronize()
        self._meet(rank, t, lead)
        return t

    def allgather(self, rank, send):
        """[world * rows, ...] tensor on every rank: the ranks' equally shaped `send` blocks in rank order."""
        import torch
        send = send.contiguous()
        out = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if self.backend == "rccl" and send.is_cuda:
            def lead(items):
                lib, comm = self._rccl()
                self._check(lib, lib.skm_allgather_bytes(comm, self._ptrs([i[0] for i in items]),
                                                         self._ptrs([i[1] for i in items]), items[0][3],
                                                         self._ptrs([i[2] for i in items])), "skm_allgather_bytes")
            self._meet(rank, (send.data_ptr(), out.data_ptr(), self._stream_of(send),
                              send.numel() * send.element_size()), lead)
            return out

        def lead(items):
            if items[0][0].is_cuda:
                for s, _ in items:
                    torch.cuda.current_stream(s.device).synchronize()
            full = torch.cat([s.detach().cpu() for s, _ in items], dim=0)
            for _, o in items:
                o.copy_(full)
            if items[0][0].is_cuda:
                for s, _ in items:
                    torch.cuda.current_stream(s.device).synchronize()
        self._meet(rank, (send, out), lead)
        return out

    def reduce_scatter(self, rank, t):
        """Sum over ranks of a [world * n, ...] float64 tensor; rank r gets rows [r * n, (r + 1) * n)."""
        import torch
        assert t.dtype == torch.float64
        t = t.contiguous()
        n = t.shape[0] // self.world
        assert n * self.world == t.shape[0]
        out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if self.backend == "rccl" and t.is_cuda:
            def lead(items):
                lib, comm = self._rccl()
                self._check(lib, lib.skm_reduce_scatter_f64(comm, self._ptrs([i[0] for i in items]),
                                                            self._ptrs([i[1] for i in items]), items[0][3],
                                                            self._ptrs([i[2] for i in items])), "skm_reduce_scatter_f64")
            self._meet(rank, (t.data_ptr(), out.data_ptr(), self._stream_of(t), out.numel()), lead)
            return out

        def lead(items):
            if items[0][0].is_cuda:
                for s, _ in items:
                    torch.cuda.current_stream(s.device).synchronize()
            tot = items[0][0].detach().cpu().clone()
            for s, _ in items[1:]:
                tot += s.detach().cpu()
            for r, (_, o) in enumerate(items):
                o.copy_(tot[r * n:(r + 1) * n])
            if items[0][0].is_cuda:
                for s, _ in items:
                    torch.cuda.current_stream(s.device).synchronize()
        self._meet(rank, (t, out), lead)
        return out

    def close(self):
        if self._comm is not None and self._lib is not None:
            self._lib.skm_comm_destroy(self._comm)
            self._comm = None


# ---------------------------------------------------------------------- the process-wide team named by the environment
def _parse_devices(spec):
    spec = spec.strip()
    if not spec:
        return None
    if spec == "all":
        import torch
        return list(range(torch.cuda.device_count()))
    return [int(x) for x in spec.replace(" ", "").split(",") if x != ""]


def process_team():
    """The team SKM_DEVICES asks for, or None: unset, a single device (unless SKM_TEAM_FORCE=1, which takes the fan-out
    and the collectives with a team of one: the single-GPU test of this path), a thread that already belongs to a team,
    or a process that is a torch.distributed rank (one mechanism at a time)."""
    global _process_team, _process_team_key
    if getattr(_tls, "team", None) is not None:
        return None
    spec = os.environ.get("SKM_DEVICES", "")
    backend = os.environ.get("SKM_TEAM_BACKEND", "rccl")
    force = os.environ.get("SKM_TEAM_FORCE") == "1"
    if not spec:
        return None
    try:
        import torch.distributed as td
        if td.is_available() and td.is_initialized():
            return None
    except Exception:
        pass
    devices = _parse_devices(spec)
    if not devices or (len(devices) < 2 and not force):
        return None
    key = (tuple(devices), backend)
    with _lock:
        if _process_team is None or _process_team_key != key:
            if _process_team is not None:
                _process_team.close()
            _process_team = Team(devices, backend=backend)
            _process_team_key = key
        return _process_team


def set_process_team(team):
    """Tests: install a team object directly (engine_factory / host backend) or remove it with None."""
    global _process_team, _process_team_key
    with _lock:
        _process_team = team
        _process_team_key = ("explicit",) if team is not None else None


def _active_team():
    if getattr(_tls, "team", None) is not None:
        return None
    if _process_team is not None and _process_team_key == ("explicit",):
        return _process_team
    return process_team()


def private_copy(x, team, rank):
    """What rank `rank` gets for an argument of a fan-out method.  Objects that hold an engine (Mixture, StatServer,
    FactorAnalyser, FeaturesServer) are shallow-copied and bound to the rank's engine -- they share their arrays with the
    caller's object, which the per-rank code only reads; a features server is also copied because the reference's
    loops flip its `keep_all_features` flag as they go."""
    if hasattr(x, "_engine") and not isinstance(x, type):
        y = copy.copy(x)
        y._engine = team.engine(rank)
        return y
    if hasattr(x, "keep_all_features"):
        return copy.copy(x)
    if isinstance(x, (list, tuple)) and any(hasattr(e, "_engine") for e in x):
        return type(x)(private_copy(e, team, rank) for e in x)
    return x


def fanout(method):
    """Method decorator: with a process team, run the body once per GPU (see the module docstring); without, call it."""
    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        team = _active_team()
        if team is None:
            return method(self, *args, **kwargs)
        keep = getattr(self, "_engine", None)

        def body(rank):
            me = self if rank == 0 else copy.copy(self)
            me._engine = team.engine(rank)
            a = [private_copy(x, team, rank) for x in args]
            k = {n: private_copy(v, team, rank) for n, v in kwargs.items()}
            return method(me, *a, **k)
        try:
            return team.run(body)[0]
        finally:
            self._engine = keep
    return wrapper
