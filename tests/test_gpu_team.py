"""One process, several GPUs (speaker-recognition_amd/team.py) on the one GPU this box has.
(1) SKM_DEVICES=0 with SKM_TEAM_FORCE=1: the fan-out with a team of ONE takes every multi-GPU branch and every
    collective goes through libskm's grouped RCCL calls (skm_comm_init_all -> ncclCommInitAll, skm_allreduce_f64,
    skm_reduce_scatter_f64, skm_allgather_bytes); results must equal the plain run bit for bit.
(2) two rank threads on the same GPU with the host backend and the REAL kernels: frames, sessions and components
    really split over two engines, collectives that really move data; against the single-process run.
(3) the entry points themselves through ctypes.
More than one RCCL rank needs more than one GPU: the world > 1 logic is covered on CPU by tests/test_team.py."""
import ctypes
import os

import numpy as np
import pytest

from test_gpu_rccl import _run_all

pytestmark = pytest.mark.gpu


def test_team_of_one_on_rccl_equals_the_plain_run(monkeypatch):
    import sidekit_amd as sk
    from sidekit_amd import team as skteam
    plain = _run_all(sk)
    monkeypatch.setenv("SKM_DEVICES", "0")
    monkeypatch.setenv("SKM_TEAM_FORCE", "1")
    monkeypatch.delenv("SKM_TEAM_BACKEND", raising=False)
    try:
        t = skteam.process_team()
        assert t is not None and t.world == 1 and t.backend == "rccl"
        fanned = _run_all(sk)
        assert t._comm is not None                    # the collectives went through skm_comm_init_all
        for k in plain:
            assert np.array_equal(fanned[k], plain[k]), k
    finally:
        skteam.set_process_team(None)


def test_two_rank_threads_on_one_gpu_equal_single_process():
    import sidekit_amd as sk
    from sidekit_amd import team as skteam
    plain = _run_all(sk)
    t = skteam.Team([0, 0], backend="host")
    skteam.set_process_team(t)
    try:
        two = _run_all(sk)
    finally:
        skteam.set_process_team(None)
    assert all(e is not None for e in t._engines) and t._engines[0] is not t._engines[1]
    # a session is computed whole on one rank
    assert np.array_equal(two["stat0"], plain["stat0"]) and np.array_equal(two["stat1"], plain["stat1"])

    def rel(a, b):
        return np.abs(a - b).max() / np.abs(b).max()
    # float32 products over other batches of sessions / frames; 7 chaotic EM stages for the UBM
    assert rel(two["F"], plain["F"]) < 1e-5 and rel(two["iv"], plain["iv"]) < 1e-5
    assert rel(two["iv_sigma"], plain["iv_sigma"]) < 1e-5
    assert rel(two["llk"], plain["llk"]) < 1e-5 and rel(two["mu"], plain["mu"]) < 1e-4 and rel(two["w"], plain["w"]) < 1e-4


def test_comm_entry_points():
    import torch
    import sidekit_amd as sk
    lib = sk.load_library()
    comm = ctypes.c_void_p()
    two = (ctypes.c_int * 2)(0, 0)
    assert lib.skm_comm_init_all(2, two, ctypes.byref(comm)) != 0          # one rank per GPU
    assert b"twice" in lib.skm_last_error()
    far = (ctypes.c_int * 1)(torch.cuda.device_count())
    assert lib.skm_comm_init_all(1, far, ctypes.byref(comm)) != 0
    one = (ctypes.c_int * 1)(0)
    assert lib.skm_comm_init_all(1, one, ctypes.byref(comm)) == 0, lib.skm_last_error()
    try:
        assert lib.skm_comm_size(comm) == 1
        dev = torch.device("cuda", 0)
        st = (ctypes.c_void_p * 1)(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        x = torch.arange(1000, dtype=torch.float64, device=dev)
        ref = x.clone()
        p = (ctypes.c_void_p * 1)(ctypes.c_void_p(x.data_ptr()))
        assert lib.skm_allreduce_f64(comm, p, x.numel(), 0, st) == 0, lib.skm_last_error()
        assert lib.skm_allreduce_f64(comm, p, x.numel(), 1, st) == 0, lib.skm_last_error()
        assert lib.skm_allreduce_f64(comm, p, x.numel(), 7, st) != 0
        out = torch.zeros_like(x)
        q = (ctypes.c_void_p * 1)(ctypes.c_void_p(out.data_ptr()))
        assert lib.skm_reduce_scatter_f64(comm, p, q, x.numel(), st) == 0, lib.skm_last_error()
        torch.cuda.synchronize()
        assert torch.equal(x, ref) and torch.equal(out, ref)
        b = torch.arange(37, dtype=torch.uint8, device=dev)
        o = torch.zeros_like(b)
        pb = (ctypes.c_void_p * 1)(ctypes.c_void_p(b.data_ptr()))
        po = (ctypes.c_void_p * 1)(ctypes.c_void_p(o.data_ptr()))
        assert lib.skm_allgather_bytes(comm, pb, po, 37, st) == 0, lib.skm_last_error()
        torch.cuda.synchronize()
        assert torch.equal(o, b)
        null = (ctypes.c_void_p * 1)(None)
        assert lib.skm_allreduce_f64(comm, null, 4, 0, st) != 0
    finally:
        assert lib.skm_comm_destroy(comm) == 0
