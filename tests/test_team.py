"""One process, several "GPUs" (speaker-recognition_amd/team.py) on CPU: the fan-out methods run once per rank in threads,
the oracle-backed test double stands in for the HIP engines and the collectives go through host memory
(backend="host").  The caller is the reference's single-process script: it calls EM_split / accumulate_stat /
total_variability / extract_ivectors on ITS objects and must find on them what the reference leaves there."""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import sidekit_amd as sk                    # noqa: E402
from fake_engine import OracleEngine        # noqa: E402
from sidekit_amd import team as skteam      # noqa: E402


@pytest.fixture(params=[2, 3])
def team(request):
    seen = []

    def factory(rank, device):
        seen.append((rank, device, threading.current_thread().name))
        return OracleEngine()
    t = skteam.Team(list(range(request.param)), backend="host", engine_factory=factory)
    skteam.set_process_team(t)
    try:
        yield t
    finally:
        skteam.set_process_team(None)


def _ubm(g, eng=None):
    u = sk.Mixture(engine=eng)
    u.w, u.mu, u.invcov = g["w"].copy(), g["mu"].copy(), g["invcov"].copy()
    u.cov_var_ctl = 1.0 / u.invcov
    u._compute_all()
    return u


def test_collectives_of_the_host_backend(team):
    import torch
    w = team.world

    def body(rank):
        assert sk.dist.in_team() and sk.dist.is_dist() and sk.dist.rank() == rank and sk.dist.world_size() == w
        assert sk.default_engine() is team.engine(rank)
        t = torch.full((5,), float(rank + 1), dtype=torch.float64)
        sk.dist.allreduce_sum_(t)
        m = torch.tensor([float(rank)], dtype=torch.float64)
        sk.dist.allreduce_max_(m)
        counts = [2 + (r == 0) for r in range(w)]                    # ragged row blocks
        rows = torch.full((counts[rank], 3), float(rank), dtype=torch.float32)
        full = sk.dist.allgather_rows(rows, counts)
        big = torch.arange(w * 2 * 4, dtype=torch.float64).reshape(w * 2, 4) * (rank + 1)
        part = sk.dist.reduce_scatter_rows(big)
        sk.dist.barrier()
        return t, m, full, part
    out = team.run(body)
    tot = w * (w + 1) / 2
    for rank, (t, m, full, part) in enumerate(out):
        assert (t == tot).all() and m.item() == w - 1
        assert full.shape == (2 * w + 1, 3)
        assert [int(v) for v in full[:, 0]] == [0, 0, 0] + [r for r in range(1, w) for _ in range(2)]
        ref = np.arange(w * 2 * 4, dtype=np.float64).reshape(w * 2, 4) * tot
        assert np.array_equal(part.numpy(), ref[2 * rank:2 * rank + 2])
    assert not sk.dist.in_team() and sk.dist.world_size() == 1


def test_a_failing_rank_ends_the_fan_out(team):
    def body(rank):
        if rank == 1:
            raise ValueError("rank 1 gives up")
        sk.dist.barrier()
    with pytest.raises(ValueError, match="rank 1 gives up"):
        team.run(body)
    assert team.run(lambda rank: rank) == list(range(team.world))      # and the team is usable again


def test_single_process_script_reaches_every_rank(team, tmp_path):
    # ---- EM_split on the caller's Mixture (mixture.py:643-733): frames sharded by rows
    g = np.load(os.path.join(GOLDEN, "em_split.npz"))
    chunks = {"f%d" % i: g["chunk%d" % i] for i in range(4)}
    ubm = sk.Mixture()
    llk = ubm.EM_split(features_server=sk.FeaturesServer(features=chunks), feature_list=list(chunks), distrib_nb=8)
    assert np.abs(np.array(llk) - g["llk"]).max() / np.abs(g["llk"]).max() < 1e-6
    assert np.abs(ubm.mu - g["mu"]).max() / np.abs(g["mu"]).max() < 1e-5 and np.abs(ubm.w - g["w"]).max() < 1e-6
    assert ubm._engine is None                         # the caller's object is not left bound to a rank's engine
    assert all(e is not None for e in team._engines)   # every rank had work (its engine was made in its thread)

    # ---- EM_uniform
    gu = np.load(os.path.join(GOLDEN, "em_uniform.npz"))
    u2 = sk.Mixture()
    l2 = u2.EM_uniform(gu["cep"], 4)
    assert np.abs(np.array(l2) - gu["llk"]).max() / np.abs(gu["llk"]).max() < 1e-6

    # ---- EM_split with VAD labels and keep_all_features = True (the reference flips the server's flag as it goes)
    gv = np.load(os.path.join(GOLDEN, "em_split_vad.npz"))
    fsv = sk.FeaturesServer(features={"f%d" % i: gv["chunk%d" % i] for i in range(3)},
                            vad={"f%d" % i: gv["vad%d" % i] for i in range(3)}, keep_all_features=True)
    uv = sk.Mixture()
    lv = uv.EM_split(features_server=fsv, feature_list=["f0", "f1", "f2"], distrib_nb=4)
    assert np.abs(np.array(lv) - gv["llk"]).max() / np.abs(gv["llk"]).max() < 1e-6
    assert np.abs(uv.mu - gv["mu"]).max() / np.abs(gv["mu"]).max() < 1e-5

    # ---- accumulate_stat: every rank fills ITS rows of the caller's arrays, nothing is gathered
    ga = np.load(os.path.join(GOLDEN, "accumulate_stat.npz"))
    off = np.concatenate([[0], np.cumsum(ga["lens"])])
    names = ["s%03d" % i for i in range(len(ga["lens"]))]
    fs = sk.FeaturesServer(features={n: ga["frames"][off[i]:off[i + 1]] for i, n in enumerate(names)})
    ub = _ubm(ga)
    ss = sk.StatServer(sk.IdMap().set(["m"] * len(names), names), ubm=ub)
    sk.dist.reset_payload()
    ss.accumulate_stat(ubm=ub, feature_server=fs, seg_indices=range(len(names)))
    assert np.abs(ss.stat0 - ga["stat0"]).max() / np.abs(ga["stat0"]).max() < 1e-6
    assert np.abs(ss.stat1 - ga["stat1"]).max() / np.abs(ga["stat1"]).max() < 1e-6
    assert sum(sk.dist.PAYLOAD.values()) == 0 and list(ss.local_rows) == list(range(len(names)))
    ss2 = sk.StatServer(sk.IdMap().set(["m"] * len(names), names), ubm=ub)       # seg_indices=None: arrays allocated here
    ss2.accumulate_stat(ubm=ub, feature_server=fs)
    assert np.array_equal(ss2.stat1, ss.stat1)

    # ---- TV training and extraction: sessions sharded, accumulators reduced over the ranks
    gt = np.load(os.path.join(GOLDEN, "tv.npz"))
    S = gt["stat0_f32"].shape[0]
    st = sk.StatServer()
    st.modelset = np.array(["m%d" % (i % 3) for i in range(S)])
    st.segset = np.array(["s%03d" % i for i in range(S)])
    st.start = np.empty(S, "|O")
    st.stop = np.empty(S, "|O")
    st.stat0 = gt["stat0_f32"].astype(np.float64)
    st.stat1 = gt["stat1_f32"].astype(np.float64)
    stat_file = str(tmp_path / "stat.npz")
    st.write(stat_file)
    ut = _ubm(gt)
    fa = sk.FactorAnalyser()
    fa.total_variability(stat_file, ut, 6, nb_iter=3, min_div=True, tv_init=gt["tv_init"].copy(), batch_size=7,
                         output_file_name=str(tmp_path / "tv"))
    assert np.abs(fa.F - gt["F_it3_pool"]).max() / np.abs(gt["F_it3_pool"]).max() < 1e-5
    assert os.path.exists(str(tmp_path / "tv") + fa._ext())
    fa.F = gt["F_it3_pool"].copy()
    ivs, ivsig = fa.extract_ivectors(ut, stat_file, batch_size=7, uncertainty=True)
    assert np.abs(ivs.stat1 - gt["iv"]).max() / np.abs(gt["iv"]).max() < 1e-5
    assert np.abs(ivsig - gt["iv_sigma"]).max() / np.abs(gt["iv_sigma"]).max() < 1e-5
    ivs1 = fa.extract_ivectors(ut, st, batch_size=7)
    assert np.abs(ivs1.stat1 - gt["iv"]).max() / np.abs(gt["iv"]).max() < 1e-5
    # an unseeded start (tv_init=None) must be ONE matrix on every rank: the run is then reproducible from the seed
    np.random.seed(3)
    fa2 = sk.FactorAnalyser()
    fa2.total_variability_single(stat_file, ut, 4, nb_iter=1, min_div=False)
    np.random.seed(3)
    skteam.set_process_team(None)
    try:
        fa3 = sk.FactorAnalyser(engine=OracleEngine())
        fa3.total_variability_single(stat_file, _ubm(gt, OracleEngine()), 4, nb_iter=1, min_div=False)
    finally:
        skteam.set_process_team(team)
    assert np.abs(fa2.F - fa3.F).max() / np.abs(fa3.F).max() < 1e-8


def test_environment_names_the_team(monkeypatch):
    monkeypatch.delenv("SKM_DEVICES", raising=False)
    assert skteam.process_team() is None
    monkeypatch.setenv("SKM_DEVICES", "0")
    assert skteam.process_team() is None                     # one device: nothing to fan out over
    monkeypatch.setenv("SKM_DEVICES", "0, 2,5")
    monkeypatch.setenv("SKM_TEAM_BACKEND", "host")
    t = skteam.process_team()
    assert t is not None and t.devices == [0, 2, 5] and t.backend == "host"
    assert skteam.process_team() is t                        # cached
    monkeypatch.setenv("SKM_DEVICES", "1,3")
    assert skteam.process_team().devices == [1, 3]
    skteam.set_process_team(None)
