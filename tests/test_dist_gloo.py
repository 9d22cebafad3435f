"""Multi-rank host logic on CPU: world_size-2 gloo runs of the drop-in classes with the oracle-backed
test double (tests/fake_engine.py) in place of the HIP engine.  Checks that sharding frames / sessions
over ranks + all-reduce of the packed statistics gives every rank the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sidekit_amd as sk
        from fake_engine import OracleEngine
        from oracle import sidekit_ref as orc
        eng = OracleEngine()
        out = {}
        assert sk.dist.world_size() == world and sk.dist.rank() == rank
        # shard_bounds == numpy.array_split
        for n in (0, 1, 7, 10, 11):
            parts = np.array_split(np.arange(n), world)
            a, b = sk.dist.shard_bounds(n)
            assert list(range(a, b)) == list(parts[rank]), (n, a, b)

        # ---- EM_split: frames sharded by rows, one all-reduce per iteration
        g = np.load(os.path.join(GOLDEN, "em_split.npz"))
        chunks = {"f%d" % i: g["chunk%d" % i] for i in range(4)}
        ubm = sk.Mixture(engine=eng)
        llk = ubm.EM_split(features_server=sk.FeaturesServer(features=chunks), feature_list=list(chunks), distrib_nb=8)
        out["em_llk"] = np.abs(np.array(llk) - g["llk"]).max() / np.abs(g["llk"]).max()
        out["em_mu"] = np.abs(ubm.mu - g["mu"]).max() / np.abs(g["mu"]).max()
        out["em_w"] = np.abs(ubm.w - g["w"]).max()

        # ---- EM_uniform
        gu = np.load(os.path.join(GOLDEN, "em_uniform.npz"))
        u2 = sk.Mixture(engine=eng)
        l2 = u2.EM_uniform(gu["cep"], 4)
        out["emu_llk"] = np.abs(np.array(l2) - gu["llk"]).max() / np.abs(gu["llk"]).max()

        # ---- EM_split on files with VAD labels, keep_all_features = True: _init on every frame, E-steps on the
        # labelled ones; every rank loads only the files that overlap its row range (frame_counts)
        gv = np.load(os.path.join(GOLDEN, "em_split_vad.npz"))
        fsv = sk.FeaturesServer(features={"f%d" % i: gv["chunk%d" % i] for i in range(3)},
                                vad={"f%d" % i: gv["vad%d" % i] for i in range(3)}, keep_all_features=True)
        loaded = []
        real_stack = fsv.stack_features_parallel
        fsv.stack_features_parallel = lambda shows, **kw: (loaded.append(list(shows)), real_stack(shows, **kw))[1]
        uv = sk.Mixture(engine=eng)
        lv = uv.EM_split(features_server=fsv, feature_list=["f0", "f1", "f2"], distrib_nb=4)
        out["emv_llk"] = np.abs(np.array(lv) - gv["llk"]).max() / np.abs(gv["llk"]).max()
        out["emv_mu"] = np.abs(uv.mu - gv["mu"]).max() / np.abs(gv["mu"]).max()
        out["emv_flag"] = bool(fsv.keep_all_features)
        out["emv_loaded"] = loaded

        # ---- accumulate_stat: sessions sharded, every rank computes ITS rows; the row blocks are all-gathered
        # (no reduction on this path: the payload a rank sends is its own rows)
        ga = np.load(os.path.join(GOLDEN, "accumulate_stat.npz"))
        off = np.concatenate([[0], np.cumsum(ga["lens"])])
        names = ["s%03d" % i for i in range(len(ga["lens"]))]
        fs = sk.FeaturesServer(features={n: ga["frames"][off[i]:off[i + 1]] for i, n in enumerate(names)})
        ub = sk.Mixture(engine=eng)
        ub.w, ub.mu, ub.invcov = ga["w"].copy(), ga["mu"].copy(), ga["invcov"].copy()
        ub.cov_var_ctl = 1.0 / ub.invcov
        ub._compute_all()
        ss = sk.StatServer(sk.IdMap().set(["m"] * len(names), names), ubm=ub, engine=eng)
        sk.dist.reset_payload()
        ss.accumulate_stat(ubm=ub, feature_server=fs, seg_indices=range(len(names)), keep_on_device=True)
        out["bw0"] = np.abs(ss.stat0 - ga["stat0"]).max() / np.abs(ga["stat0"]).max()
        out["bw1"] = np.abs(ss.stat1 - ga["stat1"]).max() / np.abs(ga["stat1"]).max()
        a, b = sk.dist.shard_bounds(len(names))
        longest = max(hi - lo for lo, hi in (sk.dist.shard_bounds(len(names), world, r) for r in range(world)))
        out["bw_payload"] = dict(sk.dist.PAYLOAD)
        out["bw_payload_expected"] = longest * (ss.stat0.shape[1] + ss.stat1.shape[1]) * 8
        assert list(ss.local_rows) == list(range(a, b))
        # gather=False: the rank keeps its shard, nothing travels
        ss_local = sk.StatServer(sk.IdMap().set(["m"] * len(names), names), ubm=ub, engine=eng)
        sk.dist.reset_payload()
        ss_local.accumulate_stat(ubm=ub, feature_server=fs, gather=False)
        assert sum(sk.dist.PAYLOAD.values()) == 0
        assert np.array_equal(ss_local.stat1[a:b], ss.stat1[a:b])
        others = np.ones(len(names), dtype=bool)
        others[a:b] = False
        assert not ss_local.stat1[others].any()
        # the device-resident rows (float32, as a file would store them) feed TV without a host round trip
        res = ss.resident_rows()
        assert res is not None and list(res[0]) == list(range(a, b)) and res[2].dtype == torch.float32
        assert np.array_equal(res[2].numpy(), ss.stat1[a:b].astype(np.float32))

        # ---- TV training + extraction: sessions sharded, accumulators all-reduced
        gt = np.load(os.path.join(GOLDEN, "tv.npz"))
        S = gt["stat0_f32"].shape[0]
        st = sk.StatServer(engine=eng)
        st.modelset = np.array(["m%d" % (i % 3) for i in range(S)])
        st.segset = np.array(["s%03d" % i for i in range(S)])
        st.start = np.empty(S, "|O")
        st.stop = np.empty(S, "|O")
        st.stat0 = gt["stat0_f32"].astype(np.float64)
        st.stat1 = gt["stat1_f32"].astype(np.float64)
        stat_file = os.path.join(tmp, "stat_r%d.npz" % rank)
        st.write(stat_file)
        ut = sk.Mixture(engine=eng)
        ut.w, ut.mu, ut.invcov = gt["w"].copy(), gt["mu"].copy(), gt["invcov"].copy()
        ut.cov_var_ctl = 1.0 / ut.invcov
        ut._compute_all()
        fa = sk.FactorAnalyser(engine=eng)
        fa.total_variability(stat_file, ut, 6, nb_iter=3, min_div=True, tv_init=gt["tv_init"].copy(), batch_size=7,
                             output_file_name=os.path.join(tmp, "tv_r%d" % rank))
        out["tv_F"] = np.abs(fa.F - gt["F_it3_pool"]).max() / np.abs(gt["F_it3_pool"]).max()
        fa.F = gt["F_it3_pool"].copy()
        sk.dist.reset_payload()
        ivs, ivsig = fa.extract_ivectors(ut, stat_file, batch_size=7, uncertainty=True)
        out["iv"] = np.abs(ivs.stat1 - gt["iv"]).max() / np.abs(gt["iv"]).max()
        out["iv_sigma"] = np.abs(ivsig - gt["iv_sigma"]).max() / np.abs(gt["iv_sigma"]).max()
        out["iv_payload"] = dict(sk.dist.PAYLOAD)
        out["iv_payload_expected"] = max(hi - lo for lo, hi in (sk.dist.shard_bounds(S, world, r) for r in range(world))) * 2 * 6 * 8
        ivs1 = fa.extract_ivectors(ut, st, batch_size=7)                 # in-memory StatServer, no uncertainty
        out["iv_mem"] = np.abs(ivs1.stat1 - gt["iv"]).max() / np.abs(gt["iv"]).max()
        # ---- MAP adaptation + GMM-UBM scoring: test segments sharded, score matrix summed back
        out["sc"], out["map"] = _scoring_case(sk, eng)
        # ---- raw cepstra through the post-processing server (batch path), sessions sharded over ranks
        out["pp0"], out["pp1"] = _postproc_case(sk, eng, orc)
        q.put((rank, out))
    finally:
        td.destroy_process_group()


def _postproc_case(sk, eng, orc):
    """accumulate_stat fed by a FeaturesServer that post-processes raw frames (delta, double delta, cmvn, VAD
    selection through load_batch) against the oracle's per-session post_processing + accumulate_stat."""
    rs = np.random.RandomState(17)
    D0, C = 5, 8
    gmm = orc.random_gmm(C, 3 * D0, seed=2)
    ubm = sk.Mixture(engine=eng)
    ubm.w, ubm.mu, ubm.invcov = gmm.w.copy(), gmm.mu.copy(), gmm.invcov.copy()
    ubm.cov_var_ctl = 1.0 / ubm.invcov
    ubm._compute_all()
    fs = sk.FeaturesServer(feat_norm="cmvn", delta=True, double_delta=True, keep_all_features=True, engine=eng)
    names, sessions = [], []
    for i in range(7):
        L = int(rs.randint(20, 90))
        x, lab = rs.randn(L, D0) * 2.0 + 1.0, rs.rand(L) > 0.3
        fs.add("p%d" % i, x, lab)
        names.append("p%d" % i)
        f, l = orc.post_processing(x, lab, delta=True, double_delta=True, feat_norm="cmvn")
        sessions.append(f[l])
    ss = sk.StatServer(sk.IdMap().set(["m"] * len(names), names), ubm=ubm, engine=eng)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(len(names)))
    s0, s1 = orc.accumulate_stat(gmm, sessions)
    return np.abs(ss.stat0 - s0).max() / np.abs(s0).max(), np.abs(ss.stat1 - s1).max() / np.abs(s1).max()


def _scoring_case(sk, eng):
    """adapt_mean_map_multisession + gmm_scoring of the drop-in against the reference's golden scores."""
    g = np.load(os.path.join(GOLDEN, "gmm_scoring.npz"))
    ubm = sk.Mixture(engine=eng)
    ubm.w, ubm.mu, ubm.invcov = g["w"].copy(), g["mu"].copy(), g["invcov"].copy()
    ubm.cov_var_ctl = 1.0 / ubm.invcov
    ubm._compute_all()
    off = np.concatenate([[0], np.cumsum(g["enr_lens"])])
    names = ["e%03d" % i for i in range(len(g["enr_lens"]))]
    feats = {n: g["enr_frames"][off[i]:off[i + 1]] for i, n in enumerate(names)}
    toff = np.concatenate([[0], np.cumsum(g["tst_lens"])])
    feats.update({str(n): g["tst_frames"][toff[i]:toff[i + 1]] for i, n in enumerate(g["ndx_segset"])})
    fs = sk.FeaturesServer(features=feats)
    ss = sk.StatServer(sk.IdMap().set([str(x) for x in g["enr_spk"]], names), ubm=ubm, engine=eng)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(len(names)))
    enroll = ss.adapt_mean_map_multisession(ubm, r=16)
    assert list(enroll.modelset) == [str(x) for x in g["enroll_modelset"]]
    e_map = np.abs(enroll.stat1 - g["enroll_stat1"]).max() / np.abs(g["enroll_stat1"]).max()
    ndx = sk.Ndx()
    ndx.modelset = np.array([str(x) for x in g["ndx_modelset"]])
    ndx.segset = np.array([str(x) for x in g["ndx_segset"]])
    ndx.trialmask = g["trialmask"].astype(bool)
    sc = sk.gmm_scoring(ubm, enroll, ndx, fs, num_thread=1)
    assert sc.validate() and (sc.scoremask == ndx.trialmask).all()
    assert (sc.scoremat[~ndx.trialmask] == 0).all()
    return np.abs(sc.scoremat - g["scores"]).max(), e_map


def test_gmm_scoring_host_logic_single_process():
    """Trial selection, masking and MAP adaptation of the drop-in (oracle-backed engine, no GPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import sidekit_amd as sk
    from fake_engine import OracleEngine
    e_sc, e_map = _scoring_case(sk, OracleEngine())
    # frames pass through float32 on upload and the fake log-likelihood is rounded to float32
    assert e_map < 1e-6 and e_sc < 1e-4, (e_sc, e_map)


@pytest.mark.timeout(300)
def test_world2_gloo_matches_single_process(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in range(world):
        r = res[rank]
        # float32 upload of the frames bounds the EM figures; everything after is float64
        assert r["em_llk"] < 1e-5 and r["em_mu"] < 1e-5 and r["em_w"] < 1e-5, r
        assert r["emu_llk"] < 1e-5, r
        assert r["emv_llk"] < 1e-5 and r["emv_mu"] < 1e-5 and r["emv_flag"] is False, r
        assert r["bw0"] < 1e-5 and r["bw1"] < 1e-5, r
        # Baum-Welch rows: one all-gather of the rank's OWN rows (padded to the longest shard), nothing reduced
        assert r["bw_payload"]["all_reduce"] == 0 and r["bw_payload"]["reduce_scatter"] == 0, r["bw_payload"]
        assert r["bw_payload"]["all_gather"] == r["bw_payload_expected"], (r["bw_payload"], r["bw_payload_expected"])
        assert r["tv_F"] < 1e-5 and r["iv"] < 1e-5 and r["iv_sigma"] < 1e-5 and r["iv_mem"] < 1e-5, r
        assert r["iv_payload"]["all_reduce"] == 0 and r["iv_payload"]["all_gather"] == r["iv_payload_expected"], r["iv_payload"]
        assert r["sc"] < 1e-4 and r["map"] < 1e-6, r
        assert r["pp0"] < 1e-6 and r["pp1"] < 1e-6, r      # frames pass through float32 on their way to the statistics
    # shard-aware loading: 453 + 506 + 559 = 1518 frames when every frame is kept -> rank 0 rows [0, 759) touch
    # files f0, f1 and rank 1 rows [759, 1518) files f1, f2; no rank stacks all three
    for rank in range(world):
        assert all(len(l) < 3 for l in res[rank]["emv_loaded"]), res[rank]["emv_loaded"]
    # both ranks hold the same model
    assert abs(res[0]["em_mu"] - res[1]["em_mu"]) < 1e-12


def test_library_exports_every_declared_symbol():
    """No GPU needed: the C-ABI library loads and exports everything include/skm.h declares."""
    import re
    sys.path.insert(0, ROOT)
    import sidekit_amd as sk
    lib = sk.load_library()
    hdr = open(os.path.join(ROOT, "include", "skm.h")).read()
    names = set(re.findall(r"\b(skm_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 18
    for n in sorted(names):
        assert hasattr(lib, n), "libskm.so does not export %s" % n
    assert lib.skm_version() >= 100
    assert lib.skm_gmm_packed_bytes(512, 60) == (512 // 32) * (15 * 256 + 32) * 4
    assert set(sk.package.engine.SIGNATURES) == names


def test_engine_refuses_without_gpu():
    sys.path.insert(0, ROOT)
    import sidekit_amd as sk
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sk.SkmError):
        sk.HipEngine()
    with pytest.raises(sk.SkmError):
        sk.Mixture().EM_uniform(np.zeros((100, 4)), 2)


@pytest.mark.timeout(180)
def test_bench_launcher_hands_out_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) must itself start two fresh rank processes and relay
    rank 0's JSON line: --launch-check stops each rank after the rendezvous (gloo, no GPU touched)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    got = json.loads(lines[0])
    assert got["n_gpus"] == 2
    assert sorted((r["rank"], r["local_rank"], r["world"]) for r in got["ranks"]) == [(0, 0, 2), (1, 1, 2)]
    assert len({r["pid"] for r in got["ranks"]}) == 2 and os.getpid() not in {r["pid"] for r in got["ranks"]}
