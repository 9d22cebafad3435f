"""The CPU oracle (oracle/sidekit_ref.py) against golden vectors produced by the real
reference (tools/gen_golden.py).  This is what pins the oracle; tolerance 1e-10 relative
(float64 vs float64, only summation order may differ)."""
import numpy as np
import pytest

from oracle import sidekit_ref as orc
from conftest import golden

RT = 1e-10


def close(a, b, rtol=RT, atol=0.0):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / scale
    assert err <= rtol + atol, "normwise-max rel err %.3e" % err


@pytest.mark.parametrize("tag", ["c8_d6", "c64_d60", "c40_d13"])
def test_estep_mstep_split(tag):
    g = golden("estep_%s.npz" % tag)
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    close(gmm.cst, g["cst"])
    close(gmm.det, g["det"])
    close(gmm.A, g["A"])
    lp = orc.log_posterior(gmm, g["cep"])
    pp, log_lk = orc.logsumexp_rows(lp)
    close(log_lk, g["log_lk"])
    if "lp" in g.files:
        close(lp, g["lp"])
        close(pp, g["pp"])
    N, F, S, llk = orc.expectation(gmm, g["cep"])
    close(N, g["N"])
    close(F, g["F"])
    close(S, g["S"])
    assert abs(llk - float(g["llk"])) <= 1e-10 * abs(float(g["llk"]))
    orc.maximization(gmm, N, F, S)
    close(gmm.w, g["m_w"])
    close(gmm.mu, g["m_mu"])
    close(gmm.invcov, g["m_invcov"])
    close(gmm.A, g["m_A"])
    orc.split(gmm)
    close(gmm.w, g["s_w"])
    close(gmm.mu, g["s_mu"])
    close(gmm.invcov, g["s_invcov"])
    close(gmm.cov_var_ctl, g["s_ctl"])
    close(gmm.A, g["s_A"])


def test_em_split_full_schedule():
    g = golden("em_split.npz")
    chunks = [g["chunk%d" % i] for i in range(4)]
    init = orc.init_single(np.vstack(chunks))
    close(init.w, g["i_w"])
    close(init.mu, g["i_mu"])
    close(init.invcov, g["i_invcov"])
    close(init.cov_var_ctl, g["i_ctl"])
    gmm, llk = orc.em_split(chunks, 8)
    assert len(llk) == len(g["llk"]) == 5
    close(np.array(llk), g["llk"], rtol=1e-9)
    close(gmm.w, g["w"], rtol=1e-8)
    close(gmm.mu, g["mu"], rtol=1e-8)
    close(gmm.invcov, g["invcov"], rtol=1e-8)
    close(gmm.cov_var_ctl, g["ctl"])


def test_em_split_with_vad_labels_and_keep_all_features():
    """_init sees every frame (server setting keep_all_features = True), the E-steps the labelled frames only
    (mixture.py:585); golden from the real reference through its own post_processing."""
    g = golden("em_split_vad.npz")
    every = [g["chunk%d" % i] for i in range(3)]
    speech = [g["chunk%d" % i][g["vad%d" % i]] for i in range(3)]
    assert int(g["keep_all_after"]) == 0
    gmm, llk = orc.em_split(speech, 4, init_chunks=every)
    assert len(llk) == len(g["llk"]) == 3
    close(np.array(llk), g["llk"], rtol=1e-9)
    close(gmm.w, g["w"], rtol=1e-8)
    close(gmm.mu, g["mu"], rtol=1e-8)
    close(gmm.invcov, g["invcov"], rtol=1e-8)
    close(gmm.cov_var_ctl, g["ctl"])
    # the other reading (E-steps on every frame) is NOT what the reference does
    gmm2, llk2 = orc.em_split(every, 4)
    assert np.abs(np.array(llk2) - g["llk"]).max() > 1e-3


def test_tv_multi_block_rank():
    """R = 40 spans two 32-blocks of the device Cholesky; C=64, D=16, 120 sessions, 3 iterations (real reference)."""
    g = golden("tv_big.npz")
    C, D, R, S = [int(v) for v in g["shape"]]
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    tv_init = 0.1 * np.random.RandomState(7).randn(C * D, R)
    F3 = orc.total_variability(g["stat0_f32"], g["stat1_f32"], gmm, R, nb_iter=3, tv_init=tv_init, batch_size=30)
    close(F3, g["F_it3"], rtol=1e-8)
    iv, iv_sigma = orc.extract_ivectors(g["stat0_f32"], g["stat1_f32"], gmm, g["F_it3"], batch_size=30)
    close(iv, g["iv"], rtol=1e-9)
    close(iv_sigma, g["iv_sigma"], rtol=1e-9)


def test_em_uniform():
    g = golden("em_uniform.npz")
    init = orc.init_uniform(g["cep"], 4)
    close(init.w, g["i_w"])
    close(init.mu, g["i_mu"])
    close(init.invcov, g["i_invcov"])
    close(init.cov_var_ctl, g["i_ctl"])
    gmm, llk = orc.em_uniform(g["cep"], 4)
    assert len(llk) == len(g["llk"])
    close(np.array(llk), g["llk"], rtol=1e-9)
    close(gmm.w, g["w"], rtol=1e-8)
    close(gmm.mu, g["mu"], rtol=1e-8)
    close(gmm.invcov, g["invcov"], rtol=1e-8)


def _sessions(g):
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    return [g["frames"][off[i]:off[i + 1]] for i in range(len(g["lens"]))]


def test_accumulate_stat():
    g = golden("accumulate_stat.npz")
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    stat0, stat1 = orc.accumulate_stat(gmm, _sessions(g))
    close(stat0, g["stat0"])
    close(stat1, g["stat1"])


def test_accumulate_stat_dim_mismatch():
    g = golden("accumulate_stat.npz")
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    with pytest.raises(Exception):
        orc.accumulate_stat(gmm, [np.zeros((5, gmm.D + 1))])


def test_e_on_batch_and_tv():
    g = golden("tv.npz")
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    s0 = g["stat0_f32"][:9].copy()
    s1 = g["stat1_f32"][:9].copy()
    assert s1.dtype == np.float32
    e_h, e_hh = orc.e_on_batch(s0, s1, gmm, g["tv_init"])
    assert s1.dtype == np.float32
    np.testing.assert_array_equal(s1, g["eob_whitened"])      # float32 in-place whitening is bit exact
    close(e_h, g["eob_e_h"])
    close(e_hh, g["eob_e_hh"])
    F1 = orc.total_variability(g["stat0_f32"], g["stat1_f32"], gmm, 6, nb_iter=1, tv_init=g["tv_init"], batch_size=7)
    close(F1, g["F_it1"], rtol=1e-9)
    F3 = orc.total_variability(g["stat0_f32"], g["stat1_f32"], gmm, 6, nb_iter=3, tv_init=g["tv_init"], batch_size=7)
    close(F3, g["F_it3"], rtol=1e-8)
    close(F3, g["F_it3_pool"], rtol=1e-8)
    F2 = orc.total_variability(g["stat0_f32"], g["stat1_f32"], gmm, 6, nb_iter=2, min_div=False,
                               tv_init=g["tv_init"], batch_size=7)
    close(F2, g["F_it2_nomindiv"], rtol=1e-8)
    iv, iv_sigma = orc.extract_ivectors(g["stat0_f32"], g["stat1_f32"], gmm, g["F_it3_pool"], batch_size=7)
    close(iv, g["iv"], rtol=1e-9)
    close(iv_sigma, g["iv_sigma"], rtol=1e-9)
    assert g["iv_stat0"].shape == (iv.shape[0], 1) and np.all(g["iv_stat0"] == 1.0)
    assert np.all(g["fa_mean"] == 0) and np.all(g["fa_sigma"] == 0)


def test_whiten_f64_matches_single_path():
    g = golden("tv.npz")
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    s0 = g["stat0_f32"].astype(np.float64)
    s1 = g["stat1_f32"].astype(np.float64)
    w = orc.whiten_stat1(s0, s1, gmm.mu.flatten(), 1.0 / gmm.invcov.flatten())
    R = g["F_it3_pool"].shape[1]
    Fm = g["F_it3_pool"]
    idx = np.repeat(np.arange(gmm.C), gmm.D)
    iv = np.zeros((s0.shape[0], R))
    for i in range(s0.shape[0]):
        L = np.eye(R) + (Fm.T * s0[i, idx]).dot(Fm)
        iv[i] = np.linalg.solve(L, Fm.T.dot(w[i]))
    close(iv, g["iv_single"], rtol=1e-9)


def test_map_adaptation_and_gmm_scoring_vs_reference_golden():
    """adapt_mean_map_multisession (statserver.py:986-1028) and gmm_scoring_singleThread
    (gmm_scoring.py:56-119) of the real reference."""
    g = golden("gmm_scoring.npz")
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    off = np.concatenate([[0], np.cumsum(g["enr_lens"])])
    sess = [g["enr_frames"][off[i]:off[i + 1]] for i in range(len(g["enr_lens"]))]
    s0, s1 = orc.accumulate_stat(gmm, sess)
    models, M = orc.adapt_mean_map_multisession(gmm, g["enr_spk"], s0, s1, r=16)
    assert list(models) == list(g["enroll_modelset"])
    np.testing.assert_allclose(M, g["enroll_stat1"], rtol=1e-10, atol=1e-12)
    toff = np.concatenate([[0], np.cumsum(g["tst_lens"])])
    segs = {str(n): g["tst_frames"][toff[i]:toff[i + 1]] for i, n in enumerate(g["ndx_segset"])}
    sc = orc.gmm_scoring(gmm, models, M, g["ndx_modelset"], g["ndx_segset"], g["trialmask"], segs)
    np.testing.assert_allclose(sc, g["scores"], rtol=1e-9, atol=1e-11)


def _cosine_case(sk, engine=None):
    """Drop-in cosine_scoring on the golden case of the real reference; returns (scores, scores_wccn, golden)."""
    g = golden("cosine_scoring.npz")

    def server(models, segs, iv):
        ss = sk.StatServer(engine=engine)
        ss.modelset = np.array([str(x) for x in models])
        ss.segset = np.array([str(x) for x in segs])
        ss.start = np.empty(len(segs), "|O")
        ss.stop = np.empty(len(segs), "|O")
        ss.stat0 = np.ones((len(segs), 1))
        ss.stat1 = np.array(iv, dtype="d")
        return ss
    enroll = server(g["enroll_modelset"], g["enroll_modelset"], g["enroll_iv"])
    test = server(g["test_segset"], g["test_segset"], g["test_iv"])
    ndx = sk.Ndx()
    ndx.modelset = np.array([str(x) for x in g["ndx_modelset"]])
    ndx.segset = np.array([str(x) for x in g["ndx_segset"]])
    ndx.trialmask = g["trialmask"].astype(bool)
    before = enroll.stat1.copy()
    sc = sk.cosine_scoring(enroll, test, ndx, wccn=None)
    scw = sk.cosine_scoring(enroll, test, ndx, wccn=g["wccn"])
    assert np.array_equal(enroll.stat1, before)                        # inputs are not modified (deep copies)
    assert list(sc.modelset) == [str(x) for x in g["out_modelset"]]
    assert list(sc.segset) == [str(x) for x in g["out_segset"]]
    assert np.array_equal(sc.scoremask, g["scoremask"]) and sc.validate()
    return sc.scoremat, scw.scoremat, g


def test_cosine_scoring_oracle_and_host_logic_vs_reference_golden():
    """cosine_scoring (iv_scoring.py:61-106).  The oracle's arithmetic and the drop-in's model /
    segment alignment (oracle-backed engine, no GPU) against the real reference."""
    import sidekit_amd as sk
    from fake_engine import OracleEngine
    g = golden("cosine_scoring.npz")
    e_idx = [list(g["enroll_modelset"]).index(m) for m in g["out_modelset"]]
    t_idx = [list(g["test_segset"]).index(s) for s in g["out_segset"]]
    close(orc.cosine_scoring(g["enroll_iv"][e_idx], g["test_iv"][t_idx]), g["scoremat"], rtol=1e-12)
    close(orc.cosine_scoring(g["enroll_iv"][e_idx], g["test_iv"][t_idx], g["wccn"]), g["scoremat_wccn"], rtol=1e-12)
    s, sw, g = _cosine_case(sk, OracleEngine())
    close(s, g["scoremat"], rtol=1e-12)
    close(sw, g["scoremat_wccn"], rtol=1e-12)


def test_iv_backend_oracle_vs_reference_golden():
    """SURVEY 8f rank 3 (remainder): covariance / LDA / WCCN / Mahalanobis matrices, whitening, mahalanobis_scoring and
    two_covariance_scoring of the real reference (tools/gen_golden.py gen_backend) against the oracle's restatement."""
    g = golden("iv_backend.npz")
    ms, x = g["modelset"], g["stat1"]
    close(orc.total_covariance(x), g["total_cov"], rtol=1e-12)
    close(orc.within_covariance(ms, x), g["within_cov"], rtol=1e-12)
    close(orc.between_covariance(ms, x), g["between_cov"], rtol=1e-12)
    close(orc.lda_matrix(ms, x, int(g["lda_rank"])), g["lda"], rtol=1e-9)
    close(scipy_inv(orc.within_covariance(ms, x)), g["mahalanobis_matrix"], rtol=1e-11)
    close(orc.wccn_choleski(ms, x), g["wccn_chol"], rtol=1e-11)
    uniq, means = orc.mean_per_model(ms, x)
    assert list(uniq) == list(g["mean_per_model_modelset"])
    close(means, g["mean_per_model_stat1"], rtol=1e-13)
    close(orc.whiten_full(x, g["mu"], g["total_cov"]), g["whitened_full"], rtol=1e-10)
    close(orc.whiten_cholesky(x, g["mu"], g["total_cov"]), g["whitened_chol"], rtol=1e-10)
    close(orc.whiten_full(x, g["mu"], g["wccn_chol"], True), g["whitened_presqrt"], rtol=1e-12)
    # the scorers average repeated enrolment models, then follow the cleaned Ndx
    uniq, e_mean = orc.mean_per_model(g["enroll_modelset"], g["enroll_iv"])
    e = e_mean[[list(uniq).index(m) for m in g["maha_modelset"]]]
    t = g["test_iv"][[list(g["test_segset"]).index(sg) for sg in g["maha_segset"]]]
    close(orc.mahalanobis_scores(e, t, g["mahalanobis_matrix"]), g["maha_scoremat"], rtol=1e-12)
    close(orc.mahalanobis_scores(e, t, g["maha_nonsym_matrix"]), g["maha_nonsym_scoremat"], rtol=1e-12)
    close(orc.two_covariance_scores(e, t, g["within_cov"], g["between_cov"]), g["twocov_scoremat"], rtol=1e-11)
    for tag, kw in (("closed", dict()), ("open", dict(p_known=0.3)), ("scaled", dict(p_known=0.1, scaling_factor=0.8))):
        close(orc.fast_plda_scores(e, t, g["plda_mu"], g["plda_F"], g["plda_Sigma"], **kw),
              g["plda_%s_scoremat" % tag], rtol=1e-12)


def test_iv_backend_host_logic_vs_reference_golden():
    """The drop-in StatServer / iv_scoring back-end methods (class layout, alignment, averaging of repeated models, the
    R x R LAPACK steps) on the oracle-backed engine, no GPU, against the real reference."""
    import sidekit_amd as sk
    from fake_engine import OracleEngine
    from backend_case import run_backend_case
    run_backend_case(sk, OracleEngine())


def test_full_plda_scoring_oracle_and_host_logic_vs_reference_golden():
    """oracle/sidekit_ref.full_plda_scores on the aligned trial sets, and the drop-in full_PLDA_scoring (Ndx filtering,
    alignment, host inverses) on the oracle-backed engine, against the real reference."""
    import sidekit_amd as sk
    from fake_engine import OracleEngine
    from backend_case import run_full_plda_case
    g = golden("full_plda.npz")
    e = np.array([g["enroll_iv"][list(g["enroll_modelset"]).index(m)] for m in g["modelset"]])
    t = np.array([g["test_iv"][list(g["test_segset"]).index(sg)] for sg in g["segset"]])
    for tag, kw in (("closed", dict()), ("open", dict(p_known=0.4)), ("scaled", dict(p_known=0.2, scaling_factor=0.7))):
        s = orc.full_plda_scores(e, t, g["mu"], g["F"], g["G"], g["Sigma"], **kw)
        assert np.abs(s - g["full_%s_scoremat" % tag]).max() <= 1e-12 * np.abs(g["full_%s_scoremat" % tag]).max()
    run_full_plda_case(sk, OracleEngine())


def test_plda_training_oracle_vs_reference_golden():
    """oracle/sidekit_ref.plda_train against the real reference's FactorAnalyser.plda (3 iterations, 1 iteration, a
    scaling factor)."""
    g = golden("plda.npz")
    for tag, kw in (("it3", dict(nb_iter=3)), ("it1", dict(nb_iter=1)), ("scaled", dict(nb_iter=2, scaling_factor=0.6))):
        m, F, S = orc.plda_train(g["modelset"], g["stat1"], int(g["rank_f"]), **kw)
        for a, b in ((m, g["mean_" + tag]), (F, g["F_" + tag]), (S, g["Sigma_" + tag])):
            assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max()


def test_plda_training_host_logic_vs_reference_golden(tmp_path):
    """The drop-in FactorAnalyser.plda (class sorting by session count, grouped inverses, the sums that replace the
    K x rank x rank array, file products) on the oracle-backed engine, no GPU, against the real reference."""
    import sidekit_amd as sk
    from fake_engine import OracleEngine
    from plda_case import run_plda_case
    run_plda_case(sk, OracleEngine(), tmp_path)


def _sliding_cases():
    g = golden("post_processing_sliding.npz")
    for i in range(int(g["n_cases"])):
        o = g["case%d_opt" % i]
        kw = dict(rasta=bool(o[0]), delta=bool(o[1]), double_delta=bool(o[2]),
                  feat_norm={3: "cms_sliding", 4: "cmvn_sliding"}[int(o[3])], keep_all_features=bool(o[4]))
        yield i, kw, g["case%d_x" % i], g["case%d_label" % i], g["case%d_feat" % i], g["case%d_lab" % i]


def test_sliding_normalisation_oracle_vs_reference_golden():
    """features_server.py:282-287 -> frontend/normfeat.py:194-229 (real reference, tools/gen_golden.py gen_sliding)."""
    for i, kw, x, label, feat, lab in _sliding_cases():
        f, l = orc.post_processing(x.copy(), label.copy(), **kw)
        assert f.shape == feat.shape, i
        np.testing.assert_array_equal(l, lab)
        close(f, feat, rtol=1e-12)


def test_sliding_normalisation_host_logic_vs_reference_golden():
    """FeaturesServer._sliding (un-normalised pass, lazy whole-session pass, compaction, dtypes) on the oracle-backed
    engine, no GPU, against the real reference."""
    import sidekit_amd as sk
    from fake_engine import OracleEngine
    for i, kw, x, label, feat, lab in _sliding_cases():
        fs = sk.FeaturesServer(engine=OracleEngine(), **kw)
        f, l = fs.post_processing(x.copy(), label.copy())
        assert f.dtype == np.float64 and f.shape == feat.shape, i
        np.testing.assert_array_equal(l, lab)
        close(f, feat, rtol=1e-12)


def scipy_inv(a):
    import scipy.linalg
    return scipy.linalg.inv(a)


def _pp_cases():
    g = golden("post_processing.npz")
    for i in range(int(g["n_cases"])):
        o = g["case%d_opt" % i]
        kw = dict(rasta=bool(o[0]), delta=bool(o[1]), double_delta=bool(o[2]), feat_norm=[None, "cms", "cmvn"][o[3]],
                  keep_all_features=bool(o[4]),
                  mask=g["case%d_mask" % i] if ("case%d_mask" % i) in g.files else None,
                  global_mean=g["case%d_gmean" % i] if ("case%d_gmean" % i) in g.files else None,
                  global_std=g["case%d_gstd" % i] if ("case%d_gstd" % i) in g.files else None)
        yield i, g["case%d_x" % i], g["case%d_label" % i], kw, g["case%d_feat" % i], g["case%d_lab" % i]


def test_post_processing_oracle_vs_reference_golden():
    """FeaturesServer.post_processing of the real reference (RASTA, delta, double delta, cms/cmvn, mask, global
    statistics, frame selection, degenerate sessions whose variance is zero -> inf/NaN): the restatement makes the
    same NumPy/SciPy calls, so it is bit-identical wherever the reference's output is finite."""
    n = 0
    for i, x, label, kw, feat, lab in _pp_cases():
        with np.errstate(all="ignore"):
            f, l = orc.post_processing(x, label, **kw)
        assert f.shape == feat.shape and f.dtype == feat.dtype, i
        assert np.array_equal(l, lab), i
        assert np.array_equal(f, feat, equal_nan=True), i
        n += 1
    assert n == 11


def test_post_processing_closed_forms():
    """What the device kernel implements, stated directly: RASTA = 5-tap FIR primed by frames 0..3 + pole 0.94
    started at frame 4; delta = zero-padded 'same' convolution rounded to float32."""
    rs = np.random.RandomState(3)
    x = rs.randn(50, 3)
    y = orc.rasta_filt(x)
    b = np.arange(.2, -.3, -.1)
    ref = np.zeros_like(x)
    for n in range(4, 50):
        ref[n] = sum(b[k] * x[n - k] for k in range(5)) + 0.94 * ref[n - 1]
    assert np.abs(y - ref).max() < 1e-14
    d = orc.compute_delta(x)
    f = orc.DELTA_FILTER
    xp = np.vstack([np.zeros((3, 3)), x, np.zeros((3, 3))])
    ref = np.array([sum(f[k] * xp[t + 6 - k] for k in range(7)) for t in range(50)]).astype(np.float32)
    assert d.dtype == np.float32 and np.array_equal(d, ref)
