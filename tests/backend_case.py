"""Shared driver of the i-vector back-end parity case (tests/golden/iv_backend.npz, made from the real reference by
tools/gen_golden.py gen_backend): runs the drop-in StatServer / iv_scoring methods on a given engine and compares
with the reference's outputs.  Used with the oracle-backed engine on CPU and with the HIP engine on the GPU."""
import copy
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(a, b, rtol):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert err <= rtol, "normwise-max rel err %.3e > %.1e" % (err, rtol)


def _server(sk, eng, modelset, segset, stat1):
    ss = sk.StatServer(engine=eng)
    ss.modelset = np.array(modelset)
    ss.segset = np.array(segset)
    ss.start = np.empty(len(segset), "|O")
    ss.stop = np.empty(len(segset), "|O")
    ss.stat0 = np.ones((len(segset), 1))
    ss.stat1 = np.array(stat1, dtype=np.float64)
    return ss


def run_backend_case(sk, eng, rtol=1e-10):
    g = np.load(os.path.join(GOLD, "iv_backend.npz"), allow_pickle=False)
    dev = _server(sk, eng, g["modelset"], g["segset"], g["stat1"])
    _close(dev.get_total_covariance_stat1(), g["total_cov"], rtol)
    _close(dev.get_within_covariance_stat1(), g["within_cov"], rtol)
    _close(dev.get_between_covariance_stat1(), g["between_cov"], rtol)
    _close(dev.get_mahalanobis_matrix_stat1(), g["mahalanobis_matrix"], 10 * rtol)
    _close(dev.get_wccn_choleski_stat1(), g["wccn_chol"], 10 * rtol)
    lda = dev.get_lda_matrix_stat1(int(g["lda_rank"]))
    sgn = np.sign((lda * g["lda"]).sum(axis=0))                 # an eigenvector's sign is LAPACK's choice
    assert np.all(sgn != 0)
    _close(lda * sgn, g["lda"], 1e-7)
    mps = dev.mean_stat_per_model()
    assert list(mps.modelset) == list(g["mean_per_model_modelset"]) and list(mps.segset) == list(mps.modelset)
    _close(mps.stat0, g["mean_per_model_stat0"], rtol)
    _close(mps.stat1, g["mean_per_model_stat1"], rtol)
    np.testing.assert_array_equal(dev.stat1, g["stat1"])        # none of the getters touches the statistics
    for name, args in (("whitened_full", (g["mu"], g["total_cov"])),
                       ("whitened_presqrt", (g["mu"], g["wccn_chol"], True))):
        w = copy.deepcopy(dev)
        w.whiten_stat1(*args)
        _close(w.stat1, g[name], 100 * rtol)
    w = copy.deepcopy(dev)
    w.whiten_cholesky_stat1(g["mu"], g["total_cov"])
    _close(w.stat1, g["whitened_chol"], 100 * rtol)

    def trial_sets():
        enroll = _server(sk, eng, g["enroll_modelset"], g["enroll_segset"], g["enroll_iv"])
        test = _server(sk, eng, g["test_segset"], g["test_segset"], g["test_iv"])
        ndx = sk.Ndx()
        ndx.modelset, ndx.segset, ndx.trialmask = g["ndx_modelset"], g["ndx_segset"], g["trialmask"]
        return enroll, test, ndx

    enroll, test, ndx = trial_sets()
    sc = sk.mahalanobis_scoring(enroll, test, ndx, g["mahalanobis_matrix"])
    assert list(sc.modelset) == list(g["maha_modelset"]) and list(sc.segset) == list(g["maha_segset"])
    np.testing.assert_array_equal(sc.scoremask, g["maha_scoremask"])
    _close(sc.scoremat, g["maha_scoremat"], rtol)
    enroll, test, ndx = trial_sets()
    _close(sk.mahalanobis_scoring(enroll, test, ndx, g["maha_nonsym_matrix"]).scoremat, g["maha_nonsym_scoremat"], rtol)
    enroll, test, ndx = trial_sets()
    sc = sk.two_covariance_scoring(enroll, test, ndx, g["within_cov"], g["between_cov"])
    assert list(sc.modelset) == list(g["twocov_modelset"]) and list(sc.segset) == list(g["twocov_segset"])
    _close(sc.scoremat, g["twocov_scoremat"], 10 * rtol)
    for tag, kw in (("closed", dict()), ("open", dict(p_known=0.3)), ("scaled", dict(p_known=0.1, scaling_factor=0.8))):
        enroll, test, ndx = trial_sets()
        e0, t0 = enroll.stat1.copy(), test.stat1.copy()
        sc = sk.fast_PLDA_scoring(enroll, test, ndx, g["plda_mu"], g["plda_F"], g["plda_Sigma"], **kw)
        assert list(sc.modelset) == list(g["plda_modelset"]) and list(sc.segset) == list(g["plda_segset"])
        _close(sc.scoremat, g["plda_%s_scoremat" % tag], 10 * rtol)
        np.testing.assert_array_equal(enroll.stat1, e0)           # works on copies, like the reference
        np.testing.assert_array_equal(test.stat1, t0)
    enroll, test, ndx = trial_sets()
    sc = sk.PLDA_scoring(enroll, test, ndx, g["plda_mu"], g["plda_F"], np.zeros((int(g["R"]), 1)), g["plda_Sigma"])
    _close(sc.scoremat, g["plda_closed_scoremat"], 10 * rtol)


def run_full_plda_case(sk, eng, rtol=1e-9):
    """full_PLDA_scoring (iv_scoring.py:266-361) against the real reference (tests/golden/full_plda.npz, tools/gen_golden.py
    gen_full_plda): closed set, open set, a scaling factor, the PLDA_scoring(full_model=True) dispatch."""
    g = np.load(os.path.join(GOLD, "full_plda.npz"), allow_pickle=False)

    def trial_sets():
        enroll = _server(sk, eng, g["enroll_modelset"], g["enroll_segset"], g["enroll_iv"])
        test = _server(sk, eng, g["test_segset"], g["test_segset"], g["test_iv"])
        ndx = sk.Ndx()
        ndx.modelset, ndx.segset, ndx.trialmask = g["ndx_modelset"], g["ndx_segset"], g["trialmask"]
        return enroll, test, ndx
    for tag, kw in (("closed", dict()), ("open", dict(p_known=0.4)), ("scaled", dict(p_known=0.2, scaling_factor=0.7))):
        enroll, test, ndx = trial_sets()
        e0, t0 = enroll.stat1.copy(), test.stat1.copy()
        sc = sk.full_PLDA_scoring(enroll, test, ndx, g["mu"], g["F"], g["G"], g["Sigma"], **kw)
        assert list(sc.modelset) == list(g["modelset"]) and list(sc.segset) == list(g["segset"])
        np.testing.assert_array_equal(sc.scoremask, g["scoremask"])
        _close(sc.scoremat, g["full_%s_scoremat" % tag], rtol)
        np.testing.assert_array_equal(enroll.stat1, e0)           # works on copies, like the reference
        np.testing.assert_array_equal(test.stat1, t0)
    enroll, test, ndx = trial_sets()
    sc = sk.PLDA_scoring(enroll, test, ndx, g["mu"], g["F"], g["G"], g["Sigma"], p_known=0.4, full_model=True)
    _close(sc.scoremat, g["full_open_scoremat"], rtol)
