"""GPU parity of the i-vector back-end: skm_class_sums / skm_rows_center / skm_row_dot /
skm_pair_combine + float64 MFMA GEMMs through the C ABI, against the real reference's outputs (golden) and the oracle
at a larger, ragged size."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import sidekit_ref as orc  # noqa: E402
from backend_case import run_backend_case, run_full_plda_case, _server, _close  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sk():
    import sidekit_amd
    return sidekit_amd


@pytest.fixture(scope="module")
def eng(sk):
    return sk.default_engine()


def test_backend_vs_reference_golden(sk, eng):
    run_backend_case(sk, eng)


def test_full_plda_scoring_vs_reference_golden(sk, eng):
    run_full_plda_case(sk, eng)


def test_plda_training_vs_reference_golden(sk, eng, tmp_path):
    """FactorAnalyser.plda (factor_analyser.py:810-910) on the device primitives against the real reference."""
    from plda_case import run_plda_case
    run_plda_case(sk, eng, tmp_path)


def test_plda_training_larger_vs_oracle(sk, eng, tmp_path):
    """4000 i-vectors of rank 200 from 600 speakers (1..14 sessions, shuffled), subspace rank 60, two iterations."""
    rs = np.random.RandomState(12)
    R, K, rank_f = 200, 600, 60
    V = rs.randn(R, rank_f)
    nsess = rs.randint(1, 15, size=K)
    models = np.repeat(np.array(["s%04d" % k for k in range(K)]), nsess)
    X = np.repeat(rs.randn(K, rank_f).dot(V.T), nsess, axis=0) + rs.randn(models.shape[0], R) * 0.8 + 0.1
    perm = rs.permutation(models.shape[0])
    models, X = models[perm], X[perm]
    ss = _server(sk, eng, models, ["g%05d" % i for i in range(models.shape[0])], X)
    fa = sk.FactorAnalyser(engine=eng)
    fa.plda(ss, rank_f, nb_iter=2, output_file_name=str(tmp_path / "plda_big"))
    m, F, S = orc.plda_train(models, X, rank_f, nb_iter=2)
    _close(fa.mean, m, 1e-12)
    _close(fa.Sigma, S, 1e-8)
    _close(fa.F.dot(fa.F.T), F.dot(F.T), 1e-8)          # the subspace, free of LAPACK's eigenvector signs
    sgn = np.sign((fa.F * F).sum(axis=0))
    _close(fa.F * sgn, F, 1e-7)


def test_backend_reductions_ragged_vs_oracle(sk, eng):
    """R = 83 (not a multiple of 64), 700 sessions of 37 speakers with 1..60 sessions each in shuffled order."""
    rs = np.random.RandomState(5)
    R, K = 83, 37
    counts = rs.randint(1, 61, size=K)
    counts[:3] = [1, 2, 60]
    models = np.repeat(["s%03d" % k for k in range(K)], counts)
    x = (rs.randn(K, R) * 2.0)[np.repeat(np.arange(K), counts)] + rs.randn(counts.sum(), R) + 3.0
    p = rs.permutation(models.size)
    models, x = models[p], x[p]
    ss = _server(sk, eng, models, ["g%d" % i for i in range(models.size)], x)
    _close(ss.get_total_covariance_stat1(), orc.total_covariance(x), 1e-11)
    _close(ss.get_within_covariance_stat1(), orc.within_covariance(models, x), 1e-11)
    _close(ss.get_between_covariance_stat1(), orc.between_covariance(models, x), 1e-11)
    _close(ss.get_wccn_choleski_stat1(), orc.wccn_choleski(models, x), 1e-9)
    uniq, means = orc.mean_per_model(models, x)
    mps = ss.mean_stat_per_model()
    assert list(mps.modelset) == list(uniq)
    _close(mps.stat1, means, 1e-13)
    X = eng._f64_dev(x)
    _close(eng.to_host(eng.col_mean(X))[0], x.mean(axis=0), 1e-13)
    # pair scores at a shape with edge tiles on both sides
    E, T = rs.randn(130, R), rs.randn(517, R)
    M = orc.within_covariance(models, x)
    e = _server(sk, eng, ["m%d" % i for i in range(130)], ["m%d" % i for i in range(130)], E)
    t = _server(sk, eng, ["t%d" % i for i in range(517)], ["t%d" % i for i in range(517)], T)
    ndx = sk.Ndx()
    ndx.modelset, ndx.segset = e.modelset.copy(), t.segset.copy()
    ndx.trialmask = np.ones((130, 517), dtype=bool)
    _close(sk.mahalanobis_scoring(e, t, ndx, M).scoremat, orc.mahalanobis_scores(E, T, M), 1e-11)
    B = orc.between_covariance(models, x)
    _close(sk.two_covariance_scoring(e, t, ndx, M, B).scoremat, orc.two_covariance_scores(E, T, M, B), 1e-10)


def test_gram_split_over_rows_with_remainder(eng):
    """S = 30 011 rows, R = 83: 4 output tiles -> the rows are cut into 29 slices of 1034 plus a 25-row remainder."""
    rs = np.random.RandomState(8)
    x = rs.randn(30011, 83) + 0.5
    got = eng.to_host(eng.gram(eng._f64_dev(x), alpha=0.25))
    _close(got, 0.25 * x.T.dot(x), 1e-12)
    again = eng.to_host(eng.gram(eng._f64_dev(x), alpha=0.25))
    np.testing.assert_array_equal(got, again)                   # fixed summation order


def test_backend_full_size_properties(sk, eng):
    """BASELINE configs[4] size (50 000 i-vectors of rank 400, 5 000 speakers) through size-independent properties:
    total = within + between covariance (law of total variance), WCCN factor W satisfies W W' = inverse of the mean
    per-speaker scatter, Mahalanobis scores with the identity are -0.5 squared distances and vanish on the diagonal
    of a set scored against itself, two-covariance scores of a set against itself are symmetric."""
    rs = np.random.RandomState(11)
    S, R, K = 50_000, 400, 5_000
    spk = rs.randint(0, K, size=S)
    spk[:K] = np.arange(K)                                        # every speaker has at least one session
    x = rs.randn(K, R)[spk] * 1.5 + rs.randn(S, R) + 0.25
    ss = _server(sk, eng, np.array(["s%04d" % k for k in spk]), np.array(["g%d" % i for i in range(S)]), x)
    tot, wit, bet = ss.get_total_covariance_stat1(), ss.get_within_covariance_stat1(), ss.get_between_covariance_stat1()
    _close(wit + bet, tot, 1e-11)
    assert np.abs(tot - tot.T).max() <= 1e-13 * np.abs(tot).max()
    W = ss.get_wccn_choleski_stat1()
    X, means, Xc, counts = ss._class_centred(True)
    scatter = eng.to_host(eng.gram(Xc, alpha=1.0 / K))
    _close(W.dot(W.T).dot(scatter), np.eye(R), 1e-8)
    n = 3000
    sub = _server(sk, eng, ["m%d" % i for i in range(n)], ["m%d" % i for i in range(n)], x[:n])
    ndx = sk.Ndx()
    ndx.modelset, ndx.segset = sub.modelset.copy(), sub.segset.copy()
    ndx.trialmask = np.ones((n, n), dtype=bool)
    d = sk.mahalanobis_scoring(sub, sub, ndx, np.eye(R)).scoremat
    sq = (x[:n] ** 2).sum(axis=1)
    ref = -0.5 * (sq[:, None] + sq[None, :] - 2.0 * x[:n].dot(x[:n].T))
    assert np.abs(d - ref).max() <= 1e-11 * np.abs(ref).max()
    assert np.abs(np.diag(d)).max() <= 1e-10 * np.abs(ref).max()
    tc = sk.two_covariance_scoring(sub, sub, ndx, wit, bet).scoremat
    assert np.abs(tc - tc.T).max() <= 1e-10 * np.abs(tc).max()
