"""Cosine, Mahalanobis and two-covariance scoring of i-vectors with the surface of sidekit/iv_scoring.py:61-205 (the step
after extract_ivectors in i-vector.py:189).  Optional WCCN rotation, length normalisation of both sets and the
score matrix enroll . test' run on the GPU: skm_gemm (float64 MFMA) and skm_row_normalize.  The trial mask does
not restrict the product (the reference scores every pair as well, iv_scoring.py:99)."""
import copy
import logging

import numpy as np

from .idmap import Ndx, Scores
from .statserver import StatServer


def _check_missing_model(enroll, test, ndx):
    """iv_scoring.py:47-58."""
    clean_ndx = ndx.filter(enroll.modelset, test.segset, True)
    enroll.align_models(clean_ndx.modelset)
    test.align_segments(clean_ndx.segset)
    return clean_ndx


def cosine_scoring(enroll, test, ndx, wccn=None, check_missing=True):
    assert isinstance(enroll, StatServer), 'First parameter should be a StatServer'
    assert isinstance(test, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be an Ndx'
    # each side is copied and aligned on its own (iv_scoring.py:84-93), also when the same StatServer is passed
    # twice: models follow modelset order, segments segset order
    enroll_copy = copy.deepcopy(enroll)
    test_copy = copy.deepcopy(test)
    clean_ndx = _check_missing_model(enroll_copy, test_copy, ndx) if check_missing else ndx

    eng = enroll.engine
    torch = eng.torch

    def prepared(ss):
        x = torch.from_numpy(np.ascontiguousarray(ss.stat1, dtype=np.float64)).to(eng.device)
        if wccn is not None:
            x = eng.gemm(x, torch.from_numpy(np.ascontiguousarray(wccn, dtype=np.float64)).to(eng.device))
        return eng.row_normalize_(x)

    e = prepared(enroll_copy)
    # one prepared matrix serves both sides only when the two aligned copies hold the same rows in the same order
    same = enroll is test and enroll_copy.stat1.shape == test_copy.stat1.shape \
        and np.array_equal(enroll_copy.stat1, test_copy.stat1)
    t = e if same else prepared(test_copy)
    s = eng.to_host(eng.gemm(e, t, transB=True))

    score = Scores()
    score.scoremat = s
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    return score


def _unique_models(enroll):
    """iv_scoring.py:127-130 / :173-176."""
    if not np.unique(enroll.modelset).shape == enroll.modelset.shape:
        logging.warning("Enrollment models are not unique, average i-vectors")
        enroll = enroll.mean_stat_per_model()
    return enroll


def _pair_scores(eng, E, T, G, Q, alpha, ca, cb):
    """alpha (e'G t + t'G e) + ca e'Q e + cb t'Q t for every (enrolment, test) pair (Q = None: Q = G): two R x R
    products per set, two cross-term GEMMs accumulated into the score matrix, two row-wise quadratic forms."""
    G = eng._f64_dev(G)
    EG, TG = eng.gemm(E, G), eng.gemm(T, G)
    if Q is None:
        qe, qt = eng.row_dot(EG, E), eng.row_dot(TG, T)
    else:
        Q = eng._f64_dev(Q)
        qe, qt = eng.row_dot(eng.gemm(E, Q), E), eng.row_dot(eng.gemm(T, Q), T)
    S = eng.gemm(EG, T, transB=True)
    S = eng.gemm(E, TG, C=S, transB=True, beta=1.0)
    return eng.pair_combine_(S, qe, qt, alpha=alpha, ca=ca, cb=cb)


def _finish(s, clean_ndx):
    score = Scores()
    score.scoremat = s
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    return score


def mahalanobis_scoring(enroll, test, ndx, m, check_missing=True):
    """-0.5 (e - t)' m (e - t) for all pairs (iv_scoring.py:109-149); like the reference, aligns `enroll` / `test` in
    place.  (e-t)'m(e-t) = e'me + t'mt - e'mt - t'me: the cross terms are GEMMs."""
    assert isinstance(enroll, StatServer), 'First parameter should be a StatServer'
    assert isinstance(test, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be an Ndx'
    assert enroll.stat1.shape[1] == test.stat1.shape[1], 'I-vectors dimension mismatch'
    assert enroll.stat1.shape[1] == m.shape[0], 'I-vectors and Mahalanobis matrix dimension mismatch'
    enroll = _unique_models(enroll)
    clean_ndx = _check_missing_model(enroll, test, ndx) if check_missing else ndx
    eng = enroll.engine
    s = _pair_scores(eng, eng._f64_dev(enroll.stat1), eng._f64_dev(test.stat1), m, None, 0.5, -0.5, -0.5)
    return _finish(eng.to_host(s), clean_ndx)


def two_covariance_scoring(enroll, test, ndx, W, B, check_missing=True):
    """(e+t)'G(e+t) - t'Ht - e'He for all pairs (iv_scoring.py:152-205); G and H from three R x R host inverses as in
    the reference, the pair terms on the device: e'Gt + t'Ge + e'(G-H)e + t'(G-H)t."""
    import scipy.linalg
    assert isinstance(enroll, StatServer), 'First parameter should be a directory'
    assert isinstance(test, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be an Ndx'
    assert enroll.stat1.shape[1] == test.stat1.shape[1], 'I-vectors dimension mismatch'
    assert enroll.stat1.shape[1] == W.shape[0], 'I-vectors and co-variance matrix dimension mismatch'
    assert enroll.stat1.shape[1] == B.shape[0], 'I-vectors and co-variance matrix dimension mismatch'
    enroll = _unique_models(enroll)
    clean_ndx = _check_missing_model(enroll, test, ndx) if check_missing else ndx
    iW = scipy.linalg.inv(W)
    iB = scipy.linalg.inv(B)
    G = iW.dot(scipy.linalg.inv(iB + 2 * iW)).dot(iW)
    H = iW.dot(scipy.linalg.inv(iB + iW)).dot(iW)
    eng = enroll.engine
    s = _pair_scores(eng, eng._f64_dev(enroll.stat1), eng._f64_dev(test.stat1), G, G - H, 1.0, 1.0, 1.0)
    return _finish(eng.to_host(s), clean_ndx)


def fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma, test_uncertainty=None, Vtrans=None, p_known=0.0,
                      scaling_factor=1., check_missing=True):
    """Simplified-PLDA log-likelihood ratios for all pairs (iv_scoring.py:364-471): 0.5 e'Phi e + 0.5 t'Phi t + e'Psi t +
    constant.  Phi, Psi and the constant come from R x R host inverses as in the reference; the pair terms are a device
    GEMM, two quadratic forms and skm_pair_combine.  The open-set correction (p_known != 0) is an element-wise pass
    over the score matrix once it is on the host (the product this function returns).  `test_uncertainty` / `Vtrans`
    are accepted and unused, as in the reference."""
    import scipy.linalg
    enroll_ctr = _unique_models(copy.deepcopy(enroll))
    test_ctr = copy.deepcopy(test)
    clean_ndx = _check_missing_model(enroll_ctr, test_ctr, ndx) if check_missing else ndx

    invSigma = scipy.linalg.inv(Sigma)
    I_spk = np.eye(F.shape[1], dtype='float')
    K = F.T.dot(invSigma * scaling_factor).dot(F)
    K1 = scipy.linalg.inv(K + I_spk)
    K2 = scipy.linalg.inv(2 * K + I_spk)
    plda_cst = np.linalg.slogdet(K2)[1] / 2.0 - np.linalg.slogdet(K1)[1]
    Sigma_ac = np.dot(F, F.T)
    Sigma_tot = Sigma_ac + Sigma
    Sigma_tot_inv = scipy.linalg.inv(Sigma_tot)
    Tmp = np.linalg.inv(Sigma_tot - Sigma_ac.dot(Sigma_tot_inv).dot(Sigma_ac))
    Phi = Sigma_tot_inv - Tmp
    Psi = Sigma_tot_inv.dot(Sigma_ac).dot(Tmp)

    eng = enroll.engine
    mu_dev = eng._f64_dev(np.asarray(mu, dtype=np.float64).reshape(1, -1))
    E = eng.rows_center(eng._f64_dev(enroll_ctr.stat1), Mu=mu_dev)          # center_stat1 on i-vectors (stat0 = 1)
    T = eng.rows_center(eng._f64_dev(test_ctr.stat1), Mu=mu_dev)
    Phi_dev = eng._f64_dev(Phi)
    model_part = eng.row_dot(eng.gemm(E, Phi_dev), E)
    seg_part = eng.row_dot(eng.gemm(T, Phi_dev), T)
    model_part += 2.0 * plda_cst                                            # 0.5 * (e'Phi e + 2 cst)
    s = eng.gemm(eng.gemm(E, eng._f64_dev(Psi)), T, transB=True)
    s = eng.to_host(eng.pair_combine_(s, model_part, seg_part, alpha=scaling_factor, ca=0.5 * scaling_factor,
                                      cb=0.5 * scaling_factor))
    if p_known != 0:
        N = s.shape[0]
        ex = np.exp(s)
        s = s - np.log(p_known * (ex.sum(axis=0)[None, :] - ex) / (N - 1) + (1 - p_known))
    return _finish(s, clean_ndx)


def full_PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, p_known=0.0, scaling_factor=1., check_missing=True):
    """PLDA log-likelihood ratios with a within-class subspace G (iv_scoring.py:266-361): i-vectors projected by
    B = F'S(I - G A G'S), S = scaling * inv(Sigma), then 0.5 (e+t)'K2(e+t) - 0.5 t'K1 t - 0.5 e'K1 e + constant for every
    pair.  The R x R / rank x rank inverses are host LAPACK as in the reference; the projections are two device GEMMs
    and the pair terms e'K2 t + 0.5 e'(K2-K1)e + 0.5 t'(K2-K1)t the same primitives as the other scorers.  Like the
    reference, repeated enrolment models are NOT averaged here (:285-288 are commented out there)."""
    import scipy.linalg
    enroll_copy = copy.deepcopy(enroll)
    test_copy = copy.deepcopy(test)
    clean_ndx = _check_missing_model(enroll_copy, test_copy, ndx) if check_missing else ndx
    invSigma = scipy.linalg.inv(Sigma)
    I_iv = np.eye(mu.shape[0], dtype='float')
    I_ch = np.eye(G.shape[1], dtype='float')
    I_spk = np.eye(F.shape[1], dtype='float')
    A = np.linalg.inv(G.T.dot(invSigma * scaling_factor).dot(G) + I_ch)
    B = F.T.dot(invSigma * scaling_factor).dot(I_iv - G.dot(A).dot(G.T).dot(invSigma * scaling_factor))
    K = B.dot(F)
    K1 = scipy.linalg.inv(K + I_spk)
    K2 = scipy.linalg.inv(2 * K + I_spk)
    constant = np.linalg.slogdet(K2)[1] / 2.0 - np.linalg.slogdet(K1)[1]

    eng = enroll.engine
    mu_dev = eng._f64_dev(np.asarray(mu, dtype=np.float64).reshape(1, -1))
    Bt = eng._f64_dev(np.ascontiguousarray(B.T))
    Ep = eng.gemm(eng.rows_center(eng._f64_dev(enroll_copy.stat1), Mu=mu_dev), Bt)      # rows = B (e - mu)
    Tp = eng.gemm(eng.rows_center(eng._f64_dev(test_copy.stat1), Mu=mu_dev), Bt)
    s = eng.to_host(_pair_scores(eng, Ep, Tp, K2, K2 - K1, 0.5, 0.5, 0.5))
    s = (s + constant) * scaling_factor
    if p_known != 0:
        N = s.shape[0]
        ex = np.exp(s)
        s = s - np.log(p_known * (ex.sum(axis=0)[None, :] - ex) / (N - 1) + (1 - p_known))
    return _finish(s, clean_ndx)


def PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, test_uncertainty=None, Vtrans=None, p_known=0.0,
                 scaling_factor=1., full_model=False):
    """iv_scoring.py:208-263."""
    assert isinstance(enroll, StatServer), 'First parameter should be a StatServer'
    assert isinstance(test, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be an Ndx'
    assert enroll.stat1.shape[1] == test.stat1.shape[1], 'I-vectors dimension mismatch'
    assert enroll.stat1.shape[1] == F.shape[0], 'I-vectors and co-variance matrix dimension mismatch'
    assert enroll.stat1.shape[1] == G.shape[0], 'I-vectors and co-variance matrix dimension mismatch'
    if full_model:
        return full_PLDA_scoring(enroll, test, ndx, mu, F, G, Sigma, p_known=p_known, scaling_factor=scaling_factor)
    return fast_PLDA_scoring(enroll, test, ndx, mu, F, Sigma, test_uncertainty, Vtrans, p_known=p_known,
                             scaling_factor=scaling_factor, check_missing=True)
