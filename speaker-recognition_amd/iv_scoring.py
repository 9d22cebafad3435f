"""Cosine scoring of i-vectors with the surface of sidekit/iv_scoring.py:61-106 (the step
after extract_ivectors in i-vector.py:189).  Optional WCCN rotation, length normalisation of both sets and the
score matrix enroll . test' run on the GPU: skm_gemm (float64 MFMA) and skm_row_normalize.  The trial mask does
not restrict the product (the reference scores every pair as well, iv_scoring.py:99)."""
import copy

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
    same = enroll is test
    enroll_copy = copy.deepcopy(enroll)
    test_copy = enroll_copy if same else copy.deepcopy(test)
    clean_ndx = _check_missing_model(enroll_copy, test_copy, ndx) if check_missing else ndx

    eng = enroll.engine
    torch = eng.torch

    def prepared(ss):
        x = torch.from_numpy(np.ascontiguousarray(ss.stat1, dtype=np.float64)).to(eng.device)
        if wccn is not None:
            x = eng.gemm(x, torch.from_numpy(np.ascontiguousarray(wccn, dtype=np.float64)).to(eng.device))
        return eng.row_normalize_(x)

    e = prepared(enroll_copy)
    t = e if same else prepared(test_copy)
    s = eng.to_host(eng.gemm(e, t, transB=True))

    score = Scores()
    score.scoremat = s
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    return score
