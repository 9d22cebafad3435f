"""Shared driver of the PLDA training parity case (tests/golden/plda.npz, made from the real reference's
FactorAnalyser.plda by tools/gen_golden.py gen_plda): the drop-in FactorAnalyser.plda on a given engine against the
reference's mean / F / Sigma.  Used with the oracle-backed engine on CPU and with the HIP engine on the GPU."""
import os

import numpy as np

from backend_case import _server, _close, GOLD


def _close_up_to_signs(F, ref, rtol):
    """The columns of F start as eigenvectors of the total covariance, whose signs are LAPACK's choice (a last-bit
    difference in the covariance can flip one) and survive the EM: the model is F F'."""
    _close(F.dot(F.T), ref.dot(ref.T), rtol)
    sgn = np.sign((F * ref).sum(axis=0))
    assert np.all(sgn != 0)
    _close(F * sgn, ref, rtol)


def run_plda_case(sk, eng, tmp_path, rtol=1e-9):
    g = np.load(os.path.join(GOLD, "plda.npz"), allow_pickle=False)
    rank_f = int(g["rank_f"])
    for tag, kw in (("it3", dict(nb_iter=3)), ("it1", dict(nb_iter=1)), ("scaled", dict(nb_iter=2, scaling_factor=0.6))):
        ss = _server(sk, eng, g["modelset"], g["segset"], g["stat1"])
        before = ss.stat1.copy()
        fa = sk.FactorAnalyser(engine=eng)
        out = os.path.join(str(tmp_path), "plda_" + tag)
        fa.plda(ss, rank_f, output_file_name=out, save_partial=(tag == "it3"), **kw)
        np.testing.assert_array_equal(ss.stat1, before)                      # the training statistics are not touched
        _close(fa.mean, g["mean_" + tag], 1e-13)
        _close(fa.Sigma, g["Sigma_" + tag], rtol)
        _close_up_to_signs(fa.F, g["F_" + tag], rtol)
        back = sk.FactorAnalyser.read(out + fa._ext())                       # the model file of the last iteration
        _close(back.F, fa.F, 0.0)
        _close(back.Sigma, fa.Sigma, 0.0)
        if tag == "it3":                                                     # partial products of iterations 0 and 1
            first = sk.FactorAnalyser.read(out + "_it-0" + fa._ext())
            _close_up_to_signs(first.F, g["F_it1"], rtol)
            assert os.path.exists(out + "_it-1" + fa._ext())
