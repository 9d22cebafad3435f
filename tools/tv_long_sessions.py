"""How the i-vector error grows with cond(L_s): sessions of 2e3 ... 1e5 frames, loading matrix after 3 EM iterations,
float32 against float64 chain, against the float64 oracle (the float32 chain on long sessions).
Also prints the a-priori bound FactorAnalyser uses to choose the chain.  python tools/tv_long_sessions.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sidekit_amd as sk  # noqa: E402
from oracle import sidekit_ref as orc  # noqa: E402

eng = sk.default_engine()
torch = eng.torch
C, D, R, S = 64, 20, 50, 300


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


for frames in (2000, 20000, 100000, 500000):
    rs = np.random.RandomState(frames % 9973)
    gmm = orc.random_gmm(C, D, seed=3)
    # speaker / session factors so that the trained loadings have a few dominant directions
    Vtrue = 0.3 * rs.randn(C * D, 8)
    y = rs.randn(S, 8)
    post = rs.dirichlet(np.full(C, 0.5), size=S) * frames
    nrep = np.repeat(post, D, axis=1)
    sd = 1.0 / np.sqrt(gmm.invcov.flatten())
    s1 = nrep * (gmm.mu.flatten()[None, :] + (y @ Vtrue.T) * sd[None, :]) + np.sqrt(nrep) * sd[None, :] * rs.randn(S, C * D)
    s0 = post.astype(np.float32)
    s1 = s1.astype(np.float32)
    Fm = 0.1 * rs.randn(C * D, R)
    for _ in range(3):                                  # three oracle EM iterations: a trained T
        A, Cc, Rr = orc.tv_accumulate(s0, s1, gmm, Fm, batch_size=300)
        Fm = orc.tv_mstep(A, Cc, Rr, D, min_div=True)
    iv, _ = orc.extract_ivectors(s0, s1, gmm, Fm)
    # cond(L_s) exactly, for the report
    index_map = np.repeat(np.arange(C), D)
    conds = []
    for s in range(0, S, 30):
        L = np.eye(R) + (Fm.T * s0[s, index_map].astype(np.float64)) @ Fm
        w = np.linalg.eigvalsh(L)
        conds.append(w[-1] / w[0])
    ft, naug = eng.tv_whiten(dev(s0), dev(s1), gmm.mu.flatten(), gmm.invcov.flatten())
    T32, G = eng.tv_prepare(dev(Fm), C, D)
    bound = sk.FactorAnalyser.chain_cond_bound(eng, dev(Fm), naug, C, D) if hasattr(sk.FactorAnalyser, "chain_cond_bound") else float("nan")
    out = []
    for chain in ("f32", "f64"):
        for unc in (False, True):
            e, _ = eng.tv_estep(naug, ft, T32, G, C, D, batch=128, uncertainty=unc, chain=chain)
            e = eng.to_host(e)
            err = np.abs(e - iv).max() / np.abs(iv).max()
            cos = ((e * iv).sum(1) / np.linalg.norm(e, axis=1) / np.linalg.norm(iv, axis=1)).min()
            out.append("%s%s %.1e (1-cos %.1e)" % (chain, "+inv" if unc else "", err, 1 - cos))
    print("frames %7d  cond(L) max %.1e  bound %.1e | %s" % (frames, max(conds), bound, " | ".join(out)), flush=True)
