"""Normalisation defect of the Baum-Welch statistics: sum_c stat0[s, c] must equal the session's frame count.
    python tools/diag_norm.py"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sidekit_amd as sk
eng = sk.default_engine(); dev = eng.device
rng = np.random.default_rng(3)
for C, D in [(2, 156), (16, 132), (64, 132), (100, 159), (256, 159), (512, 132), (1500, 156), (64, 60), (512, 60), (2048, 60), (300, 60)]:
    lens = rng.integers(1, 4, 30) * 32 + rng.integers(0, 2, 30) * 7
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    T = int(off[-1])
    X = (torch.from_numpy(rng.standard_normal((T, D)).astype(np.float32))).to(dev)
    w = rng.random(C) + 0.05; w /= w.sum()
    mu = rng.standard_normal((C, D)) * 0.5
    invcov = 1.0 / (0.5 + rng.random((C, D)))
    packed = eng.pack(w, mu, invcov)
    eng.profile(True)
    s0, s1 = eng.bw_stats(X, off, packed)
    torch.cuda.synchronize()
    kern = list(eng.profile_read().keys()); eng.profile(False)
    em = eng.estep(X, packed)
    L = torch.from_numpy(lens.astype(np.float64)).to(dev)
    defect = ((s0.sum(1) - L) / L)
    emdef = (em[:C].sum().item() - T) / T
    print("C=%4d D=%3d  BW defect mean %+.2e max|.| %.2e   EM defect %+.2e   %s" % (
        C, D, defect.mean().item(), defect.abs().max().item(), emdef, kern), flush=True)
