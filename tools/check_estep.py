"""E-step against the oracle for a list of C,D,T shapes: python tools/check_estep.py 96,20,5000 128,60,4000 ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sidekit_amd as sk
from oracle import sidekit_ref as orc
eng = sk.default_engine()
for arg in sys.argv[1:]:
    C, D, T = [int(v) for v in arg.split(",")]
    gmm = orc.random_gmm(C, D, seed=C + D)
    cep = orc.synth_frames(T, dim=D, clusters=16, seed=T)
    eng.profile(True)
    flat = eng.to_host(eng.estep(eng.upload_frames(cep), eng.pack(gmm.w, gmm.mu, gmm.invcov)))
    kern = sorted(eng.profile_read())
    eng.profile(False)
    N, F, S, llk = orc.expectation(gmm, cep.astype(np.float64))
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    print("C=%d D=%d T=%d %s: N %.2e F %.2e S %.2e llk %.2e  sumN %.3f (T=%d)" % (
        C, D, T, kern, rel(flat[:C], N), rel(flat[C:C + C * D], F.ravel()), rel(flat[C + C * D:C + 2 * C * D], S.ravel()),
        abs(flat[-1] - llk) / abs(llk), flat[:C].sum(), T), flush=True)
