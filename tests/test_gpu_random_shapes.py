"""Random shapes through the GMM entry points against the oracle (the part of tools/fuzz_gmm.py that fits a test:
small frame counts, the CPU oracle as the checker).  Each seed draws (C, D), a ragged set of sessions with empty and
single-frame ones, and compares skm_gmm_estep, skm_gmm_loglik and skm_bw_stats; bar 1e-4 of the largest entry of each
array, as everywhere in tests/."""
import numpy as np
import pytest

from oracle import sidekit_ref as orc
from conftest import rel_rows, rel_rownorm, check_first_order

pytestmark = pytest.mark.gpu

CS = [1, 2, 3, 7, 16, 31, 32, 33, 40, 64, 65, 96, 100, 128, 129, 160, 200, 256, 257, 300, 384, 385, 500, 512, 513, 600,
      768, 1000, 1024, 1025, 1500, 2048]
DS = [1, 2, 5, 13, 16, 20, 24, 31, 32, 33, 39, 40, 48, 59, 60, 63, 64, 65, 72, 80, 96, 100, 120, 128, 132, 140, 156, 159]
TOL = 1e-4


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def eng():
    import sidekit_amd
    return sidekit_amd.default_engine()


@pytest.mark.parametrize("seed", range(32))
def test_random_shape_vs_oracle(eng, seed):
    rs = np.random.RandomState(9000 + seed)
    C, D = int(rs.choice(CS)), int(rs.choice(DS))
    kind = rs.randint(4)
    ns = int(rs.randint(1, 30))
    lens = (rs.randint(0, 70, ns) if kind == 0 else rs.randint(1, 500, ns) if kind == 1
            else np.where(rs.rand(ns) < 0.3, 0, rs.randint(1, 300, ns)) if kind == 2 else rs.randint(1, 4, ns) * 32)
    if lens.sum() == 0:
        lens[0] = 5
    off = np.concatenate([[0], np.cumsum(lens)])
    T = int(off[-1])
    gmm = orc.random_gmm(C, D, seed=seed)
    frames = orc.synth_frames(T, dim=D, clusters=8, seed=100 + seed)
    X = eng.upload_frames(frames)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)      # as the drop-in classes pack for E-step / Baum-Welch
    what = "seed %d: C=%d D=%d T=%d sessions=%d" % (seed, C, D, T, ns)

    flat = eng.to_host(eng.estep(X, packed))
    N, F, S, llk = flat[:C], flat[C:C + C * D].reshape(C, D), flat[C + C * D:C + 2 * C * D].reshape(C, D), flat[-1]
    x64 = frames.astype(np.float64)
    rN, rF, rS, rsum = orc.expectation(gmm, x64)
    assert rel(N, rN) < TOL and rel(F, rF) < TOL and rel(S, rS) < TOL, what
    rows = (rel_rows(N, rN), rel_rows(S, rS), rel_rownorm(S, rS))      # element-wise, per component row
    assert max(rows) < TOL, (what, rows)
    check_first_order(F, rF, what=what)
    assert abs(llk - rsum) <= 1e-6 * abs(rsum), what
    rllk = orc.logsumexp_rows(orc.log_posterior(gmm, x64))[1]
    assert rel(eng.to_host(eng.loglik(X, eng.pack(gmm.w, gmm.mu, gmm.invcov))), rllk) < 1e-5, what

    s0, s1 = eng.bw_stats(X, off, packed)
    s0, s1 = eng.to_host(s0), eng.to_host(s1)
    r0, r1 = orc.accumulate_stat(gmm, [x64[off[i]:off[i + 1]] for i in range(ns)])
    assert rel(s0, r0) < TOL and rel(s1, r1) < TOL, what
    assert rel_rows(s0, r0) < TOL, (what, rel_rows(s0, r0))            # element-wise, per session row
    check_first_order(s1, r1, what=what)
    empty = lens == 0
    assert np.all(s0[empty] == 0) and np.all(s1[empty] == 0), what
    assert np.abs(s0.sum(1) - lens).max() <= 1e-5 * max(lens.max(), 1), what
