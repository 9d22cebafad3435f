"""GPU parity tests of the GMM path: the HIP kernels (through the C ABI / the Python drop-in
classes) against the CPU oracle and against the golden vectors of the real reference.

Tolerance (BASELINE.json north star): UBM w / mu / cov and N / F statistics within 1e-4
relative (float32 MFMA compute, float64 reductions).  "Relative" is measured normwise per
array (max |delta| / max |ref|): first-order statistics of a component have entries that cancel
to ~0, so an element-wise ratio is meaningless there; large entries are also checked
element-wise.
"""
import numpy as np
import pytest

from oracle import sidekit_ref as orc
from conftest import golden, rel, rel_large, rel_rows, rel_rownorm, check_first_order

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def sk():
    import sidekit_amd
    return sidekit_amd


@pytest.fixture(scope="module")
def eng(sk):
    return sk.default_engine()


def unpack(flat, C, D):
    return flat[:C], flat[C:C + C * D].reshape(C, D), flat[C + C * D:C + 2 * C * D].reshape(C, D), flat[-1]


def run_estep(eng, gmm, cep):
    X = eng.upload_frames(cep)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    return unpack(eng.to_host(eng.estep(X, packed)), gmm.C, gmm.D)


@pytest.mark.parametrize("tag", ["c8_d6", "c64_d60", "c40_d13"])
def test_estep_vs_reference_golden(eng, tag):
    g = golden("estep_%s.npz" % tag)
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    N, F, S, llk = run_estep(eng, gmm, g["cep"])
    assert rel(N, g["N"]) < TOL
    assert rel(F, g["F"]) < TOL
    assert rel(S, g["S"]) < TOL
    assert rel_large(N, g["N"]) < TOL
    assert rel_large(F, g["F"]) < TOL
    assert rel_large(S, g["S"]) < TOL
    # element-wise on the entries that carry their component's row (every N_c; F_c and S_c per component)
    assert rel_rows(N, g["N"]) < TOL and rel_rows(S, g["S"]) < TOL and rel_rownorm(S, g["S"]) < TOL
    check_first_order(F, g["F"])
    assert abs(llk - float(g["llk"])) < 1e-5 * abs(float(g["llk"]))
    # M-step from the device statistics reproduces the reference's next model
    orc.maximization(gmm, N, F, S)
    assert rel(gmm.w, g["m_w"]) < TOL
    assert rel(gmm.mu, g["m_mu"]) < TOL
    assert rel(1.0 / gmm.invcov, 1.0 / g["m_invcov"]) < TOL


@pytest.mark.parametrize("C,D,T", [(1, 60, 1000), (2, 60, 33), (33, 60, 4097), (512, 60, 20000), (96, 20, 5000),
                                   (64, 132, 3000), (130, 7, 2049), (256, 39, 7001), (2048, 60, 6000),
                                   (160, 60, 5000), (192, 20, 3000), (135, 36, 70000),      # 5 or 6 component tiles: fat kernel with padding
                                   # one-wave kernel (C <= 128): every ending of the logits GEMM (pair count even / odd, with /
                                   # without a tail group) at one, two and four component tiles; several tiles per wave
                                   (64, 20, 320), (64, 4, 320), (128, 36, 5000), (20, 36, 320), (100, 28, 777), (17, 16, 100),
                                   (128, 60, 300000), (32, 60, 200000), (48, 13, 150000),
                                   # wide rows (D = 64 .. 156, C <= 128): one component tile per wave, 2 or 4 waves; every
                                   # ending of the logits GEMM at 3, 4 and 5 statistics column tiles; padding tiles (C = 17,
                                   # 32, 40, 96, 100); several folds per wave
                                   (64, 64, 3000), (128, 68, 5000), (32, 72, 777), (96, 76, 4097), (64, 96, 320),
                                   (128, 100, 300000), (17, 104, 100), (100, 108, 2049), (128, 128, 3000),
                                   (64, 132, 200000), (40, 136, 1234), (128, 140, 999), (64, 156, 5000),
                                   # wide rows above 128 components: teams of C/128 workgroups (2, 4, 8, 16), padding tiles in the
                                   # last workgroup (C = 448), fewer tiles than teams, several folds
                                   (256, 132, 3000), (512, 100, 70000), (1024, 68, 4097), (2048, 132, 2500), (448, 128, 999),
                                   (256, 64, 600000),
                                   # team kernel (C = 1024 / 2048 / 4096): ragged last tile, fewer tiles than teams, one frame
                                   (1024, 60, 9999), (2048, 20, 12345), (2048, 60, 95), (1024, 64, 1), (4096, 60, 4100),
                                   (1000, 60, 7000), (2040, 20, 3000), (900, 36, 50000),        # last wave on padding tiles
                                   # rows that are not a multiple of 16 bytes on the team kernels (staging float by float)
                                   (2048, 39, 6000), (1024, 57, 9999), (4096, 13, 4100), (1024, 63, 70000), (2000, 19, 333)])
def test_estep_vs_oracle_shapes(eng, C, D, T):
    gmm = orc.random_gmm(C, D, seed=C + D)
    cep = orc.synth_frames(T, dim=D, clusters=16, seed=T)
    N, F, S, llk = run_estep(eng, gmm, cep)
    rN, rF, rS, rl = orc.expectation(gmm, cep.astype(np.float64))
    assert rel(N, rN) < TOL
    assert rel(F, rF) < TOL
    assert rel(S, rS) < TOL
    # north star: "within 1e-4 relative" -- element-wise on the entries that carry their component's row
    rows = (rel_rows(N, rN), rel_rows(S, rS), rel_rownorm(S, rS))
    assert max(rows) < TOL, rows
    check_first_order(F, rF)
    assert abs(llk - rl) < 1e-5 * abs(rl)
    assert abs(N.sum() - T) < 1e-4 * T          # posteriors sum to one per frame


def test_estep_accumulate_and_empty(eng):
    gmm = orc.random_gmm(16, 12, seed=1)
    cep = orc.synth_frames(3000, dim=12, clusters=8, seed=2)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    full = eng.to_host(eng.estep(eng.upload_frames(cep), packed))
    out = eng.estep(eng.upload_frames(cep[:1111]), packed)
    out = eng.estep(eng.upload_frames(cep[1111:]), packed, out=out, accumulate=True)
    part = eng.to_host(out)
    assert rel(part, full) < 1e-6
    # determinism: same launch twice is bit-identical
    again = eng.to_host(eng.estep(eng.upload_frames(cep), packed))
    np.testing.assert_array_equal(full, again)


def test_loglik_per_frame(eng):
    gmm = orc.random_gmm(100, 24, seed=3)
    cep = orc.synth_frames(2500, dim=24, clusters=8, seed=4)
    ll = eng.to_host(eng.loglik(eng.upload_frames(cep), eng.pack(gmm.w, gmm.mu, gmm.invcov)))
    _, ref = orc.logsumexp_rows(orc.log_posterior(gmm, cep.astype(np.float64)))
    assert np.abs(ll - ref).max() < 2e-4 * np.abs(ref).max()


def test_moments(eng):
    cep = orc.synth_frames(12345, dim=60, seed=9)
    m = eng.to_host(eng.moments(eng.upload_frames(cep)))
    x = cep.astype(np.float64)
    assert rel(m[:60], x.sum(0)) < 1e-12
    assert rel(m[60:], (x ** 2).sum(0)) < 1e-12


def _sessions(g):
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    return [g["frames"][off[i]:off[i + 1]] for i in range(len(g["lens"]))], off


def test_bw_stats_vs_reference_golden(eng):
    g = golden("accumulate_stat.npz")
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    sess, off = _sessions(g)
    s0, s1 = eng.bw_stats(eng.upload_frames(g["frames"]), off, eng.pack_stats(gmm.w, gmm.mu, gmm.invcov))
    s0, s1 = eng.to_host(s0), eng.to_host(s1)
    assert rel(s0, g["stat0"]) < TOL
    assert rel(s1, g["stat1"]) < TOL
    assert rel_large(s0, g["stat0"]) < TOL and rel_large(s1, g["stat1"]) < TOL     # element-wise on the entries that matter
    assert rel_rows(s0, g["stat0"]) < TOL                                           # ... of every session's row
    check_first_order(s1, g["stat1"])
    # per-session normwise too (short sessions must be as accurate as long ones)
    for i in range(len(sess)):
        assert rel(s0[i], g["stat0"][i]) < TOL
        assert rel(s1[i], g["stat1"][i]) < TOL


def test_bw_stats_ragged_and_empty_sessions(eng):
    C, D = 70, 60
    gmm = orc.random_gmm(C, D, seed=5)
    lens = [0, 1, 31, 32, 33, 0, 500, 64, 1]
    frames = orc.synth_frames(sum(lens), dim=D, seed=6)
    off = np.concatenate([[0], np.cumsum(lens)])
    s0, s1 = eng.bw_stats(eng.upload_frames(frames), off, eng.pack_stats(gmm.w, gmm.mu, gmm.invcov))
    s0, s1 = eng.to_host(s0), eng.to_host(s1)
    r0, r1 = orc.accumulate_stat(gmm, [frames[off[i]:off[i + 1]].astype(np.float64) for i in range(len(lens))])
    assert rel(s0, r0) < TOL and rel(s1, r1) < TOL
    assert rel_rows(s0, r0) < TOL and rel_rownorm(s1, r1) < TOL, (rel_rows(s0, r0), rel_rownorm(s1, r1))
    check_first_order(s1, r1)
    assert np.all(s0[0] == 0) and np.all(s1[5] == 0)
    assert np.allclose(s0.sum(1), lens, rtol=1e-4)


@pytest.mark.parametrize("C,D", [(512, 60), (1024, 60), (2048, 20), (4096, 60)])
def test_bw_stats_team_kernel_ragged_and_empty_sessions(eng, C, D):
    """C = 1024 / 2048 / 4096 takes the single-sweep team kernel in Baum-Welch mode (one pipeline segment per
    session, sessions cut over the teams by frame count): empty sessions at the start, in the middle, at a
    team boundary and at the end, one-frame and non-multiple-of-32 sessions, more teams than sessions' tiles;
    cross-checked against the two-sweep path (SKM_TEAM=0) too."""
    import os
    gmm = orc.random_gmm(C, D, seed=9)
    lens = [0, 0, 1, 31, 32, 33, 0, 257, 64, 1, 0, 700, 95, 0, 0]
    frames = orc.synth_frames(sum(lens), dim=D, seed=10)
    off = np.concatenate([[0], np.cumsum(lens)])
    X = eng.upload_frames(frames)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    s0, s1 = eng.bw_stats(X, off, packed)
    s0, s1 = eng.to_host(s0), eng.to_host(s1)
    r0, r1 = orc.accumulate_stat(gmm, [frames[off[i]:off[i + 1]].astype(np.float64) for i in range(len(lens))])
    assert rel(s0, r0) < TOL and rel(s1, r1) < TOL
    assert rel_rows(s0, r0) < TOL and rel_rownorm(s1, r1) < TOL, (rel_rows(s0, r0), rel_rownorm(s1, r1))
    check_first_order(s1, r1)
    for i, n in enumerate(lens):
        if n == 0:
            assert np.all(s0[i] == 0) and np.all(s1[i] == 0)
        else:
            assert rel(s0[i], r0[i]) < TOL and rel(s1[i], r1[i]) < TOL
    assert np.allclose(s0.sum(1), lens, rtol=1e-4)
    os.environ["SKM_TEAM"] = "0"
    try:
        t0, t1 = eng.bw_stats(X, off, packed)
    finally:
        del os.environ["SKM_TEAM"]
    assert rel(s0, eng.to_host(t0)) < 5e-5 and rel(s1, eng.to_host(t1)) < 5e-5   # two float32 paths, each within 1e-4 of the oracle


@pytest.mark.parametrize("C,D", [(64, 132), (128, 60), (96, 20), (100, 64), (40, 156), (128, 128), (17, 100),    # one tile per wave
                                 # one or two component tiles: workgroups of two waves, two per CU (three-tile frame ring)
                                 (64, 60), (33, 20), (20, 13), (64, 39), (5, 7), (48, 64), (64, 96), (64, 160 - 4),
                                 (256, 60), (200, 36), (130, 8), (256, 100), (512, 132), (480, 72), (1024, 96),  # two tiles per wave
                                 (2048, 68),
                                 (448, 60), (1000, 20), (2040, 36),       # four tiles per wave, the last wave on padding tiles
                                 (2048, 39), (512, 57), (1024, 13), (128, 39), (256, 13), (96, 63)])   # rows of any length
def test_bw_stats_small_blocks_and_wide_rows(eng, C, D):
    """Baum-Welch on the single-sweep segment kernel below 512 components (one or two component tiles per wave, a team
    of one workgroup) and on rows of 64..159 features (teams of 256-component workgroups): ragged / empty sessions,
    models that end inside a component tile or leave whole waves on padding; against the oracle and against the
    two-sweep path (SKM_BW_SMALL=0)."""
    import os
    gmm = orc.random_gmm(C, D, seed=C + D)
    lens = [0, 0, 1, 31, 32, 33, 0, 257, 64, 1, 0, 700, 95, 0, 0]
    frames = orc.synth_frames(sum(lens), dim=D, seed=10)
    off = np.concatenate([[0], np.cumsum(lens)])
    X = eng.upload_frames(frames)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    s0, s1 = eng.bw_stats(X, off, packed)
    s0, s1 = eng.to_host(s0), eng.to_host(s1)
    r0, r1 = orc.accumulate_stat(gmm, [frames[off[i]:off[i + 1]].astype(np.float64) for i in range(len(lens))])
    assert rel(s0, r0) < TOL and rel(s1, r1) < TOL
    assert rel_rows(s0, r0) < TOL and rel_rownorm(s1, r1) < TOL, (rel_rows(s0, r0), rel_rownorm(s1, r1))
    check_first_order(s1, r1)
    for i, n in enumerate(lens):
        if n == 0:
            assert np.all(s0[i] == 0) and np.all(s1[i] == 0)
        else:
            assert rel(s0[i], r0[i]) < TOL and rel(s1[i], r1[i]) < TOL
    assert np.allclose(s0.sum(1), lens, rtol=1e-4)
    os.environ["SKM_BW_SMALL"] = "0"
    try:
        t0, t1 = eng.bw_stats(X, off, packed)
    finally:
        del os.environ["SKM_BW_SMALL"]
    # two float32 paths, each within 1e-4 of the oracle (logits of 2 * 156 terms round further apart than those of 120)
    tol2 = 5e-5 if D <= 64 else 1e-4
    assert rel(s0, eng.to_host(t0)) < tol2 and rel(s1, eng.to_host(t1)) < tol2
    # the kernel that ran is the one the plan names (round 3: a model of one or two component tiles takes the two-wave
    # workgroups of the segment kernel; it used to stay on the two sweeps)
    eng.profile(True)
    eng.bw_stats(X, off, packed)
    eng.torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    assert ("k_bw_team" in prof) == (eng.lib.skm_bw_team_count(eng.ctx, C, D) > 0), prof
    njt, ct = (D + 1 + 31) // 32, (C + 31) // 32
    if ct <= 2:         # two-wave workgroups: wide rows always, two tiles from 32 columns on; else the two sweeps
        assert ("k_bw_team" in prof) == (njt >= 3 or (ct == 2 and njt == 2)), prof


@pytest.mark.parametrize("C,D", [(2, 156), (16, 132), (100, 159), (256, 159), (1500, 156), (64, 60), (300, 60),   # two sweeps
                                 (64, 132), (512, 132), (512, 60), (2048, 60), (128, 39)])                         # one kernel
def test_posteriors_of_a_frame_sum_to_one(eng, C, D):
    """Size-independent property: sum_c stat0[s, c] = frames of session s and sum_c N_c = T, to float32 rounding of one
    log-sum-exp -- not to the accumulation error of a logit (1e-4 at 156 columns).  On the two-sweep path this holds
    because sweep 2 accumulates the logits bit-for-bit as sweep 1 did; it missed by 4e-5 when sweep 2 started its
    accumulator at kc - lse."""
    rng = np.random.default_rng(C * 1000 + D)
    lens = rng.integers(1, 4, 30) * 32 + rng.integers(0, 2, 30) * 7
    off = np.concatenate([[0], np.cumsum(lens)])
    gmm = orc.random_gmm(C, D, seed=C + D)
    frames = orc.synth_frames(int(off[-1]), dim=D, seed=12)
    X = eng.upload_frames(frames)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    s0, _ = eng.bw_stats(X, off, packed)
    s0 = eng.to_host(s0)
    defect = np.abs(s0.sum(1) - lens) / lens
    assert defect.max() < 1e-5, defect.max()
    N = unpack(eng.to_host(eng.estep(X, packed)), C, D)[0]
    assert abs(N.sum() - off[-1]) / off[-1] < 2e-6


@pytest.mark.parametrize("C,D", [(512, 60), (2048, 20), (128, 60), (512, 132), (64, 132)])
def test_bw_stats_few_long_sessions_are_split(eng, C, D):
    """Few long sessions on the team kernel: the host cuts them into 32-frame-aligned pieces (engine._bw_split_plan) and
    adds the pieces' rows with a 0/1 float64 product.  Same statistics as the unsplit launch (the float32 MFMA accumulators
    are folded into float64 at other tile boundaries: ~2e-6) and as the oracle; an empty session between two long ones
    stays zero."""
    gmm = orc.random_gmm(C, D, seed=31)
    lens = [5000, 0, 3001, 2049]
    frames = orc.synth_frames(sum(lens), dim=D, seed=32)
    off = np.concatenate([[0], np.cumsum(lens)])
    plan = eng._bw_split_plan(off.astype(np.int64), C, D)
    assert plan is not None and len(plan[1]) > len(lens)
    pieces, owner = plan
    assert pieces[0] == 0 and pieces[-1] == off[-1] and np.all(np.diff(pieces) >= 0)
    assert set(off.tolist()) <= set(pieces.tolist())
    for k, s in enumerate(owner):                                  # every piece lies inside its session, 32-aligned start
        assert off[s] <= pieces[k] and pieces[k + 1] <= off[s + 1] and (pieces[k] - off[s]) % 32 == 0
    X = eng.upload_frames(frames)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    s0, s1 = eng.bw_stats(X, off, packed)
    u0, u1 = eng.bw_stats(X, off, packed, split_long=False)
    s0, s1, u0, u1 = eng.to_host(s0), eng.to_host(s1), eng.to_host(u0), eng.to_host(u1)
    assert s0.shape == (len(lens), C) and s1.shape == (len(lens), C * D)
    assert rel(s0, u0) < 2e-5 and rel(s1, u1) < 2e-5
    assert np.all(s0[1] == 0) and np.all(s1[1] == 0)
    r0, r1 = orc.accumulate_stat(gmm, [frames[off[i]:off[i + 1]].astype(np.float64) for i in range(len(lens))])
    assert rel(s0, r0) < TOL and rel(s1, r1) < TOL
    assert np.allclose(s0.sum(1), lens, rtol=1e-4)


def test_statserver_accumulate_stat_dropin(sk):
    g = golden("accumulate_stat.npz")
    sess, off = _sessions(g)
    names = ["s%03d" % i for i in range(len(sess))]
    fs = sk.FeaturesServer(features=dict(zip(names, sess)))
    ubm = sk.Mixture()
    ubm.w, ubm.mu, ubm.invcov = g["w"].copy(), g["mu"].copy(), g["invcov"].copy()
    ubm.cov_var_ctl = 1.0 / ubm.invcov
    ubm._compute_all()
    idmap = sk.IdMap().set(["spk%d" % (i % 5) for i in range(len(sess))], names)
    ss = sk.StatServer(idmap, ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(len(sess)), num_thread=1)
    assert ss.validate()
    assert rel(ss.stat0, g["stat0"]) < TOL
    assert rel(ss.stat1, g["stat1"]) < TOL
    assert rel_rows(ss.stat0, g["stat0"]) < TOL
    check_first_order(ss.stat1, g["stat1"])
    with pytest.raises(Exception):
        bad = sk.FeaturesServer(features={n: np.zeros((5, ubm.dim() + 1)) for n in names})
        ss.accumulate_stat(ubm=ubm, feature_server=bad, seg_indices=range(2))


def test_statserver_accumulate_stat_pipelined_batches(sk, eng):
    """The drop-in call on a job of several device batches (2.2 M frames: three batches of about 2^20 frames, uploads,
    kernels and the copy off the device overlapped on two streams and a worker thread): every row lands in its place
    and equals what one call of the kernel entry point over all frames returns; a sample of sessions against the
    oracle."""
    C, D, S = 64, 20, 1300
    rs = np.random.RandomState(77)
    lens = rs.randint(500, 3000, size=S)
    lens[[3, 700]] = [1, 2]
    off = np.concatenate([[0], np.cumsum(lens)])
    frames = orc.synth_frames(int(off[-1]), dim=D, clusters=16, seed=78)
    assert frames.shape[0] > 2 * (1 << 20)
    gmm = orc.random_gmm(C, D, seed=79)
    names = ["s%04d" % i for i in range(S)]
    fs = sk.FeaturesServer(features={n: frames[off[i]:off[i + 1]] for i, n in enumerate(names)})
    ubm = sk.Mixture()
    ubm.w, ubm.mu, ubm.invcov = gmm.w.copy(), gmm.mu.copy(), gmm.invcov.copy()
    ubm.cov_var_ctl = 1.0 / ubm.invcov
    ubm._compute_all()
    ss = sk.StatServer(sk.IdMap().set(["spk%d" % (i % 50) for i in range(S)], names), ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(S), num_thread=1)
    assert ss.stat0.shape == (S, C) and ss.stat1.shape == (S, C * D)
    assert np.abs(ss.stat0.sum(1) - lens).max() / lens.max() < 1e-5
    t0, t1 = eng.bw_stats(eng.upload_frames(frames), off, eng.pack_stats(gmm.w, gmm.mu, gmm.invcov))
    t0, t1 = eng.to_host(t0), eng.to_host(t1)
    for i in range(S):      # row by row: a row that landed in the wrong place or was never written shows here
        assert np.abs(ss.stat0[i] - t0[i]).max() <= 5e-5 * max(np.abs(t0[i]).max(), 1e-30), i
        assert np.abs(ss.stat1[i] - t1[i]).max() <= 5e-5 * max(np.abs(t1[i]).max(), 1e-30), i
    pick = [0, 3, 411, 700, 1299]
    r0, r1 = orc.accumulate_stat(gmm, [frames[off[i]:off[i + 1]].astype(np.float64) for i in pick])
    assert rel(ss.stat0[pick], r0) < TOL and rel(ss.stat1[pick], r1) < TOL


def test_em_split_dropin_vs_reference_golden(sk):
    g = golden("em_split.npz")
    chunks = {"f%d" % i: g["chunk%d" % i] for i in range(4)}
    fs = sk.FeaturesServer(features=chunks)
    ubm = sk.Mixture()
    llk = ubm.EM_split(features_server=fs, feature_list=list(chunks.keys()), distrib_nb=8, num_thread=1)
    assert len(llk) == 5
    assert rel(np.array(llk), g["llk"]) < TOL
    assert rel(ubm.w, g["w"]) < TOL
    assert rel(ubm.mu, g["mu"]) < TOL
    assert rel(1.0 / ubm.invcov, 1.0 / g["invcov"]) < TOL
    # UBM w / mu / covariance element-wise (per component, the entries that carry the row)
    assert rel_rows(ubm.w, g["w"]) < TOL and rel_rows(1.0 / ubm.invcov, 1.0 / g["invcov"]) < TOL
    check_first_order(ubm.mu, g["mu"])
    assert rel(ubm.cov_var_ctl, g["ctl"]) < 1e-6
    assert ubm.validate()


def test_em_uniform_dropin_vs_reference_golden(sk):
    g = golden("em_uniform.npz")
    ubm = sk.Mixture()
    llk = ubm.EM_uniform(g["cep"], 4, iteration_min=3, iteration_max=10, llk_gain=0.01, do_init=True)
    assert len(llk) == len(g["llk"])
    assert rel(np.array(llk), g["llk"]) < TOL
    assert rel(ubm.w, g["w"]) < TOL
    assert rel(ubm.mu, g["mu"]) < TOL
    assert rel(1.0 / ubm.invcov, 1.0 / g["invcov"]) < TOL
    # UBM w / mu / covariance element-wise (per component, the entries that carry the row)
    assert rel_rows(ubm.w, g["w"]) < TOL and rel_rows(1.0 / ubm.invcov, 1.0 / g["invcov"]) < TOL
    check_first_order(ubm.mu, g["mu"])


def test_one_em_step_from_identical_params_large(sk, eng):
    """Step-wise parity at a size where EM is chaotic end-to-end."""
    C, D, T = 256, 60, 60000
    gmm = orc.random_gmm(C, D, seed=77)
    cep = orc.synth_frames(T, dim=D, seed=78)
    N, F, S, llk = run_estep(eng, gmm, cep)
    rN, rF, rS, rl = orc.expectation(gmm, cep.astype(np.float64))
    a, b = gmm.copy(), gmm.copy()
    orc.maximization(a, N, F, S)
    orc.maximization(b, rN, rF, rS)
    assert rel(a.w, b.w) < TOL and rel(a.mu, b.mu) < TOL and rel(1 / a.invcov, 1 / b.invcov) < TOL


def _mixture_from(sk, g):
    m = sk.Mixture()
    m.w, m.mu, m.invcov = g["w"].copy(), g["mu"].copy(), g["invcov"].copy()
    m.cov_var_ctl = 1.0 / m.invcov
    m._compute_all()
    return m


@pytest.mark.parametrize("tag", ["c8_d6", "c40_d13"])
def test_log_posterior_matrix_vs_reference_golden(sk, tag):
    """Mixture.compute_log_posterior_probabilities (mixture.py:461-487): the T x C matrix itself.
    float32 logits: absolute error scales with the magnitude of the quadratic form, so the bound is
    1e-5 of the largest |lp| (well inside what 1e-4-relative posteriors need near the row maximum)."""
    g = golden("estep_%s.npz" % tag)
    m = _mixture_from(sk, g)
    lp = m.compute_log_posterior_probabilities(g["cep"])
    assert lp.shape == g["lp"].shape and lp.dtype == np.float64
    assert np.abs(lp - g["lp"]).max() <= 1e-5 * np.abs(g["lp"]).max()
    # posteriors from it agree with the reference's to 1e-4 relative on entries that matter
    pp = np.exp(lp - g["log_lk"][:, None])
    big = g["pp"] > 1e-3
    assert (np.abs(pp - g["pp"])[big] / g["pp"][big]).max() < TOL


@pytest.mark.parametrize("C,D,T", [(64, 60, 1000), (130, 20, 333), (2048, 60, 300), (1, 7, 50)])
def test_log_posterior_matrix_mu_override(sk, C, D, T):
    """`mu=` (MAP-adapted means, gmm_scoring.py:102-111) against the oracle; single-row input too."""
    gmm = orc.random_gmm(C, D, seed=5)
    cep = orc.synth_frames(T, D, seed=6)
    m = sk.Mixture()
    m.w, m.mu, m.invcov = gmm.w.copy(), gmm.mu.copy(), gmm.invcov.copy()
    m.cov_var_ctl = 1.0 / m.invcov
    m._compute_all()
    mu2 = gmm.mu + 0.1 * np.random.RandomState(3).randn(C, D)
    for mu in (None, mu2, mu2.reshape(-1)):
        ref = orc.log_posterior(gmm, cep, mu=mu)
        lp = m.compute_log_posterior_probabilities(cep, mu=mu)
        assert lp.shape == (T, C)
        assert np.abs(lp - ref).max() <= 1e-5 * np.abs(ref).max()
    one = m.compute_log_posterior_probabilities(cep[0])
    assert one.shape == (1, C)
    assert np.abs(one - orc.log_posterior(gmm, cep[:1])).max() <= 1e-5 * np.abs(ref).max()


def test_team_kernel_matches_two_sweeps_and_accumulates(eng):
    """C = 2048 takes the single-sweep team kernel (cross-workgroup logsumexp hand-off); SKM_TEAM=0 is the
    two-sweep path.  Same statistics to float32 rounding, bit-identical on repeat, and `accumulate` adds."""
    import os
    C, D, T = 2048, 60, 70001
    gmm = orc.random_gmm(C, D, seed=3)
    cep = orc.synth_frames(T, D, seed=4)
    X = eng.upload_frames(cep)
    packed = eng.pack(gmm.w, gmm.mu, gmm.invcov)
    a = eng.to_host(eng.estep(X, packed))
    b = eng.to_host(eng.estep(X, packed))
    np.testing.assert_array_equal(a, b)
    assert np.isfinite(a).all()
    os.environ["SKM_TEAM"] = "0"
    try:
        two = eng.to_host(eng.estep(X, packed))
    finally:
        del os.environ["SKM_TEAM"]
    N1, F1, S1, l1 = unpack(a, C, D)
    N2, F2, S2, l2 = unpack(two, C, D)
    assert rel(N1, N2) < 2e-5 and rel(F1, F2) < 2e-5 and rel(S1, S2) < 2e-5 and abs(l1 - l2) < 1e-5 * abs(l2)
    out = eng.estep(X, packed)
    out = eng.estep(X, packed, out=out, accumulate=True)
    np.testing.assert_allclose(eng.to_host(out), 2.0 * a, rtol=1e-12)


def test_full_size_invariants_cfg2(eng):
    """BASELINE configs[1] at full size (C=512, D=60, 10 M frames resident in HBM) through size-independent
    properties of the statistics: sum_c N_c = T, sum_c F_c = sum_t x_t, sum_c S_c = sum_t x_t^2 (posteriors
    sum to one per frame), bit-identical repeat, and a 1/64 sub-sample cross-checked against the oracle."""
    import torch
    C, D, T = 512, 60, 10_000_000
    g = torch.Generator(device=eng.device)
    g.manual_seed(99)
    X = torch.randn(T, D, generator=g, device=eng.device, dtype=torch.float32)
    gmm = orc.random_gmm(C, D, seed=21)
    packed = eng.pack(gmm.w, gmm.mu, gmm.invcov)
    out = eng.to_host(eng.estep(X, packed))
    N, F, S, llk = unpack(out, C, D)
    assert np.isfinite(out).all()
    assert abs(N.sum() - T) < 2e-5 * T     # float32 posteriors: v_exp_f32 / v_rcp_f32 are 1-ulp instructions
    mom = np.asarray(eng.to_host(eng.moments(X)))
    m1, m2 = mom[:D], mom[D:]
    assert np.abs(F.sum(0) - m1).max() < 1e-5 * np.abs(m2).max()
    assert rel(S.sum(0), m2) < 1e-5
    np.testing.assert_array_equal(out, eng.to_host(eng.estep(X, packed)))
    sub = X[::64].contiguous()
    Ns, Fs, Ss, ls = unpack(eng.to_host(eng.estep(sub, packed)), C, D)
    rN, rF, rS, rl = orc.expectation(gmm, sub.cpu().numpy().astype(np.float64))
    assert rel(Ns, rN) < TOL and rel(Fs, rF) < TOL and rel(Ss, rS) < TOL and abs(ls - rl) < 1e-5 * abs(rl)


def test_full_size_invariants_cfg3_cfg4_shape(eng):
    """C=2048 at a per-GPU shard size of configs[2]/[3] (4 M frames, 2000 sessions x 2000 frames): the
    single-sweep team kernels.  EM: sum_c N_c = T and the first/second moment identities; Baum-Welch: every
    stat0 row sums to its session length, stat1 rows sum over components to the session's sum of frames,
    and the session statistics add up to the EM statistics of the same frames."""
    import torch
    C, D, S, L = 2048, 60, 2000, 2000
    T = S * L
    g = torch.Generator(device=eng.device)
    g.manual_seed(7)
    X = torch.randn(T, D, generator=g, device=eng.device, dtype=torch.float32)
    gmm = orc.random_gmm(C, D, seed=22)
    packed = eng.pack(gmm.w, gmm.mu, gmm.invcov)
    N, F, Sx, llk = unpack(eng.to_host(eng.estep(X, packed)), C, D)
    assert abs(N.sum() - T) < 2e-5 * T     # float32 posteriors: v_exp_f32 / v_rcp_f32 are 1-ulp instructions
    mom = np.asarray(eng.to_host(eng.moments(X)))
    m1, m2 = mom[:D], mom[D:]
    assert np.abs(F.sum(0) - m1).max() < 1e-5 * np.abs(m2).max()
    assert rel(Sx.sum(0), m2) < 1e-5
    seg = torch.arange(0, S + 1, dtype=torch.int64, device=eng.device) * L
    s0, s1 = eng.bw_stats(X, seg, packed)
    rows = s0.sum(1).cpu().numpy()
    assert np.abs(rows - L).max() < 2e-5 * L
    assert rel(s0.sum(0).cpu().numpy(), N) < 1e-5
    assert rel(s1.sum(0).cpu().numpy().reshape(C, D), F) < 1e-5
    sess_sum = X[:L].double().sum(0).cpu().numpy()
    got = s1[0].cpu().numpy().reshape(C, D).sum(0)
    assert np.abs(got - sess_sum).max() < 1e-4 * np.abs(sess_sum).max() + 1e-3


# ------------------------------------------------------------------ next row: MAP adaptation + GMM-UBM scoring
def _ubm_from(sk, w, mu, invcov):
    ubm = sk.Mixture()
    ubm.w, ubm.mu, ubm.invcov = np.array(w, dtype="d"), np.array(mu, dtype="d"), np.array(invcov, dtype="d")
    ubm.cov_var_ctl = 1.0 / ubm.invcov
    ubm._compute_all()
    return ubm


def test_segment_mean(eng):
    import torch
    rs = np.random.RandomState(3)
    v = rs.randn(5000).astype(np.float32) * 50 - 70
    off = np.array([0, 1, 1, 300, 1301, 5000], dtype=np.int64)       # a 1-frame and an empty segment
    got = eng.to_host(eng.segment_mean(torch.from_numpy(v).to(eng.device), off))
    ref = np.array([v[off[i]:off[i + 1]].astype(np.float64).mean() if off[i + 1] > off[i] else np.nan
                    for i in range(len(off) - 1)])
    assert np.isnan(got[1])
    m = ~np.isnan(ref)
    np.testing.assert_allclose(got[m], ref[m], rtol=1e-13)


def test_gmm_scoring_dropin_vs_reference_golden(sk):
    """adapt_mean_map_multisession (statserver.py:986-1028) + gmm_scoring (gmm_scoring.py:56-185) against the
    scores of the real reference.  Scores are differences of mean frame log-likelihoods (|llk| ~ 20): float32
    logits bound the absolute error, 2e-4 absolute on scores of magnitude 0.1-3."""
    g = golden("gmm_scoring.npz")
    ubm = _ubm_from(sk, g["w"], g["mu"], g["invcov"])
    off = np.concatenate([[0], np.cumsum(g["enr_lens"])])
    names = ["e%03d" % i for i in range(len(g["enr_lens"]))]
    feats = {n: g["enr_frames"][off[i]:off[i + 1]] for i, n in enumerate(names)}
    toff = np.concatenate([[0], np.cumsum(g["tst_lens"])])
    feats.update({str(n): g["tst_frames"][toff[i]:toff[i + 1]] for i, n in enumerate(g["ndx_segset"])})
    fs = sk.FeaturesServer(features=feats)
    ss = sk.StatServer(sk.IdMap().set([str(x) for x in g["enr_spk"]], names), ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(len(names)))
    enroll = ss.adapt_mean_map_multisession(ubm, r=16)
    assert list(enroll.modelset) == [str(x) for x in g["enroll_modelset"]]
    assert rel(enroll.stat1, g["enroll_stat1"]) < TOL
    ndx = sk.Ndx()
    ndx.modelset = np.array([str(x) for x in g["ndx_modelset"]])
    ndx.segset = np.array([str(x) for x in g["ndx_segset"]])
    ndx.trialmask = g["trialmask"].astype(bool)
    sc = sk.gmm_scoring(ubm, enroll, ndx, fs)
    assert sc.validate()
    assert (sc.scoremat[~ndx.trialmask] == 0).all()
    assert np.abs(sc.scoremat - g["scores"]).max() < 2e-4, np.abs(sc.scoremat - g["scores"]).max()


def test_gmm_scoring_vs_oracle_c512(sk):
    """Headline-shaped UBM (C=512, D=60), models absent from the enrolment and segments absent from the server
    are dropped as the reference's ndx.filter does."""
    C, D = 512, 60
    gmm = orc.random_gmm(C, D, seed=5)
    ubm = _ubm_from(sk, gmm.w, gmm.mu, gmm.invcov)
    rs = np.random.RandomState(6)
    models = np.array(["m%02d" % i for i in range(6)])
    means = gmm.mu.reshape(1, -1) + 0.05 * rs.randn(5, C * D)           # m05 is not enrolled
    enroll = sk.StatServer()
    enroll.modelset = models[:5].copy()
    enroll.segset = models[:5].copy()
    enroll.start = np.empty(5, "|O")
    enroll.stop = np.empty(5, "|O")
    enroll.stat0 = np.ones((5, 1))
    enroll.stat1 = means
    segs = np.array(["t%02d" % i for i in range(7)])
    lens = [257, 31, 1000, 1, 640, 333, 90]
    feats = {s: orc.synth_frames(L, dim=D, seed=50 + i) for i, (s, L) in enumerate(zip(segs[:6], lens))}  # t06 missing
    ndx = sk.Ndx()
    ndx.modelset, ndx.segset = models, segs
    ndx.trialmask = rs.rand(6, 7) < 0.7
    sc = sk.gmm_scoring(ubm, enroll, ndx, sk.FeaturesServer(features=feats))
    assert list(sc.modelset) == list(models[:5]) and list(sc.segset) == list(segs[:6])
    ref = orc.gmm_scoring(gmm, list(models[:5]), means, sc.modelset, sc.segset, sc.scoremask,
                          {s: np.asarray(f, dtype=np.float64) for s, f in feats.items()})
    assert np.abs(sc.scoremat - ref).max() < 2e-4, np.abs(sc.scoremat - ref).max()
    assert (sc.scoremat[~sc.scoremask] == 0).all()


@pytest.mark.parametrize("seed", range(8))
def test_gmm_scoring_random_shapes_vs_oracle(sk, seed):
    """Random (C, D), numbers of models and segments, segment lengths from 1 frame to two thousand, random trial masks
    (rows / columns without any trial included) against the oracle's log-likelihood ratios."""
    rs = np.random.RandomState(7000 + seed)
    C = int(rs.choice([1, 2, 33, 64, 100, 256, 512, 1000, 2048]))
    D = int(rs.choice([5, 13, 20, 39, 60, 64, 100, 132]))
    M, K = int(rs.randint(1, 9)), int(rs.randint(1, 12))
    gmm = orc.random_gmm(C, D, seed=seed)
    ubm = _ubm_from(sk, gmm.w, gmm.mu, gmm.invcov)
    models = np.array(["m%02d" % i for i in range(M)])
    means = gmm.mu.reshape(1, -1) + 0.05 * rs.randn(M, C * D)
    enroll = sk.StatServer()
    enroll.modelset = models.copy()
    enroll.segset = models.copy()
    enroll.start = np.empty(M, "|O")
    enroll.stop = np.empty(M, "|O")
    enroll.stat0 = np.ones((M, 1))
    enroll.stat1 = means
    segs = np.array(["t%02d" % i for i in range(K)])
    lens = [int(rs.choice([1, 2, 31, 32, 33, int(rs.randint(34, 2000))])) for _ in range(K)]
    feats = {s: orc.synth_frames(L, dim=D, clusters=8, seed=300 + 20 * seed + i) for i, (s, L) in enumerate(zip(segs, lens))}
    ndx = sk.Ndx()
    ndx.modelset, ndx.segset = models, segs
    ndx.trialmask = rs.rand(M, K) < 0.6
    sc = sk.gmm_scoring(ubm, enroll, ndx, sk.FeaturesServer(features=feats))
    assert list(sc.modelset) == list(models) and list(sc.segset) == list(segs)
    ref = orc.gmm_scoring(gmm, list(models), means, sc.modelset, sc.segset, sc.scoremask,
                          {s: np.asarray(f, dtype=np.float64) for s, f in feats.items()})
    what = "seed %d: C=%d D=%d models=%d segments=%d lens=%s" % (seed, C, D, M, K, lens)
    assert np.abs(sc.scoremat - ref).max() < 2e-4, (what, np.abs(sc.scoremat - ref).max())
    assert (sc.scoremat[~sc.scoremask] == 0).all(), what


@pytest.mark.parametrize("C,D", [(8, 6), (40, 13), (512, 60), (96, 20)])
def test_device_mstep_and_repack_equal_host_path(sk, eng, C, D):
    """skm_gmm_mstep_pack (M-step + variance control + re-pack in HBM) against the host float64 M-step of the
    drop-in (mixture.py:590-611) followed by skm_gmm_pack: parameters to 1e-14, the packed float32 image
    bit-identical up to one float32 ulp of the log-domain constants, including a component that hits the
    variance floor, one that hits the ceiling and an empty one (N = 0 -> NaN, as NumPy gives)."""
    import copy
    torch = eng.torch
    rs = np.random.RandomState(C + D)
    gmm = orc.random_gmm(C, D, seed=3)
    cep = orc.synth_frames(4000, dim=D, clusters=max(2, C // 4), seed=5, dtype=np.float64)
    N, F, S, llk = orc.expectation(gmm, cep)
    ctl = 1.0 / gmm.invcov
    ctl[0] *= 1e4                                 # floor above the estimate: component 0 is floored
    ctl[1] *= 1e-4                                # ceiling below the estimate: component 1 is capped
    stats = np.concatenate([N, F.ravel(), S.ravel(), [llk]])
    m = sk.Mixture()
    m.w, m.mu, m.invcov = gmm.w.copy(), gmm.mu.copy(), gmm.invcov.copy()
    m.cov_var_ctl = ctl.copy()
    m._compute_all()
    accum = copy.deepcopy(m)
    accum.w, accum.mu, accum.invcov = N.copy(), F.copy(), S.copy()
    host = copy.deepcopy(m)
    host._maximization(accum, ceil_cov=10, floor_cov=1e-2)
    assert np.any(1.0 / host.invcov[0] == 1e-2 * ctl[0]) and np.any(1.0 / host.invcov[1] == 10 * ctl[1])

    packed = eng.pack(m.w, m.mu, m.invcov)
    llk_dev = torch.zeros(1, dtype=torch.float64, device=eng.device)
    params = eng.mstep_pack(torch.from_numpy(stats).to(eng.device), torch.from_numpy(ctl.ravel().copy()).to(eng.device),
                            packed, 10, 1e-2, llk_mean=llk_dev)
    flat = eng.to_host(params)
    assert rel(flat[:C], host.w) < 1e-14
    assert rel(flat[C:C + C * D].reshape(C, D), host.mu) < 1e-14
    assert rel(flat[C + C * D:C + 2 * C * D].reshape(C, D), host.invcov) < 1e-14
    assert abs(flat[-1] - N.sum()) <= 1e-12 * N.sum()
    assert abs(float(llk_dev.item()) - llk / N.sum()) <= 1e-14 * abs(llk / N.sum())
    ref_img = eng.pack(host.w, host.mu, host.invcov).packed.cpu().numpy().view(np.float32)
    img = packed.packed.cpu().numpy().view(np.float32)
    assert img.shape == ref_img.shape
    assert np.abs(img - ref_img).max() <= 2e-7 * np.abs(ref_img[np.abs(ref_img) < 1e29]).max()
    # the E-step on the re-packed model equals the E-step on the host-packed one
    X = eng.upload_frames(cep)
    a = eng.to_host(eng.estep(X, packed))
    b = eng.to_host(eng.estep(X, eng.pack(host.w, host.mu, host.invcov)))
    assert rel(a, b) < 1e-5


def test_em_split_device_loop_equals_host_loop(sk, eng):
    """EM_split with the model resident on the GPU (skm_gmm_mstep_pack in the loop) against the same schedule
    driven through the host M-step (_global_stats + _maximization): same model, same log-likelihood trace."""
    import copy
    g = golden("em_split.npz")
    chunks = {"f%d" % i: g["chunk%d" % i] for i in range(4)}
    fs = sk.FeaturesServer(features=chunks)
    dev = sk.Mixture()
    llk_dev = dev.EM_split(features_server=fs, feature_list=list(chunks.keys()), distrib_nb=8, num_thread=1)
    host = sk.Mixture()
    X = host._init(fs, list(chunks.keys()), 1)
    llk_host = []
    for it in (1, 2, 2):
        host._split_ditribution()
        accum = copy.deepcopy(host)
        for _ in range(it):
            accum._reset()
            # same conditioning of the float32 logits as the device loop (from the data's moments, Mixture._init)
            N, F, S, l = host._global_stats(X, host._pack(shift=host._em_shift, ref_var=host._em_ref_var))
            accum.w += N
            accum.mu += F
            accum.invcov += S
            llk_host.append(l / np.sum(accum.w))
            host._maximization(accum)
    assert rel(np.array(llk_dev), np.array(llk_host)) < 1e-9
    assert rel(dev.w, host.w) < 1e-7 and rel(dev.mu, host.mu) < 1e-7 and rel(dev.invcov, host.invcov) < 1e-7
    assert rel(dev.A, host.A) < 1e-7
