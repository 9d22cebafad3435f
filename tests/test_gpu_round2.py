"""GPU tests added in round 2: the parts of the drop-in surface that had no test (Mixture._expectation /
_expectation_list, save_partial + Mixture.read, StatServer.read_subset, VAD-selected E-steps of EM_split), the
Baum-Welch hand-off time-out (statistics become NaN, never silently wrong), float64 folding inside long
sessions, the device-resident StatServer, and the M-step workspace for D > R."""
import os

import numpy as np
import pytest

from oracle import sidekit_ref as orc
from conftest import golden

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def rel_large(a, b, frac=0.05):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = np.abs(b) >= frac * np.abs(b).max()
    return (np.abs(a - b)[m] / np.abs(b)[m]).max()


@pytest.fixture(scope="module")
def sk():
    import sidekit_amd
    return sidekit_amd


@pytest.fixture(scope="module")
def eng(sk):
    return sk.default_engine()


def _mixture(sk, gmm):
    m = sk.Mixture()
    m.w, m.mu, m.invcov = gmm.w.copy(), gmm.mu.copy(), gmm.invcov.copy()
    m.cov_var_ctl = 1.0 / m.invcov
    m._compute_all()
    return m


# ------------------------------------------------------------------ Mixture._expectation / _expectation_list
def test_expectation_and_expectation_list_dropin(sk):
    """mixture.py:537-588 through the drop-in methods themselves: statistics are ADDED to the accumulator,
    _expectation returns the summed log-likelihood, _expectation_list resets the accumulator, walks the files
    with keep_all_features = False (VAD-selected frames) and adds the log-likelihood to llk_acc."""
    import copy
    g = golden("estep_c40_d13.npz")
    gmm = orc.GMM(g["w"], g["mu"], g["invcov"])
    m = _mixture(sk, gmm)
    accum = copy.deepcopy(m)
    accum._reset()
    llk = m._expectation(accum, g["cep"])
    assert rel(accum.w, g["N"]) < TOL and rel(accum.mu, g["F"]) < TOL and rel(accum.invcov, g["S"]) < TOL
    assert abs(llk - float(g["llk"])) < 1e-5 * abs(float(g["llk"]))
    llk2 = m._expectation(accum, g["cep"][:100])                     # a second block adds
    rN, rF, rS, rl = orc.expectation(gmm, g["cep"][:100])
    assert rel(accum.w, g["N"] + rN) < TOL and rel(accum.mu, g["F"] + rF) < TOL
    assert abs(llk2 - rl) < 1e-5 * abs(rl)

    # two files, one with VAD labels; the server starts with keep_all_features = True
    cep = g["cep"]
    vad = np.random.RandomState(1).rand(400) > 0.4
    fs = sk.FeaturesServer(features={"a": cep[:400], "b": cep[400:]}, vad={"a": vad}, keep_all_features=True)
    accum.w += 123.0                                                   # must be reset by _expectation_list
    llk_acc = np.zeros(1)
    m._expectation_list(stat_acc=accum, feature_list=["a", "b"], feature_server=fs, llk_acc=llk_acc, num_thread=1)
    assert fs.keep_all_features is False
    sel = np.vstack([cep[:400][vad], cep[400:]])
    rN, rF, rS, rl = orc.expectation(gmm, sel)
    assert rel(accum.w, rN) < TOL and rel(accum.mu, rF) < TOL and rel(accum.invcov, rS) < TOL
    assert rel_large(accum.mu, rF) < TOL
    assert abs(llk_acc[0] - rl) < 1e-5 * abs(rl)


def test_em_split_vad_labels_vs_reference_golden(sk):
    """EM_split on a server WITH VAD labels and keep_all_features = True (the experiment's setting,
    model_interface.py:52): _init stacks every frame, the E-steps see the labelled frames only
    (mixture.py:585), and the flag is left False -- golden from the real reference."""
    g = golden("em_split_vad.npz")
    fs = sk.FeaturesServer(features={"f%d" % i: g["chunk%d" % i] for i in range(3)},
                           vad={"f%d" % i: g["vad%d" % i] for i in range(3)}, keep_all_features=True)
    ubm = sk.Mixture()
    llk = ubm.EM_split(features_server=fs, feature_list=["f0", "f1", "f2"], distrib_nb=4, num_thread=1)
    assert fs.keep_all_features is False
    assert len(llk) == 3
    assert rel(np.array(llk), g["llk"]) < TOL
    assert rel(ubm.w, g["w"]) < TOL and rel(ubm.mu, g["mu"]) < TOL
    assert rel(1.0 / ubm.invcov, 1.0 / g["invcov"]) < TOL
    assert rel(ubm.cov_var_ctl, g["ctl"]) < 1e-6


def test_em_split_save_partial_and_mixture_read(sk, tmp_path):
    """save_partial writes the model before every split (mixture.py:678-679) under '<name>_<C>g'; Mixture.read /
    Mixture(file) give the same parameters back and the result equals the run without save_partial."""
    g = golden("em_split.npz")
    chunks = {"f%d" % i: g["chunk%d" % i] for i in range(4)}
    fs = sk.FeaturesServer(features=chunks)
    base = str(tmp_path / "ubm")
    ubm = sk.Mixture()
    llk = ubm.EM_split(features_server=fs, feature_list=list(chunks), distrib_nb=8, save_partial=True, output_file_name=base)
    assert rel(np.array(llk), g["llk"]) < TOL
    ext = ".h5" if sk.mixture._has_h5py() else ".npz"
    for c in (1, 2, 4):
        assert os.path.exists("%s_%dg%s" % (base, c, ext))
    one = sk.Mixture("%s_1g%s" % (base, ext))
    assert one.distrib_nb() == 1 and one.validate()
    assert rel(one.mu, g["i_mu"]) < 1e-6 and rel(one.invcov, g["i_invcov"]) < 1e-6 and rel(one.cov_var_ctl, g["i_ctl"]) < 1e-6
    four = sk.Mixture()
    four.read("%s_4g%s" % (base, ext))
    assert four.distrib_nb() == 4 and four.validate() and abs(four.w.sum() - 1.0) < 1e-12
    out = str(tmp_path / ("final" + ext))
    ubm.write(out)
    back = sk.Mixture(out)
    for k in ("w", "mu", "invcov", "cov_var_ctl", "cst", "det"):
        np.testing.assert_array_equal(getattr(back, k), getattr(ubm, k))
    np.testing.assert_array_equal(back.A, ubm.A)


def test_statserver_write_read_subset(sk, tmp_path):
    """statserver.py:377-476, 1870-1913: float32 storage, read, read_subset by index list and by IdMap."""
    rs = np.random.RandomState(2)
    S, C, D = 9, 4, 3
    ss = sk.StatServer()
    ss.modelset = np.array(["m%d" % (i % 3) for i in range(S)])
    ss.segset = np.array(["s%02d" % i for i in range(S)])
    ss.start = np.empty(S, "|O")
    ss.stop = np.empty(S, "|O")
    ss.start[2], ss.stop[2] = 5, 90
    ss.stat0 = rs.rand(S, C) * 100
    ss.stat1 = rs.randn(S, C * D) * 100
    f = str(tmp_path / "stats.npz")
    ss.write(f)
    full = sk.StatServer.read(f)
    assert full.validate() and list(full.segset) == list(ss.segset)
    np.testing.assert_array_equal(full.stat1, ss.stat1.astype(np.float32).astype(np.float64))   # the float32 contract
    assert full.start[2] == 5 and full.stop[2] == 90 and full.start[0] is None
    sub = sk.StatServer.read_subset(f, [7, 1, 4])
    assert list(sub.segset) == ["s07", "s01", "s04"] and list(sub.modelset) == ["m1", "m1", "m1"]
    np.testing.assert_array_equal(sub.stat0, full.stat0[[7, 1, 4]])
    idm = sk.IdMap().set(["x", "y"], ["s08", "s02"])
    sub2 = sk.StatServer(f, index=idm)
    assert list(sub2.segset) == ["s08", "s02"] and sub2.stop[1] == 90
    np.testing.assert_array_equal(sub2.stat1, full.stat1[[8, 2]])


# ------------------------------------------------------------------ Baum-Welch: time-out, long sessions
@pytest.mark.parametrize("C", [512, 2048])
def test_bw_stats_handoff_timeout_poisons_the_statistics(eng, C):
    """A team member that never raises its hand-off flag (SKM_DEBUG_TEAM_DROP=1, tests only) makes every bounded
    poll of its team run out: the kernel raises its error flag, carries on, and skm_bw_stats turns stat0 / stat1
    into NaN on the stream instead of returning stale numbers."""
    D = 20
    gmm = orc.random_gmm(C, D, seed=4)
    lens = [64, 100, 33, 257]
    frames = orc.synth_frames(sum(lens), dim=D, seed=5)
    off = np.concatenate([[0], np.cumsum(lens)])
    X = eng.upload_frames(frames)
    packed = eng.pack(gmm.w, gmm.mu, gmm.invcov)
    good0, good1 = eng.bw_stats(X, off, packed)
    assert np.isfinite(eng.to_host(good0)).all()
    os.environ["SKM_DEBUG_TEAM_DROP"] = "1"
    try:
        s0, s1 = eng.bw_stats(X, off, packed)
        s0, s1 = eng.to_host(s0), eng.to_host(s1)
    finally:
        del os.environ["SKM_DEBUG_TEAM_DROP"]
    assert np.isnan(s0).all() and np.isnan(s1).all()
    again0, again1 = eng.bw_stats(X, off, packed)                      # the next call is clean again
    np.testing.assert_array_equal(eng.to_host(again0), eng.to_host(good0))
    np.testing.assert_array_equal(eng.to_host(again1), eng.to_host(good1))


@pytest.mark.parametrize("C,D", [(512, 20), (96, 20)])
def test_bw_stats_many_long_sessions_fold_into_float64(eng, C, D):
    """Sessions far longer than one float32 accumulation run (8192 frames): 1024 sessions x 100 000 frames, four per
    team of the C = 512 kernel, too many for the host-side split.  Each session is a 1000-frame block repeated 100
    times, so its exact statistics are 100 x the block's (oracle, float64); the float32 accumulators are folded
    into the float64 rows every 256 tiles inside the kernel.  Element-wise 1e-4 on the large entries."""
    torch = eng.torch
    nblk, L, reps, S = 8, 1000, 100, (1024 if C == 512 else 64)
    gmm = orc.random_gmm(C, D, seed=8)
    blocks = [orc.synth_frames(L, dim=D, clusters=6, seed=40 + i) for i in range(nblk)]
    ref = [orc.accumulate_stat(gmm, [b.astype(np.float64)]) for b in blocks]
    dev_blocks = [eng.upload_frames(b) for b in blocks]
    X = torch.cat([dev_blocks[s % nblk].repeat(reps, 1) for s in range(S)], dim=0)
    seg = torch.arange(0, S + 1, dtype=torch.int64, device=eng.device) * (L * reps)
    s0, s1 = eng.bw_stats(X, seg, eng.pack(gmm.w, gmm.mu, gmm.invcov), split_long=False)
    for s in (0, 1, 5, S // 2 + 3, S - 1):
        r0, r1 = ref[s % nblk]
        g0, g1 = s0[s].cpu().numpy(), s1[s].cpu().numpy()
        assert rel(g0, reps * r0[0]) < TOL and rel(g1, reps * r1[0]) < TOL
        assert rel_large(g0, reps * r0[0]) < TOL and rel_large(g1, reps * r1[0]) < TOL
    rows = s0.sum(1).cpu().numpy()
    assert np.abs(rows - L * reps).max() < 2e-5 * L * reps
    # all sessions built from the same block carry the same statistics bit for bit within a launch? (no: teams fold at
    # the same tile positions of every session, so yes)
    np.testing.assert_array_equal(s1[0].cpu().numpy(), s1[nblk].cpu().numpy())


# ------------------------------------------------------------------ StatServer -> TV without a host round trip
def test_accumulate_stat_keep_on_device_feeds_tv(sk, eng):
    """accumulate_stat(keep_on_device=True) leaves this rank's rows in HBM rounded to float32 as a file would
    store them; total_variability / extract_ivectors given that StatServer start from them and give the results
    of the file route bit for bit."""
    g = golden("accumulate_stat.npz")
    gt = golden("tv.npz")
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    names = ["s%03d" % i for i in range(len(g["lens"]))]
    fs = sk.FeaturesServer(features={n: g["frames"][off[i]:off[i + 1]] for i, n in enumerate(names)})
    ubm = _mixture(sk, orc.GMM(g["w"], g["mu"], g["invcov"]))
    ss = sk.StatServer(sk.IdMap().set(["spk%d" % (i % 5) for i in range(len(names))], names), ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(len(names)), keep_on_device=True)
    res = ss.resident_rows()
    assert res is not None and res[1].is_cuda and res[2].dtype == eng.torch.float32
    np.testing.assert_array_equal(res[2].cpu().numpy(), ss.stat1.astype(np.float32))
    fa1 = sk.FactorAnalyser()
    fa1.total_variability_single(ss, ubm, 6, nb_iter=2, min_div=True, tv_init=gt["tv_init"].copy())
    plain = sk.StatServer(sk.IdMap().set(list(ss.modelset), names), ubm=ubm)
    plain.stat0, plain.stat1 = ss.stat0.copy(), ss.stat1.copy()
    assert plain.resident_rows() is None
    fa2 = sk.FactorAnalyser()
    fa2.total_variability_single(plain, ubm, 6, nb_iter=2, min_div=True, tv_init=gt["tv_init"].copy())
    np.testing.assert_array_equal(fa1.F, fa2.F)
    iv1 = fa1.extract_ivectors(ubm, ss)
    iv2 = fa2.extract_ivectors(ubm, plain)
    np.testing.assert_array_equal(iv1.stat1, iv2.stat1)
    # replacing the host statistics invalidates the device copy
    ss.stat1 = ss.stat1 * 2.0
    assert ss.resident_rows() is None
    # ... and so does editing them IN PLACE (same arrays, other numbers): row edits and whole-array updates
    for edit in ("row", "all", "stat0"):
        ss2 = sk.StatServer(sk.IdMap().set(["spk%d" % (i % 5) for i in range(len(names))], names), ubm=ubm)
        ss2.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(len(names)), keep_on_device=True)
        assert ss2.resident_rows() is not None
        if edit == "row":
            ss2.stat1[3] = 0.5 * ss2.stat1[3]
        elif edit == "all":
            ss2.stat1 -= 0.25
        else:
            ss2.stat0 *= 1.5
        assert ss2.resident_rows() is None, edit


# ------------------------------------------------------------------ TV
def test_tv_multi_block_rank_vs_reference_golden(sk):
    """R = 40 (two 32-wide blocks of the device Cholesky, full rank, well conditioned), C=64, D=16, 120 sessions,
    3 EM iterations and extraction with uncertainty against the REAL reference (tools/gen_golden.py round2)."""
    g = golden("tv_big.npz")
    C, D, R, S = [int(v) for v in g["shape"]]
    ubm = _mixture(sk, orc.GMM(g["w"], g["mu"], g["invcov"]))
    ss = sk.StatServer()
    ss.modelset = np.array(["m%d" % (i % 12) for i in range(S)])
    ss.segset = np.array(["s%03d" % i for i in range(S)])
    ss.start = np.empty(S, "|O")
    ss.stop = np.empty(S, "|O")
    ss.stat0 = g["stat0_f32"].astype(np.float64)
    ss.stat1 = g["stat1_f32"].astype(np.float64)
    tv_init = 0.1 * np.random.RandomState(7).randn(C * D, R)
    fa = sk.FactorAnalyser()
    fa.total_variability_single(ss, ubm, R, nb_iter=3, min_div=True, tv_init=tv_init.copy(), batch_size=30)
    assert rel(fa.F, g["F_it3"]) < TOL, rel(fa.F, g["F_it3"])
    assert rel_large(fa.F, g["F_it3"]) < TOL
    fa.F = g["F_it3"].copy()
    ivs, sig = fa.extract_ivectors(ubm, ss, uncertainty=True)
    cos = (ivs.stat1 * g["iv"]).sum(1) / np.linalg.norm(ivs.stat1, axis=1) / np.linalg.norm(g["iv"], axis=1)
    assert cos.min() >= 0.9999
    assert rel(ivs.stat1, g["iv"]) < TOL and rel(sig, g["iv_sigma"]) < TOL
    ivs2 = fa.extract_ivectors(ubm, ss)                                # solve path (no uncertainty)
    assert rel(ivs2.stat1, g["iv"]) < TOL


def test_tv_mstep_feature_dimension_larger_than_rank(eng):
    """D > 1.5 R + 64: the [batch][D][R] product of the M-step has its own workspace region (it used to alias the
    R x R scratch and overran it).  Against numpy.linalg.solve per component."""
    torch = eng.torch
    C, D, R = 5, 80, 8
    rs = np.random.RandomState(3)
    A = np.zeros((C + 1, R, R))
    for c in range(C + 1):
        M = rs.randn(R, R)
        A[c] = M.dot(M.T) + R * np.eye(R)
    Cacc = rs.randn(R, C * D)
    ref = np.vstack([np.linalg.solve(A[c], Cacc[:, c * D:(c + 1) * D]).T for c in range(C)])
    guard = torch.full((4096,), 7.0, dtype=torch.float64, device=eng.device)   # lives behind the workspace growth
    F = eng.tv_mstep(torch.from_numpy(A.reshape(C + 1, R * R).copy()).to(eng.device),
                     torch.from_numpy(Cacc.copy()).to(eng.device), C, D, R)
    assert rel(eng.to_host(F), ref) < 1e-10
    assert float(guard.sum().item()) == 7.0 * 4096
    need = eng.lib.skm_tv_workspace_bytes(eng.ctx, C, D, R, 0, C, 0)
    small = eng.lib.skm_tv_workspace_bytes(eng.ctx, C, R, R, 0, C, 0)
    assert need >= small + C * D * R * 8


# ------------------------------------------------------------------ cosine scoring of a StatServer against itself
def test_cosine_scoring_same_statserver_twice(sk):
    """enroll is test (iv_scoring.py:84-93 copies and aligns each side on its own): several sessions per model,
    modelset order different from segset order -- rows of the score matrix follow the Ndx's models (first session
    of each), columns its segments."""
    rs = np.random.RandomState(4)
    S, R = 7, 12
    ss = sk.StatServer()
    ss.modelset = np.array(["b", "a", "b", "c", "a", "c", "b"])
    ss.segset = np.array(["s%d" % i for i in (3, 0, 6, 1, 5, 2, 4)])
    ss.start = np.empty(S, "|O")
    ss.stop = np.empty(S, "|O")
    ss.stat0 = np.ones((S, 1))
    ss.stat1 = rs.randn(S, R)
    ndx = sk.Ndx()
    ndx.modelset = np.array(["a", "b", "c"])
    ndx.segset = np.array(["s%d" % i for i in range(7)])
    ndx.trialmask = np.ones((3, 7), dtype=bool)
    sc = sk.cosine_scoring(ss, ss, ndx)
    first = {m: np.argwhere(ss.modelset == m)[0][0] for m in ("a", "b", "c")}
    E = np.vstack([ss.stat1[first[m]] for m in sc.modelset])
    Tm = np.vstack([ss.stat1[np.argwhere(ss.segset == s)[0][0]] for s in sc.segset])
    ref = orc.cosine_scoring(E, Tm)
    assert sc.scoremat.shape == (3, 7)
    assert np.abs(sc.scoremat - ref).max() < 1e-12
