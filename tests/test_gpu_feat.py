"""GPU parity tests of the device feature post-processing (skm_feat_postproc through the FeaturesServer
drop-in) against the golden vectors of the real reference's FeaturesServer.post_processing and against the
CPU oracle.

Tolerance: the kernel computes in float64 like the reference; only the summation order inside the filters
and the one-pass (shifted) variance differ, i.e. ~1e-15 relative on the static block.  Deltas are ROUNDED TO
FLOAT32 by the reference (frontend/features.py:159), so a last-bit difference before the rounding can move a
delta by one float32 ulp (6e-8 relative): 1e-6 normwise on the float64 outputs, 2e-6 on float32 outputs.
"""
import numpy as np
import pytest

from oracle import sidekit_ref as orc
from conftest import golden
from test_oracle_golden import _pp_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sk():
    import sidekit_amd
    return sidekit_amd


def check(a, b, tol):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.array_equal(np.isposinf(a), np.isposinf(b)) and np.array_equal(np.isneginf(a), np.isneginf(b))
    m = np.isfinite(b)
    if m.any():
        # per output column: the three blocks (static, delta, double delta) differ in scale
        for j in range(b.shape[1]):
            mj = m[:, j]
            if mj.any():
                scale = max(np.abs(b[mj, j]).max(), 1e-300)
                assert np.abs(a[mj, j] - b[mj, j]).max() / scale <= tol, (j, np.abs(a[mj, j] - b[mj, j]).max() / scale)


def server(sk, kw):
    return sk.FeaturesServer(mask=kw["mask"], feat_norm=kw["feat_norm"], delta=kw["delta"],
                             double_delta=kw["double_delta"], rasta=kw["rasta"],
                             keep_all_features=kw["keep_all_features"])


def test_post_processing_vs_reference_golden(sk):
    n = 0
    for i, x, label, kw, feat, lab in _pp_cases():
        fs = server(sk, kw)
        f, l = fs.post_processing(x.copy(), label.copy(), kw["global_mean"], kw["global_std"])
        assert f.dtype == np.float64 and l.dtype == bool
        assert np.array_equal(l, lab), i
        check(f, feat, 1e-6)
        n += 1
    assert n == 11


@pytest.mark.parametrize("keep_all", [True, False])
@pytest.mark.parametrize("in_dtype", [np.float32, np.float64])
def test_batch_of_ragged_sessions_vs_oracle(sk, keep_all, in_dtype):
    """One launch over many sessions of different length (3 ... 2000 frames = 1 ... 8 chunks of 256 with every
    boundary case of the chunk hand-over, one without speech, one fully speech), experiment configuration (RASTA + delta + double delta + cmvn), float32 frames out."""
    rs = np.random.RandomState(5)
    lens = [3, 4, 5, 9, 64, 65, 257, 700, 31, 100, 256, 512, 1030, 2000, 262, 263] + [int(v) for v in rs.randint(10, 300, size=30)]
    D0 = 20
    fs = sk.FeaturesServer(feat_norm="cmvn", delta=True, double_delta=True, rasta=True, keep_all_features=keep_all)
    shows, ref_f, ref_l = [], [], []
    for i, L in enumerate(lens):
        x = (2.0 * rs.randn(L, D0) + 4.0 * rs.randn(1, D0) + np.cumsum(0.2 * rs.randn(L, D0), axis=0)).astype(in_dtype)
        lab = rs.rand(L) > 0.3
        lab[2] = lab[2] or (i % 2 == 0)
        if i == 4:
            lab[:] = False
        if i == 5:
            lab[:] = True
        name = "s%02d" % i
        fs.add(name, x, lab)
        shows.append(name)
        with np.errstate(all="ignore"):
            f, l = orc.post_processing(x.astype(np.float64), lab, rasta=True, delta=True, double_delta=True,
                                       feat_norm="cmvn", keep_all_features=keep_all)
        ref_f.append(f)
        ref_l.append(l)
    X, seg_off, lab_out = fs.load_batch(shows)
    import torch
    assert X.dtype == torch.float32 and X.is_cuda
    ref = np.vstack(ref_f)
    assert X.shape == ref.shape
    assert np.array_equal(np.diff(seg_off), [f.shape[0] for f in ref_f])
    assert np.array_equal(lab_out.cpu().numpy().astype(bool), np.concatenate(ref_l))
    check(X.cpu().numpy(), ref, 2e-6)
    # bit-identical from run to run (no atomics)
    X2, _, _ = fs.load_batch(shows)
    assert torch.equal(X, X2) or np.array_equal(X.cpu().numpy(), X2.cpu().numpy(), equal_nan=True)
    # stack_features_parallel is the same batch path (what Mixture.EM_split calls)
    Xs = fs.stack_features_parallel(shows)
    assert np.array_equal(Xs.cpu().numpy(), X.cpu().numpy(), equal_nan=True)


@pytest.mark.parametrize("opts", [
    dict(rasta=False, delta=True, double_delta=True, feat_norm="cms"),
    dict(rasta=False, delta=True, double_delta=False, feat_norm=None),
    dict(rasta=True, delta=False, double_delta=False, feat_norm="cmvn"),
    dict(rasta=False, delta=False, double_delta=True, feat_norm="cmvn"),       # double delta needs delta: ignored
    dict(rasta=False, delta=False, double_delta=False, feat_norm="cmvn", mask="[0,2-4]"),
])
def test_option_combinations_vs_oracle(sk, opts):
    rs = np.random.RandomState(8)
    fs = sk.FeaturesServer(keep_all_features=False, **opts)
    okw = dict(opts)
    mask = okw.pop("mask", None)
    for T, D0 in [(1, 5), (2, 5), (7, 6), (200, 13), (1000, 44)]:
        if opts["rasta"] and T < 3:
            continue
        x = rs.randn(T, D0) * rs.uniform(0.5, 3.0, size=D0) + 3.0 * rs.randn(D0)
        lab = rs.rand(T) > 0.4
        lab[0] = True
        with np.errstate(all="ignore"):
            f_ref, l_ref = orc.post_processing(x, lab, keep_all_features=False,
                                               mask=sk.package.features_server.parse_mask(mask) if mask else None, **okw)
        f, l = fs.post_processing(x.copy(), lab.copy())
        assert np.array_equal(l, l_ref)
        check(f, f_ref, 1e-6)


def test_errors_match_reference(sk):
    fs = sk.FeaturesServer(rasta=True)
    with pytest.raises(IndexError):                       # features_server.py:322 on a 2-frame session
        fs.post_processing(np.zeros((2, 4)), np.ones(2, dtype=bool))
    fs = sk.FeaturesServer(feat_norm="stg")
    with pytest.raises(TypeError):                        # frontend/normfeat.py:166 (float slice index)
        fs.post_processing(np.zeros((400, 4)), np.ones(400, dtype=bool))
    fs = sk.FeaturesServer(delta=True, delta_filter=[.2, .1, 0, -.1, -.2])
    with pytest.raises(ValueError):                       # frontend/features.py:167: 5-tap result does not fit
        fs.post_processing(np.zeros((40, 4)), np.ones(40, dtype=bool))
    with pytest.raises(Exception):
        sk.FeaturesServer(mask="[1-a]")


def test_accumulate_stat_from_raw_frames_on_device(sk):
    """Raw cepstra + VAD in, Baum-Welch statistics out, frames never on the host in between: equals the oracle's
    post_processing (all frames kept, as accumulate_stat forces) -> cep[vad] -> accumulate_stat."""
    rs = np.random.RandomState(12)
    D0, C = 20, 64
    gmm = orc.random_gmm(C, 3 * D0, seed=4)
    ubm = sk.Mixture()
    ubm.w, ubm.mu, ubm.invcov = gmm.w.copy(), gmm.mu.copy(), gmm.invcov.copy()
    ubm.cov_var_ctl = 1.0 / ubm.invcov
    ubm._compute_all()
    fs = sk.FeaturesServer(feat_norm="cmvn", delta=True, double_delta=True, rasta=True, keep_all_features=True)
    names, sessions = [], []
    for i in range(12):
        L = int(rs.randint(30, 400))
        x = orc.synth_frames(L, dim=D0, clusters=6, seed=50 + i, dtype=np.float64) * 2.0 + 1.0
        lab = rs.rand(L) > 0.3
        fs.add("u%02d" % i, x, lab)
        names.append("u%02d" % i)
        f, l = orc.post_processing(x, lab, rasta=True, delta=True, double_delta=True, feat_norm="cmvn",
                                   keep_all_features=True)
        sessions.append(f[l])
    idmap = sk.IdMap()
    idmap.leftids = np.array(["spk%d" % (i % 3) for i in range(12)])
    idmap.rightids = np.array(names)
    idmap.start = np.empty(12, "|O")
    idmap.stop = np.empty(12, "|O")
    ss = sk.StatServer(idmap, ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(12))
    s0, s1 = orc.accumulate_stat(gmm, sessions)
    assert np.abs(ss.stat0 - s0).max() / np.abs(s0).max() < 1e-4
    assert np.abs(ss.stat1 - s1).max() / np.abs(s1).max() < 1e-4
    # per-session path (explicit start/stop force the load() loop): same statistics
    ss2 = sk.StatServer(idmap, ubm=ubm)
    ss2.start[:] = 0
    ss2.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(12))
    assert np.abs(ss2.stat0 - ss.stat0).max() / np.abs(s0).max() < 1e-6
    assert np.abs(ss2.stat1 - ss.stat1).max() / np.abs(s1).max() < 1e-6


def _sliding_cases():
    g = golden("post_processing_sliding.npz")
    for i in range(int(g["n_cases"])):
        o = g["case%d_opt" % i]
        kw = dict(mask=None, rasta=bool(o[0]), delta=bool(o[1]), double_delta=bool(o[2]),
                  feat_norm={3: "cms_sliding", 4: "cmvn_sliding"}[int(o[3])], keep_all_features=bool(o[4]))
        yield i, g["case%d_x" % i], g["case%d_label" % i], kw, g["case%d_feat" % i], g["case%d_lab" % i]


def test_sliding_normalisation_vs_reference_golden(sk):
    """'cmvn_sliding' / 'cms_sliding' (features_server.py:282-287 -> normfeat.py:194-229) of the real reference:
    long sessions (rolling statistics over 301 labelled frames), sessions with at most 301 labelled frames
    (whole-session fallback; 301 and 302 labelled frames sit on either side of the switch)."""
    n = 0
    for i, x, label, kw, feat, lab in _sliding_cases():
        f, l = server(sk, kw).post_processing(x.copy(), label.copy())
        assert f.dtype == np.float64 and l.dtype == bool
        assert np.array_equal(l, lab), i
        check(f, feat, 1e-8)
        n += 1
    assert n == 7


@pytest.mark.parametrize("keep_all", [True, False])
@pytest.mark.parametrize("norm", ["cmvn_sliding", "cms_sliding"])
def test_sliding_normalisation_batch_vs_oracle(sk, keep_all, norm):
    """A batch of long, short, unlabelled and chunk-boundary sessions in one launch (load_batch): every session equals
    the oracle run on it alone."""
    rs = np.random.RandomState(21)
    lens = [1500, 40, 302, 701, 3, 2048, 429]
    kw = dict(mask=None, rasta=False, delta=True, double_delta=False, feat_norm=norm, keep_all_features=keep_all)
    fs = server(sk, kw)
    want, want_lab = [], []
    for k, T in enumerate(lens):
        x = 2.0 * rs.randn(T, 6) + 4.0 * rs.randn(1, 6) + np.cumsum(0.2 * rs.randn(T, 6), axis=0)
        lab = rs.rand(T) > (0.25 if k != 3 else 0.0)
        if k == 1:
            lab[:] = True
        fs.add("s%d" % k, x, lab)
        f, l = orc.post_processing(x.copy(), lab.copy(), delta=True, feat_norm=norm, keep_all_features=keep_all)
        want.append(f)
        want_lab.append(l)
    X, off, labels = fs.load_batch(["s%d" % k for k in range(len(lens))])
    X, labels = X.cpu().numpy(), labels.cpu().numpy().astype(bool)
    assert off[-1] == X.shape[0] == sum(w.shape[0] for w in want)
    for k in range(len(lens)):
        assert off[k + 1] - off[k] == want[k].shape[0], k
        assert np.array_equal(labels[off[k]:off[k + 1]], want_lab[k])
        check(X[off[k]:off[k + 1]], want[k].astype(np.float32), 2e-6)


def test_sliding_normalisation_without_any_labelled_frame(sk):
    """keep_all_features=False and no speech at all: an empty result, no launch on a NULL buffer."""
    fs = server(sk, dict(mask=None, rasta=False, delta=False, double_delta=False, feat_norm="cmvn_sliding",
                         keep_all_features=False))
    x = np.random.RandomState(1).randn(400, 3)
    f, l = fs.post_processing(x, np.zeros(400, dtype=bool))
    assert f.shape == (0, 3) and l.shape == (0,)


@pytest.mark.parametrize("seed", range(16))
def test_random_configurations_vs_oracle(sk, seed):
    """Random option sets, column counts and ragged batches (sessions of 3 ... 2500 frames, some without speech, some
    all speech) through load_batch: every session equals the oracle run on it alone."""
    rs = np.random.RandomState(1000 + seed)
    norm = [None, "cms", "cmvn", "cmvn_sliding", "cms_sliding"][rs.randint(5)]
    delta = bool(rs.randint(2))
    kw = dict(mask=None, rasta=bool(rs.randint(2)), delta=delta, double_delta=delta and bool(rs.randint(2)),
              feat_norm=norm, keep_all_features=bool(rs.randint(2)))
    D0 = int(rs.choice([1, 2, 5, 13, 20, 33, 44, 70]))
    in_dtype = [np.float32, np.float64][rs.randint(2)]
    fs = server(sk, kw)
    names, want, want_lab = [], [], []
    for k in range(int(rs.randint(3, 13))):
        kind = rs.randint(3)
        T = int(rs.randint(3, 11) if kind == 0 else rs.randint(30, 400) if kind == 1 else rs.randint(500, 2500))
        x = (2.0 * rs.randn(T, D0) + 4.0 * rs.randn(1, D0) + np.cumsum(0.2 * rs.randn(T, D0), axis=0)).astype(in_dtype)
        p = rs.rand()
        lab = np.ones(T, dtype=bool) if p < 0.15 else np.zeros(T, dtype=bool) if p < 0.25 else rs.rand(T) > rs.uniform(0.1, 0.8)
        fs.add("s%d" % k, x, lab)
        names.append("s%d" % k)
        with np.errstate(all="ignore"):
            f, l = orc.post_processing(x.astype(np.float64), lab.copy(), rasta=kw["rasta"], delta=kw["delta"],
                                       double_delta=kw["double_delta"], feat_norm=norm,
                                       keep_all_features=kw["keep_all_features"])
        want.append(f)
        want_lab.append(l)
    X, off, labels = fs.load_batch(names)
    X, labels = X.cpu().numpy(), labels.cpu().numpy().astype(bool)
    assert off[-1] == X.shape[0] == sum(w.shape[0] for w in want), kw
    for k in range(len(names)):
        assert off[k + 1] - off[k] == want[k].shape[0], (k, kw)
        assert np.array_equal(labels[off[k]:off[k + 1]], want_lab[k]), (k, kw)
        if want[k].shape[0]:
            check(X[off[k]:off[k + 1]], want[k].astype(np.float32), 2e-6)
