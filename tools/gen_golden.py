#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (runs only in the build container).

The reference (/root/reference, SIDEKIT 1.3.2 fork) is imported read-only with
`sys.dont_write_bytecode`, exercised on small seeded inputs with `num_thread=1`, and only
input/output ARRAYS are written to tests/golden/*.npz.  No reference source, bytecode or
derivative travels with the repo.  The recipe follows SURVEY.md Appendix A:

* `h5py` is not installed: a dict-of-ndarrays stand-in is registered in sys.modules so that
  `import sidekit` succeeds and the file-based drivers (StatServer.write, total_variability,
  extract_ivectors) can round-trip through "files";
* matplotlib 3.10 rejects `matplotlib.use(..., warn=False)` in bosaris/detplot.py:39.

Usage:  SIDEKIT=libsvm=false python3 -W ignore tools/gen_golden.py
"""
import copy
import os
import sys
import types
import pickle
import tempfile

os.environ.setdefault("SIDEKIT", "libsvm=false")
sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


# ------------------------------------------------------------------ h5py stand-in
class _DS(object):
    def __init__(self, arr):
        self.arr = np.asarray(arr)

    @property
    def value(self):
        return self.arr

    @property
    def shape(self):
        return self.arr.shape

    def __getitem__(self, k):
        return self.arr[k]

    def __setitem__(self, k, v):
        self.arr[k] = v

    def resize(self, shape):
        new = np.zeros(shape, dtype=self.arr.dtype)
        sl = tuple(slice(0, min(a, b)) for a, b in zip(self.arr.shape, shape))
        new[sl] = self.arr[sl]
        self.arr = new


class _File(object):
    def __init__(self, name, mode="r"):
        self.name, self.mode = name, mode
        self.d = {}
        if mode in ("r", "a", "r+") and os.path.exists(name):
            with open(name, "rb") as f:
                self.d = {k: _DS(v) for k, v in pickle.load(f).items()}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        if self.mode != "r":
            with open(self.name, "wb") as f:
                pickle.dump({k: v.arr for k, v in self.d.items()}, f)

    def create_dataset(self, name, shape=None, dtype=None, data=None, **kw):
        arr = np.zeros(shape, dtype=dtype) if data is None else np.array(data)
        self.d[name] = _DS(arr)
        return self.d[name]

    def __getitem__(self, k):
        return self.d[k]

    def get(self, k):
        return self.d.get(k)

    def __contains__(self, k):
        return k in self.d

    def keys(self):
        return self.d.keys()


def import_reference():
    fake = types.ModuleType("h5py")
    fake.File = _File
    sys.modules["h5py"] = fake
    import matplotlib
    matplotlib.use = lambda *a, **k: None
    sys.path.insert(0, REF)
    import sidekit
    return sidekit


sys.path.insert(0, ROOT)
from oracle import sidekit_ref as orc  # only for the shared synthetic-data helpers


def make_ubm(sidekit, w, mu, invcov):
    m = sidekit.Mixture()
    m.w = np.array(w, dtype="d")
    m.mu = np.array(mu, dtype="d")
    m.invcov = np.array(invcov, dtype="d")
    m.cov_var_ctl = 1.0 / m.invcov
    m.cst = np.zeros(m.w.shape)
    m.det = np.zeros(m.w.shape)
    m._compute_all()
    return m


def mem_server(sidekit, table):
    """In-memory FeaturesServer (must subclass: isinstance assert at statserver.py:623)."""
    class Mem(sidekit.FeaturesServer):
        def __init__(self):
            super().__init__(dataset_list=["cep"])
            self.table = table

        def load(self, show, channel=0, input_feature_filename=None, label=None, start=None, stop=None):
            x = self.table[show]
            return x, np.ones(x.shape[0], dtype=bool)

        def stack_features_parallel(self, show_list, num_thread=1, **kw):
            return np.vstack([self.table[s] for s in show_list])
    return Mem()


def gen_estep(sidekit, tag, C, D, T, seed):
    g = orc.random_gmm(C, D, seed=seed)
    cep = orc.synth_frames(T, dim=D, clusters=max(4, C // 2), seed=seed + 1, dtype=np.float64)
    ubm = make_ubm(sidekit, g.w, g.mu, g.invcov)
    lp = ubm.compute_log_posterior_probabilities(cep)
    pp, llk = sidekit.mixture.sum_log_probabilities(lp)
    pp2, log_lk = sidekit.statserver.sum_log_probabilities(lp)
    import copy
    accum = copy.deepcopy(ubm)
    accum._reset()
    llk2 = ubm._expectation(accum, cep)
    out = dict(w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, cst=ubm.cst, det=ubm.det, A=ubm.A, cep=cep,
               N=accum.w.copy(), F=accum.mu.copy(), S=accum.invcov.copy(), llk=np.float64(llk2),
               log_lk=log_lk)
    if T * C <= 40000:
        out["lp"] = lp
        out["pp"] = pp
    new = copy.deepcopy(ubm)
    new._maximization(accum)
    out.update(m_w=new.w, m_mu=new.mu, m_invcov=new.invcov, m_cst=new.cst, m_det=new.det, m_A=new.A)
    sp = copy.deepcopy(new)
    sp._split_ditribution()
    out.update(s_w=sp.w, s_mu=sp.mu, s_invcov=sp.invcov, s_ctl=sp.cov_var_ctl, s_A=sp.A)
    np.savez_compressed(os.path.join(OUT, "estep_%s.npz" % tag), **out)
    print("estep", tag, "llk", llk2)


def gen_em_split(sidekit):
    rs = np.random.RandomState(11)
    D, K = 6, 8
    centers = rs.randn(K, D) * 6.0
    chunks = {}
    for i in range(4):
        lab = rs.randint(0, K, size=500 + 37 * i)
        chunks["f%d" % i] = centers[lab] + 0.5 * rs.randn(lab.shape[0], D)
    fs = mem_server(sidekit, chunks)
    ubm = sidekit.Mixture()
    llk = ubm.EM_split(features_server=fs, feature_list=list(chunks.keys()), distrib_nb=8, num_thread=1,
                       save_partial=False)
    out = {"chunk%d" % i: chunks["f%d" % i] for i in range(4)}
    out.update(llk=np.array(llk), w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, ctl=ubm.cov_var_ctl, A=ubm.A)
    # init only
    u0 = sidekit.Mixture()
    u0._init(fs, list(chunks.keys()), 1)
    out.update(i_w=u0.w, i_mu=u0.mu, i_invcov=u0.invcov, i_ctl=u0.cov_var_ctl, i_A=u0.A)
    np.savez_compressed(os.path.join(OUT, "em_split.npz"), **out)
    print("em_split llk", llk[-3:], "iters", len(llk))


def gen_em_uniform(sidekit):
    rs = np.random.RandomState(21)
    D, K = 5, 4
    centers = rs.randn(K, D) * 4.0
    lab = rs.randint(0, K, size=1500)
    cep = centers[lab] + 0.7 * rs.randn(1500, D)
    ubm = sidekit.Mixture()
    llk = ubm.EM_uniform(cep, 4, iteration_min=3, iteration_max=10, llk_gain=0.01, do_init=True)
    out = dict(cep=cep, llk=np.array(llk), w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, ctl=ubm.cov_var_ctl)
    u0 = sidekit.Mixture()
    u0._init_uniform(cep, 4)
    out.update(i_w=u0.w, i_mu=u0.mu, i_invcov=u0.invcov, i_ctl=u0.cov_var_ctl)
    np.savez_compressed(os.path.join(OUT, "em_uniform.npz"), **out)
    print("em_uniform llk", llk)


def gen_stats_and_tv(sidekit, tmp):
    C, D, R = 16, 10, 6
    g = orc.random_gmm(C, D, seed=31)
    ubm = make_ubm(sidekit, g.w, g.mu, g.invcov)
    rs = np.random.RandomState(32)
    lens = [1, 7, 33, 64, 100, 257] + list(rs.randint(40, 300, size=34))
    S = len(lens)
    table = {}
    for i, L in enumerate(lens):
        table["s%03d" % i] = orc.synth_frames(L, dim=D, clusters=8, seed=100 + i, dtype=np.float64)
    fs = mem_server(sidekit, table)
    idmap = sidekit.IdMap()
    idmap.leftids = np.array(["spk%d" % (i % 5) for i in range(S)])
    idmap.rightids = np.array(sorted(table.keys()))
    idmap.start = np.empty(S, "|O")
    idmap.stop = np.empty(S, "|O")
    ss = sidekit.StatServer(idmap, ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(S), num_thread=1)
    out = dict(w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, lens=np.array(lens),
               frames=np.vstack([table[k] for k in sorted(table.keys())]),
               stat0=ss.stat0.copy(), stat1=ss.stat1.copy())
    np.savez_compressed(os.path.join(OUT, "accumulate_stat.npz"), **out)
    print("accumulate_stat", ss.stat0.shape, ss.stat1.shape)

    # --- TV: statistics go through the float32 "HDF5" file like i-vector.py does
    stat_file = os.path.join(tmp, "stat.h5")
    ss.write(stat_file)
    s0_32 = ss.stat0.astype(np.float32)
    s1_32 = ss.stat1.astype(np.float32)
    tv_init = 0.1 * np.random.RandomState(7).randn(C * D, R)

    from sidekit.factor_analyser import e_on_batch
    a0, a1 = s0_32[:9].copy(), s1_32[:9].copy()
    e_h, e_hh = e_on_batch(a0, a1, ubm, tv_init.copy())
    tv = dict(w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, stat0_f32=s0_32, stat1_f32=s1_32, tv_init=tv_init,
              eob_e_h=e_h, eob_e_hh=e_hh, eob_whitened=a1)

    for nit, key in ((1, "F_it1"), (3, "F_it3")):
        fa = sidekit.FactorAnalyser()
        fa.total_variability_single(stat_file, ubm, R, nb_iter=nit, min_div=True, tv_init=tv_init.copy(),
                                    batch_size=7, save_init=False, output_file_name=None)
        tv[key] = fa.F.copy()
    fa_nomd = sidekit.FactorAnalyser()
    fa_nomd.total_variability_single(stat_file, ubm, R, nb_iter=2, min_div=False, tv_init=tv_init.copy(),
                                     batch_size=7, save_init=False, output_file_name=None)
    tv["F_it2_nomindiv"] = fa_nomd.F.copy()

    # pooled driver (the north-star entry point) -- must equal _single
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        fa2 = sidekit.FactorAnalyser()
        fa2.total_variability(stat_file, ubm, R, nb_iter=3, min_div=True, tv_init=tv_init.copy(),
                              batch_size=7, save_init=False, output_file_name=os.path.join(tmp, "tv"),
                              num_thread=1)
        tv["F_it3_pool"] = fa2.F.copy()
        tv["fa_mean"] = fa2.mean
        tv["fa_sigma"] = fa2.Sigma
        ivs, iv_sigma = fa2.extract_ivectors(ubm, stat_file, prefix="", batch_size=7, uncertainty=True,
                                             num_thread=1)
        tv["iv"] = ivs.stat1.copy()
        tv["iv_stat0"] = ivs.stat0.copy()
        tv["iv_sigma"] = iv_sigma.copy()
        # single-process extraction (float64 whitening path)
        ss2 = sidekit.StatServer(stat_file)
        ivs2, sig2 = fa2.extract_ivectors_single(ubm, ss2, uncertainty=True)
        tv["iv_single"] = ivs2.stat1.copy()
        tv["iv_sigma_single"] = sig2.copy()
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "tv.npz"), **tv)
    print("tv |F_it3 - F_it3_pool| =", np.abs(tv["F_it3"] - tv["F_it3_pool"]).max(),
          " |iv - iv_single| =", np.abs(tv["iv"] - tv["iv_single"]).max())


def gen_gmm_scoring(sidekit):
    """MAP adaptation (statserver.py:986-1028) + GMM-UBM log-likelihood-ratio scoring (gmm_scoring.py:56-119)."""
    C, D = 16, 10
    g = orc.random_gmm(C, D, seed=41)
    ubm = make_ubm(sidekit, g.w, g.mu, g.invcov)
    rs = np.random.RandomState(42)
    enr_lens = [int(x) for x in rs.randint(60, 200, size=7)]
    enr_spk = ["spkA", "spkB", "spkC", "spkA", "spkB", "spkC", "spkA"]
    tst_lens = [int(x) for x in rs.randint(30, 150, size=5)]
    table = {}
    for i, L in enumerate(enr_lens):
        table["e%03d" % i] = orc.synth_frames(L, dim=D, clusters=4, seed=200 + 7 * (ord(enr_spk[i][-1]) - 65), dtype=np.float64) \
            + 0.3 * np.random.RandomState(300 + i).randn(L, D)
    for i, L in enumerate(tst_lens):
        table["t%03d" % i] = orc.synth_frames(L, dim=D, clusters=4, seed=200 + 7 * (i % 3), dtype=np.float64) \
            + 0.3 * np.random.RandomState(400 + i).randn(L, D)
    fs = mem_server(sidekit, table)
    idmap = sidekit.IdMap()
    idmap.leftids = np.array(enr_spk)
    idmap.rightids = np.array(["e%03d" % i for i in range(len(enr_lens))])
    idmap.start = np.empty(len(enr_lens), "|O")
    idmap.stop = np.empty(len(enr_lens), "|O")
    ss = sidekit.StatServer(idmap, ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(len(enr_lens)), num_thread=1)
    enroll = ss.adapt_mean_map_multisession(ubm, r=16)
    ndx = sidekit.Ndx()
    ndx.modelset = np.array(["spkA", "spkB", "spkC"])
    ndx.segset = np.array(["t%03d" % i for i in range(len(tst_lens))])
    ndx.trialmask = np.array([[1, 1, 0, 1, 1], [1, 0, 1, 1, 0], [0, 1, 1, 1, 1]], dtype=bool)
    scores = np.zeros(ndx.trialmask.shape)
    sys.modules["sidekit.gmm_scoring"].gmm_scoring_singleThread(ubm, enroll, ndx, fs, scores)
    out = dict(w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, enr_lens=np.array(enr_lens), enr_spk=np.array(enr_spk),
               enr_frames=np.vstack([table["e%03d" % i] for i in range(len(enr_lens))]),
               tst_lens=np.array(tst_lens), tst_frames=np.vstack([table["t%03d" % i] for i in range(len(tst_lens))]),
               enroll_modelset=np.array(enroll.modelset), enroll_stat1=enroll.stat1.copy(),
               ndx_modelset=ndx.modelset, ndx_segset=ndx.segset, trialmask=ndx.trialmask, scores=scores)
    np.savez_compressed(os.path.join(OUT, "gmm_scoring.npz"), **out)
    print("gmm_scoring", scores)


def gen_cosine(sidekit):
    """cosine_scoring (iv_scoring.py:61-106) with and without a WCCN rotation; the Ndx names a model that was
    never enrolled and a segment that has no i-vector, and lists them in another order than the StatServers."""
    rs = np.random.RandomState(77)
    R = 12

    def iv_server(names_model, names_seg):
        ss = sidekit.StatServer()
        ss.modelset = np.array(names_model)
        ss.segset = np.array(names_seg)
        ss.start = np.empty(len(names_seg), "|O")
        ss.stop = np.empty(len(names_seg), "|O")
        ss.stat0 = np.ones((len(names_seg), 1))
        ss.stat1 = rs.randn(len(names_seg), R) * rs.uniform(0.5, 3.0, size=(len(names_seg), 1))
        return ss
    enroll = iv_server(["m%d" % i for i in range(6)], ["m%d" % i for i in range(6)])
    test = iv_server(["t%d" % i for i in range(8)], ["t%d" % i for i in range(8)])
    test.stat1[3] = 0.0                                   # a null i-vector: the 1e-8 clip of norm_stat1
    ndx = sidekit.Ndx()
    ndx.modelset = np.array(["m4", "m1", "mX", "m0", "m5"])
    ndx.segset = np.array(["t7", "t3", "tY", "t0", "t1", "t5"])
    ndx.trialmask = rs.rand(5, 6) < 0.6
    wccn = np.linalg.cholesky(np.cov(rs.randn(R, 200)) + 0.1 * np.eye(R))
    sc = sidekit.iv_scoring.cosine_scoring(enroll, test, ndx, wccn=None)
    scw = sidekit.iv_scoring.cosine_scoring(enroll, test, ndx, wccn=wccn)
    np.savez_compressed(os.path.join(OUT, "cosine_scoring.npz"),
                        enroll_modelset=enroll.modelset, enroll_iv=enroll.stat1, test_segset=test.segset,
                        test_iv=test.stat1, ndx_modelset=ndx.modelset, ndx_segset=ndx.segset, trialmask=ndx.trialmask,
                        wccn=wccn, scoremat=sc.scoremat, scoremask=sc.scoremask, out_modelset=sc.modelset,
                        out_segset=sc.segset, scoremat_wccn=scw.scoremat)
    print("cosine", sc.scoremat.shape, list(sc.modelset), list(sc.segset))


def gen_backend(sidekit):
    """i-vector back-end: StatServer covariance / LDA / WCCN / Mahalanobis matrices and whitening
    (statserver.py:732-925, 1228-1245), mahalanobis_scoring and two_covariance_scoring (iv_scoring.py:109-205).
    Speakers with 1..9 sessions, enrolment with repeated models (averaged by the scorers), an Ndx with a missing model
    and a missing segment."""
    rs = np.random.RandomState(99)
    R, K = 10, 14
    nsess = rs.randint(2, 10, size=K)
    nsess[3] = 1                                          # a one-session speaker: zero within-class scatter
    spk_mean = rs.randn(K, R) * 1.5
    models, rows = [], []
    for k in range(K):
        for j in range(nsess[k]):
            models.append("spk%02d" % k)
            rows.append(spk_mean[k] + rs.randn(R) * rs.uniform(0.3, 1.0, size=R) + 0.7)
    order = rs.permutation(len(models))                   # sessions of a speaker are not contiguous
    models = [models[i] for i in order]
    X = np.array([rows[i] for i in order])

    def server(modelset, segset, stat1):
        ss = sidekit.StatServer()
        ss.modelset = np.array(modelset)
        ss.segset = np.array(segset)
        ss.start = np.empty(len(segset), "|O")
        ss.stop = np.empty(len(segset), "|O")
        ss.stat0 = np.ones((len(segset), 1))
        ss.stat1 = np.array(stat1, dtype=np.float64)
        return ss

    segs = ["seg%03d" % i for i in range(len(models))]
    dev = server(models, segs, X)
    out = dict(R=R, modelset=dev.modelset, segset=dev.segset, stat1=dev.stat1.copy())
    out["total_cov"] = dev.get_total_covariance_stat1()
    out["within_cov"] = dev.get_within_covariance_stat1()
    out["between_cov"] = dev.get_between_covariance_stat1()
    out["lda_rank"] = 6
    out["lda"] = dev.get_lda_matrix_stat1(6)
    out["mahalanobis_matrix"] = dev.get_mahalanobis_matrix_stat1()
    out["wccn_chol"] = dev.get_wccn_choleski_stat1()
    mps = dev.mean_stat_per_model()
    out["mean_per_model_modelset"] = mps.modelset
    out["mean_per_model_stat0"] = mps.stat0
    out["mean_per_model_stat1"] = mps.stat1
    mu = dev.get_mean_stat1()
    out["mu"] = mu
    w = copy.deepcopy(dev)
    w.whiten_stat1(mu, out["total_cov"])
    out["whitened_full"] = w.stat1
    w = copy.deepcopy(dev)
    w.whiten_cholesky_stat1(mu, out["total_cov"])
    out["whitened_chol"] = w.stat1
    w = copy.deepcopy(dev)
    w.whiten_stat1(mu, out["wccn_chol"], isSqrInvSigma=True)
    out["whitened_presqrt"] = w.stat1

    # scoring: enrolment sessions with repeated models, test segments, an Ndx with unknown entries
    e_models = ["spk%02d" % k for k in (0, 1, 1, 2, 5, 5, 5, 7)]
    enroll = server(e_models, ["e%d" % i for i in range(len(e_models))], spk_mean[[0, 1, 1, 2, 5, 5, 5, 7]]
                    + 0.4 * rs.randn(len(e_models), R))
    test = server(["t%d" % i for i in range(9)], ["t%d" % i for i in range(9)],
                  spk_mean[rs.randint(0, K, size=9)] + 0.5 * rs.randn(9, R))
    ndx = sidekit.Ndx()
    ndx.modelset = np.array(["spk05", "spk01", "spkXX", "spk00", "spk07"])
    ndx.segset = np.array(["t7", "t3", "tY", "t0", "t1", "t5", "t8"])
    ndx.trialmask = rs.rand(5, 7) < 0.6
    out.update(enroll_modelset=enroll.modelset, enroll_segset=enroll.segset, enroll_iv=enroll.stat1.copy(),
               test_segset=test.segset, test_iv=test.stat1.copy(), ndx_modelset=ndx.modelset, ndx_segset=ndx.segset,
               trialmask=ndx.trialmask)
    sc = sidekit.iv_scoring.mahalanobis_scoring(copy.deepcopy(enroll), copy.deepcopy(test), ndx,
                                                out["mahalanobis_matrix"])
    out.update(maha_scoremat=sc.scoremat, maha_modelset=sc.modelset, maha_segset=sc.segset, maha_scoremask=sc.scoremask)
    M_ns = out["mahalanobis_matrix"] + 0.05 * rs.randn(R, R)          # a non-symmetric matrix is legal input
    out["maha_nonsym_matrix"] = M_ns
    out["maha_nonsym_scoremat"] = sidekit.iv_scoring.mahalanobis_scoring(copy.deepcopy(enroll), copy.deepcopy(test),
                                                                         ndx, M_ns).scoremat
    sc = sidekit.iv_scoring.two_covariance_scoring(copy.deepcopy(enroll), copy.deepcopy(test), ndx,
                                                   out["within_cov"], out["between_cov"])
    out.update(twocov_scoremat=sc.scoremat, twocov_modelset=sc.modelset, twocov_segset=sc.segset)
    # fast_PLDA_scoring (iv_scoring.py:364-471): closed set, open set (p_known), a scaling factor
    rank_f = 5
    Fp = 0.8 * rs.randn(R, rank_f)
    A = rs.randn(R, 3 * R)
    Sig = A.dot(A.T) / (3 * R) + 0.2 * np.eye(R)
    mu_p = dev.stat1.mean(axis=0)
    out.update(plda_F=Fp, plda_Sigma=Sig, plda_mu=mu_p)
    for tag, kw in (("closed", dict()), ("open", dict(p_known=0.3)), ("scaled", dict(p_known=0.1, scaling_factor=0.8))):
        sc = sidekit.iv_scoring.fast_PLDA_scoring(copy.deepcopy(enroll), copy.deepcopy(test), ndx, mu_p, Fp, Sig, **kw)
        out["plda_%s_scoremat" % tag] = sc.scoremat
    out.update(plda_modelset=sc.modelset, plda_segset=sc.segset)
    np.savez_compressed(os.path.join(OUT, "iv_backend.npz"), **out)
    print("backend", X.shape, out["lda"].shape, sc.scoremat.shape, list(sc.modelset))


def gen_post_processing(sidekit):
    """FeaturesServer.post_processing (features_server.py:202-246) on a few synthetic sessions: the experiment's
    configuration (model_interface.py:45-52: RASTA + delta + double delta + cmvn, all frames kept) and the other
    option combinations the device path covers.  Cases are stored as case<i>_{opt,x,label,feat,lab}."""
    rs = np.random.RandomState(91)
    out, n = {}, 0
    cases = [
        dict(rasta=True, delta=True, double_delta=True, feat_norm="cmvn", keep_all_features=True, T=311, D=20),
        dict(rasta=True, delta=True, double_delta=True, feat_norm="cmvn", keep_all_features=False, T=97, D=7),
        dict(rasta=False, delta=True, double_delta=True, feat_norm="cms", keep_all_features=False, T=150, D=13),
        dict(rasta=False, delta=True, double_delta=False, feat_norm="cmvn", keep_all_features=True, T=64, D=5),
        dict(rasta=True, delta=False, double_delta=False, feat_norm=None, keep_all_features=False, T=40, D=3),
        dict(rasta=False, delta=False, double_delta=False, feat_norm="cmvn", keep_all_features=True, T=33, D=4),
        dict(rasta=True, delta=True, double_delta=True, feat_norm="cmvn", keep_all_features=True, T=5, D=2),
        dict(rasta=False, delta=True, double_delta=True, feat_norm="cmvn", keep_all_features=True, T=1, D=2),
        dict(rasta=False, delta=True, double_delta=True, feat_norm="cmvn", keep_all_features=False, T=50, D=6,
             mask="[1-3,5]"),
        dict(rasta=True, delta=True, double_delta=True, feat_norm="cmvn", keep_all_features=True, T=80, D=4,
             use_global=True),
        dict(rasta=False, delta=True, double_delta=True, feat_norm="cmvn", keep_all_features=True, T=30, D=3,
             no_speech=True),
    ]
    for c in cases:
        T, D = c["T"], c["D"]
        x = 3.0 * rs.randn(T, D) + 5.0 * rs.randn(1, D) + np.cumsum(0.3 * rs.randn(T, D), axis=0)
        label = rs.rand(T) > 0.35
        if c.get("no_speech"):
            label[:] = False
        elif not label.any():
            label[0] = True
        fs = sidekit.FeaturesServer(dataset_list=["cep"], mask=c.get("mask"), feat_norm=c["feat_norm"],
                                    delta=c["delta"], double_delta=c["double_delta"], rasta=c["rasta"],
                                    keep_all_features=c["keep_all_features"])
        gm = gs = None
        if c.get("use_global"):
            dout = D * (1 + int(c["delta"]) + int(c["delta"] and c["double_delta"]))
            gm, gs = rs.randn(dout), rs.uniform(0.5, 2.0, size=dout)
        feat, lab = fs.post_processing(x.copy(), label.copy(), gm, gs)
        opt = np.array([int(c["rasta"]), int(c["delta"]), int(c["double_delta"]),
                        {None: 0, "cms": 1, "cmvn": 2}[c["feat_norm"]], int(c["keep_all_features"])])
        out["case%d_opt" % n] = opt
        out["case%d_x" % n] = x
        out["case%d_label" % n] = label
        out["case%d_feat" % n] = np.asarray(feat)
        out["case%d_lab" % n] = np.asarray(lab)
        if c.get("mask"):
            out["case%d_mask" % n] = np.array(fs.mask)
        if gm is not None:
            out["case%d_gmean" % n] = gm
            out["case%d_gstd" % n] = gs
        print("post_processing case", n, c, "->", np.asarray(feat).shape, np.asarray(feat).dtype)
        n += 1
    out["n_cases"] = np.array(n)
    np.savez_compressed(os.path.join(OUT, "post_processing.npz"), **out)


def gen_sliding(sidekit):
    """FeaturesServer.post_processing with feat_norm 'cmvn_sliding' / 'cms_sliding' (features_server.py:282-287 ->
    frontend/normfeat.py:194-229: pandas rolling mean / sample std over 301 labelled frames, edges replicated;
    sessions with at most 301 labelled frames fall back to whole-session cmvn / cms)."""
    rs = np.random.RandomState(4242)
    out, n = {}, 0
    cases = [
        dict(rasta=False, delta=True, double_delta=False, feat_norm="cmvn_sliding", keep_all_features=True, T=900, D=5),
        dict(rasta=True, delta=False, double_delta=False, feat_norm="cms_sliding", keep_all_features=False, T=700, D=4),
        dict(rasta=False, delta=False, double_delta=False, feat_norm="cmvn_sliding", keep_all_features=False, T=1203, D=3),
        dict(rasta=False, delta=True, double_delta=True, feat_norm="cmvn_sliding", keep_all_features=True, T=350, D=3),
        dict(rasta=False, delta=False, double_delta=False, feat_norm="cms_sliding", keep_all_features=True, T=120, D=2),
        dict(rasta=False, delta=False, double_delta=False, feat_norm="cmvn_sliding", keep_all_features=True, T=302, D=2,
             all_speech=True),
        dict(rasta=False, delta=False, double_delta=False, feat_norm="cmvn_sliding", keep_all_features=True, T=301, D=2,
             all_speech=True),
    ]
    for c in cases:
        T, D = c["T"], c["D"]
        x = 3.0 * rs.randn(T, D) + 5.0 * rs.randn(1, D) + np.cumsum(0.3 * rs.randn(T, D), axis=0)
        label = rs.rand(T) > 0.3
        if c.get("all_speech"):
            label[:] = True
        fs = sidekit.FeaturesServer(dataset_list=["cep"], feat_norm=c["feat_norm"], delta=c["delta"],
                                    double_delta=c["double_delta"], rasta=c["rasta"],
                                    keep_all_features=c["keep_all_features"])
        feat, lab = fs.post_processing(x.copy(), label.copy())
        out["case%d_opt" % n] = np.array([int(c["rasta"]), int(c["delta"]), int(c["double_delta"]),
                                          {"cms_sliding": 3, "cmvn_sliding": 4}[c["feat_norm"]],
                                          int(c["keep_all_features"])])
        out["case%d_x" % n] = x
        out["case%d_label" % n] = label
        out["case%d_feat" % n] = np.asarray(feat)
        out["case%d_lab" % n] = np.asarray(lab)
        print("sliding case", n, c, "->", np.asarray(feat).shape, int(label.sum()), "labelled")
        n += 1
    out["n_cases"] = np.array(n)
    np.savez_compressed(os.path.join(OUT, "post_processing_sliding.npz"), **out)


def vad_server(sidekit, table, labels, keep_all_features):
    """In-memory server that goes through the reference's OWN post_processing (features_server.py:202-246), so
    that keep_all_features / VAD selection behave exactly as for a file-backed server; stack_features_parallel
    loads every show with the server's current setting, as the reference's does (features_server.py:634-666)."""
    class Mem(sidekit.FeaturesServer):
        def __init__(self):
            super().__init__(dataset_list=["cep"], keep_all_features=keep_all_features)
            self.table = table

        def load(self, show, channel=0, input_feature_filename=None, label=None, start=None, stop=None):
            return self.post_processing(self.table[show].copy(), labels[show].copy())

        def stack_features_parallel(self, show_list, num_thread=1, **kw):
            return np.vstack([self.load(s)[0] for s in show_list])
    return Mem()


def gen_em_split_vad(sidekit):
    """EM_split on files WITH VAD labels and keep_all_features=True (model_interface.py:52): _init sees every
    frame, the E-steps only the labelled ones (mixture.py:585)."""
    rs = np.random.RandomState(23)
    D, K = 5, 4
    centers = rs.randn(K, D) * 6.0
    chunks, labels = {}, {}
    for i in range(3):
        n = 400 + 53 * i
        lab = rs.randint(0, K, size=n)
        x = centers[lab] + 0.5 * rs.randn(n, D)
        vad = rs.rand(n) > 0.4
        x[~vad] = 0.3 * rs.randn(int((~vad).sum()), D)        # "silence" frames look different
        chunks["f%d" % i], labels["f%d" % i] = x, vad
    fs = vad_server(sidekit, chunks, labels, True)
    ubm = sidekit.Mixture()
    llk = ubm.EM_split(features_server=fs, feature_list=list(chunks.keys()), distrib_nb=4, num_thread=1,
                       save_partial=False)
    out = {"chunk%d" % i: chunks["f%d" % i] for i in range(3)}
    out.update({"vad%d" % i: labels["f%d" % i] for i in range(3)})
    out.update(llk=np.array(llk), w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, ctl=ubm.cov_var_ctl,
               keep_all_after=np.array(int(fs.keep_all_features)))
    np.savez_compressed(os.path.join(OUT, "em_split_vad.npz"), **out)
    print("em_split_vad llk", llk[-2:], "iters", len(llk), "keep_all_features after:", fs.keep_all_features)


def gen_tv_big(sidekit, tmp):
    """A TV case whose rank spans several 32-wide blocks of the device Cholesky (R = 40 < C*D, full rank,
    well conditioned): C=64, D=16, S=120 sessions, 3 EM iterations + extraction.  tv_init is
    0.1 * RandomState(7).randn(C*D, R) and is regenerated by the test."""
    C, D, R, S = 64, 16, 40, 120
    g = orc.random_gmm(C, D, seed=41)
    ubm = make_ubm(sidekit, g.w, g.mu, g.invcov)
    rs = np.random.RandomState(42)
    lens = list(rs.randint(100, 400, size=S))
    table = {"s%03d" % i: orc.synth_frames(L, dim=D, clusters=16, seed=500 + i, dtype=np.float64)
             for i, L in enumerate(lens)}
    fs = mem_server(sidekit, table)
    idmap = sidekit.IdMap()
    idmap.leftids = np.array(["spk%d" % (i % 12) for i in range(S)])
    idmap.rightids = np.array(sorted(table.keys()))
    idmap.start = np.empty(S, "|O")
    idmap.stop = np.empty(S, "|O")
    ss = sidekit.StatServer(idmap, ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(S), num_thread=1)
    stat_file = os.path.join(tmp, "stat_big.h5")
    ss.write(stat_file)
    tv_init = 0.1 * np.random.RandomState(7).randn(C * D, R)
    fa = sidekit.FactorAnalyser()
    fa.total_variability_single(stat_file, ubm, R, nb_iter=3, min_div=True, tv_init=tv_init.copy(),
                                batch_size=30, save_init=False, output_file_name=None)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        ivs, iv_sigma = fa.extract_ivectors(ubm, stat_file, prefix="", batch_size=30, uncertainty=True, num_thread=1)
    finally:
        os.chdir(cwd)
    out = dict(w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, stat0_f32=ss.stat0.astype(np.float32),
               stat1_f32=ss.stat1.astype(np.float32), F_it3=fa.F.copy(), iv=ivs.stat1.copy(),
               iv_sigma=iv_sigma.copy(), shape=np.array([C, D, R, S]))
    np.savez_compressed(os.path.join(OUT, "tv_big.npz"), **out)
    print("tv_big F", fa.F.shape, "|F|max", np.abs(fa.F).max(), "iv", ivs.stat1.shape)


def gen_plda(sidekit, tmp):
    """PLDA training (FactorAnalyser.plda, factor_analyser.py:810-910) on synthetic i-vectors: speakers with 1..8
    sessions in shuffled order; plain run (3 iterations), a scaling factor, and one iteration alone (the partial
    product of the first).  The model files go through the stand-in h5py into a temporary directory."""
    rs = np.random.RandomState(2024)
    R, K, rank_f = 24, 40, 8
    nsess = rs.randint(1, 9, size=K)
    V = rs.randn(R, rank_f) * 1.2
    models, rows = [], []
    for k in range(K):
        y = rs.randn(rank_f)
        for j in range(nsess[k]):
            models.append("spk%02d" % k)
            rows.append(V.dot(y) + rs.randn(R) * rs.uniform(0.4, 1.1, size=R) + 0.3)
    order = rs.permutation(len(models))
    ss = sidekit.StatServer()
    ss.modelset = np.array([models[i] for i in order])
    ss.segset = np.array(["seg%03d" % i for i in range(len(models))])
    ss.start = np.empty(len(models), "|O")
    ss.stop = np.empty(len(models), "|O")
    ss.stat0 = np.ones((len(models), 1))
    ss.stat1 = np.array([rows[i] for i in order], dtype=np.float64)
    out = dict(R=R, rank_f=rank_f, modelset=ss.modelset, segset=ss.segset, stat1=ss.stat1.copy())
    for tag, kw in (("it3", dict(nb_iter=3)), ("it1", dict(nb_iter=1)), ("scaled", dict(nb_iter=2, scaling_factor=0.6))):
        fa = sidekit.FactorAnalyser()
        fa.plda(copy.deepcopy(ss), rank_f, output_file_name=os.path.join(tmp, "plda_" + tag), **kw)
        out["mean_" + tag], out["F_" + tag], out["Sigma_" + tag] = fa.mean, fa.F, fa.Sigma
    np.savez_compressed(os.path.join(OUT, "plda.npz"), **out)
    print("plda", ss.stat1.shape, out["F_it3"].shape, float(np.abs(out["Sigma_it3"]).max()))


def gen_full_plda(sidekit):
    """full_PLDA_scoring (iv_scoring.py:266-361) and its dispatch through PLDA_scoring(full_model=True): closed set, open
    set, a scaling factor; an Ndx naming an unknown model and an unknown segment."""
    rs = np.random.RandomState(77)
    R, rank_f, rank_g = 12, 5, 4

    def server(modelset, segset, stat1):
        ss = sidekit.StatServer()
        ss.modelset = np.array(modelset)
        ss.segset = np.array(segset)
        ss.start = np.empty(len(segset), "|O")
        ss.stop = np.empty(len(segset), "|O")
        ss.stat0 = np.ones((len(segset), 1))
        ss.stat1 = np.array(stat1, dtype=np.float64)
        return ss

    F = 0.9 * rs.randn(R, rank_f)
    G = 0.5 * rs.randn(R, rank_g)
    A = rs.randn(R, 3 * R)
    Sigma = A.dot(A.T) / (3 * R) + 0.3 * np.eye(R)
    mu = 0.4 * rs.randn(R)
    e_models = ["m%02d" % k for k in range(7)]
    enroll = server(e_models, ["e%d" % i for i in range(7)], F.dot(rs.randn(rank_f, 7)).T + 0.5 * rs.randn(7, R) + mu)
    test = server(["t%d" % i for i in range(9)], ["t%d" % i for i in range(9)],
                  F.dot(rs.randn(rank_f, 9)).T + 0.6 * rs.randn(9, R) + mu)
    ndx = sidekit.Ndx()
    ndx.modelset = np.array(["m05", "m01", "mXX", "m00", "m06", "m03"])
    ndx.segset = np.array(["t7", "t3", "tY", "t0", "t1", "t5", "t8"])
    ndx.trialmask = rs.rand(6, 7) < 0.7
    out = dict(F=F, G=G, Sigma=Sigma, mu=mu, enroll_modelset=enroll.modelset, enroll_segset=enroll.segset,
               enroll_iv=enroll.stat1.copy(), test_segset=test.segset, test_iv=test.stat1.copy(),
               ndx_modelset=ndx.modelset, ndx_segset=ndx.segset, trialmask=ndx.trialmask)
    for tag, kw in (("closed", dict()), ("open", dict(p_known=0.4)), ("scaled", dict(p_known=0.2, scaling_factor=0.7))):
        sc = sidekit.iv_scoring.full_PLDA_scoring(copy.deepcopy(enroll), copy.deepcopy(test), ndx, mu, F, G, Sigma, **kw)
        out["full_%s_scoremat" % tag] = sc.scoremat
    sc2 = sidekit.iv_scoring.PLDA_scoring(copy.deepcopy(enroll), copy.deepcopy(test), ndx, mu, F, G, Sigma, p_known=0.4,
                                          full_model=True)
    assert np.array_equal(sc2.scoremat, out["full_open_scoremat"])
    out.update(modelset=sc.modelset, segset=sc.segset, scoremask=sc.scoremask)
    np.savez_compressed(os.path.join(OUT, "full_plda.npz"), **out)
    print("full_plda", sc.scoremat.shape, list(sc.modelset))


def gen_raw(sidekit):
    """Un-normalised, raw-MFCC-like frames (per-dimension offsets up to +-30, variances 1..100, no cmvn): one
    Mixture._expectation (mixture.py:537-568) and one StatServer.accumulate_stat (statserver.py:609-667), plus a
    short EM_split, for the frame-centring path (round 3)."""
    import copy
    C, D, T = 64, 60, 2000
    g = orc.raw_gmm(C, D, seed=41)
    cep = orc.synth_raw_frames(T, dim=D, clusters=24, seed=42, dtype=np.float32).astype(np.float64)
    ubm = make_ubm(sidekit, g.w, g.mu, g.invcov)
    accum = copy.deepcopy(ubm)
    accum._reset()
    llk = ubm._expectation(accum, cep)
    lp = ubm.compute_log_posterior_probabilities(cep[:200])
    _, log_lk = sidekit.statserver.sum_log_probabilities(ubm.compute_log_posterior_probabilities(cep))
    out = dict(w=ubm.w, mu=ubm.mu, invcov=ubm.invcov, cep=cep.astype(np.float32), N=accum.w.copy(), F=accum.mu.copy(),
               S=accum.invcov.copy(), llk=np.float64(llk), lp200=lp, log_lk=log_lk)
    # Baum-Welch statistics of ragged sessions of the same kind of data
    rs = np.random.RandomState(43)
    lens = [1, 5, 33, 64, 150] + list(rs.randint(30, 250, size=15))
    table = {"r%03d" % i: orc.synth_raw_frames(L, dim=D, clusters=24, seed=200 + i, dtype=np.float32).astype(np.float64)
             for i, L in enumerate(lens)}
    fs = mem_server(sidekit, table)
    S = len(lens)
    idmap = sidekit.IdMap()
    idmap.leftids = np.array(["spk%d" % (i % 5) for i in range(S)])
    idmap.rightids = np.array(sorted(table.keys()))
    idmap.start = np.empty(S, "|O")
    idmap.stop = np.empty(S, "|O")
    ss = sidekit.StatServer(idmap, ubm=ubm)
    ss.accumulate_stat(ubm=ubm, feature_server=fs, seg_indices=range(S), num_thread=1)
    out.update(bw_lens=np.array(lens), bw_frames=np.vstack([table[k] for k in sorted(table.keys())]).astype(np.float32),
               stat0=ss.stat0.copy(), stat1=ss.stat1.copy())
    # EM_split to 8 components on well-separated raw-scale clusters (split arg-max stays robust)
    rs = np.random.RandomState(44)
    D2, K = 8, 8
    off = rs.uniform(-30, 30, size=D2)
    centers = off + rs.randn(K, D2) * 12.0
    sd = np.sqrt(np.exp(rs.uniform(0.0, np.log(9.0), size=(K, D2))))
    chunks = {}
    for i in range(3):
        lab = rs.randint(0, K, size=700 + 41 * i)
        chunks["f%d" % i] = (centers[lab] + sd[lab] * rs.randn(lab.shape[0], D2)).astype(np.float32).astype(np.float64)
    fs2 = mem_server(sidekit, chunks)
    u = sidekit.Mixture()
    llk_em = u.EM_split(features_server=fs2, feature_list=list(chunks.keys()), distrib_nb=8, num_thread=1,
                        save_partial=False)
    for i in range(3):
        out["em_chunk%d" % i] = chunks["f%d" % i].astype(np.float32)
    out.update(em_llk=np.array(llk_em), em_w=u.w, em_mu=u.mu, em_invcov=u.invcov)
    np.savez_compressed(os.path.join(OUT, "raw_features.npz"), **out)
    print("raw_features: llk", llk, "stat0", ss.stat0.shape, "EM_split llk", llk_em[-2:])


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "raw":
        gen_raw(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        sidekit = import_reference()
        gen_em_split_vad(sidekit)
        with tempfile.TemporaryDirectory() as tmp:
            gen_tv_big(sidekit, tmp)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "full_plda":
        gen_full_plda(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "plda":
        with tempfile.TemporaryDirectory() as tmp:
            gen_plda(import_reference(), tmp)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "sliding":
        gen_sliding(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "backend":
        gen_backend(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cosine":
        gen_cosine(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "postproc":
        gen_post_processing(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "scoring":
        gen_gmm_scoring(import_reference())
        return
    sidekit = import_reference()
    print("reference sidekit", getattr(sidekit, "__version__", "?"), "STAT_TYPE", sidekit.STAT_TYPE)
    gen_estep(sidekit, "c8_d6", C=8, D=6, T=300, seed=3)
    gen_estep(sidekit, "c64_d60", C=64, D=60, T=2000, seed=5)
    gen_estep(sidekit, "c40_d13", C=40, D=13, T=777, seed=9)
    gen_em_split(sidekit)
    gen_em_uniform(sidekit)
    with tempfile.TemporaryDirectory() as tmp:
        gen_stats_and_tv(sidekit, tmp)
    gen_gmm_scoring(sidekit)
    gen_cosine(sidekit)
    gen_backend(sidekit)
    gen_sliding(sidekit)
    gen_post_processing(sidekit)
    gen_em_split_vad(sidekit)
    with tempfile.TemporaryDirectory() as tmp:
        gen_tv_big(sidekit, tmp)
        gen_plda(sidekit, tmp)
    gen_full_plda(sidekit)
    gen_raw(sidekit)


if __name__ == "__main__":
    main()
