"""CPU oracle for the GMM-UBM / i-vector statistics hot path (TEST INFRASTRUCTURE ONLY).

This is a plain NumPy/SciPy float64 restatement of what the reference (SIDEKIT 1.3.2 as
vendored in Anwarvic/Speaker-Recognition) computes on the hot path.  It is NOT the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the shipped package (``speaker-recognition_amd/``) never does and fails loudly when the
HIP library is missing.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by importing the real reference in the build container
(``tools/gen_golden.py``; fixtures in ``tests/golden/*.npz``).  The reference has no tests or
golden vectors of its own, so those generated fixtures are the pin.

Each function cites the reference file:line it follows (paths relative to the reference
root).  Quirks of the reference are reproduced on purpose (second moment used as covariance
in the 1-Gaussian init, `<=`/`>=` in the variance control, float32 in-place whitening in the
TV E-step, upper Cholesky factor in minimum divergence ...).
"""
import math

import numpy as np
import scipy.linalg

F64 = np.float64


# ----------------------------------------------------------------------------------------
# GMM container
# ----------------------------------------------------------------------------------------
class GMM(object):
    """Diagonal-covariance GMM parameters (sidekit/mixture.py:181-197)."""

    __slots__ = ("w", "mu", "invcov", "cov_var_ctl", "cst", "det", "A")

    def __init__(self, w, mu, invcov, cov_var_ctl=None):
        self.w = np.array(w, dtype=F64)
        self.mu = np.array(mu, dtype=F64)
        self.invcov = np.array(invcov, dtype=F64)
        self.cov_var_ctl = (1.0 / self.invcov) if cov_var_ctl is None else np.array(cov_var_ctl, dtype=F64)
        self.cst = None
        self.det = None
        self.A = None
        self.refresh()

    def copy(self):
        g = GMM.__new__(GMM)
        for k in self.__slots__:
            v = getattr(self, k)
            setattr(g, k, None if v is None else np.array(v, copy=True))
        return g

    @property
    def C(self):
        return self.w.shape[0]

    @property
    def D(self):
        return self.mu.shape[1]

    def refresh(self):
        """det, cst and the data independent term A (sidekit/mixture.py:382-396)."""
        dim = self.mu.shape[1]
        self.det = 1.0 / np.prod(self.invcov, axis=1)
        self.cst = 1.0 / (np.sqrt(self.det) * (2.0 * np.pi) ** (dim / 2.0))
        self.A = (np.square(self.mu) * self.invcov).sum(1) - 2.0 * (np.log(self.w) + np.log(self.cst))


# ----------------------------------------------------------------------------------------
# E-step pieces
# ----------------------------------------------------------------------------------------
def log_posterior(gmm, cep, mu=None):
    """Weighted per-component log-likelihoods, T x C (sidekit/mixture.py:461-487)."""
    cep = np.asarray(cep, dtype=F64)
    if cep.ndim == 1:
        cep = cep[None, :]
    A = gmm.A
    m = gmm.mu
    if mu is not None:
        m = np.asarray(mu, dtype=F64).reshape(gmm.mu.shape)
        A = (np.square(m) * gmm.invcov).sum(1) - 2.0 * (np.log(gmm.w) + np.log(gmm.cst))
    B = np.dot(np.square(cep), gmm.invcov.T) - 2.0 * np.dot(cep, (m * gmm.invcov).T)
    return -0.5 * (B + A)


def logsumexp_rows(lp):
    """Row-wise log-sum-exp and posteriors.

    Returns (pp, log_lk) with log_lk per row (sidekit/statserver.py:83-95); the Mixture
    flavour (sidekit/mixture.py:52-64) returns log_lk.sum() instead.
    Rows whose max is not finite keep log_lk = max (mixture.py:59-61).
    """
    top = np.max(lp, axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        log_lk = top + np.log(np.sum(np.exp(lp - top[:, None]), axis=1))
        bad = ~np.isfinite(top)
        if bad.any():
            log_lk[bad] = top[bad]
        pp = np.exp(lp - log_lk[:, None])
    return pp, log_lk


def expectation(gmm, cep):
    """One E-step over a block of frames (sidekit/mixture.py:537-568).

    Returns zero/first/second order statistics N[C], F[C,D], S[C,D] and the summed
    log-likelihood.
    """
    cep = np.asarray(cep, dtype=F64)
    lp = log_posterior(gmm, cep)
    pp, log_lk = logsumexp_rows(lp)
    N = pp.sum(0)
    F = np.dot(cep.T, pp).T
    S = np.dot(np.square(cep.T), pp).T
    return N, F, S, float(log_lk.sum())


def expectation_chunked(gmm, chunks):
    """E-step over a list of frame blocks, accumulating like mixture.py:584-588."""
    C, D = gmm.C, gmm.D
    N = np.zeros(C)
    F = np.zeros((C, D))
    S = np.zeros((C, D))
    llk = 0.0
    for cep in chunks:
        n, f, s, l = expectation(gmm, cep)
        N += n
        F += f
        S += s
        llk += l
    return N, F, S, llk


# ----------------------------------------------------------------------------------------
# M-step, split, init
# ----------------------------------------------------------------------------------------
def variance_control(cov, flooring, ceiling, cov_ctl):
    """Floor / ceil a covariance with inclusive comparisons (sidekit/mixture.py:489-506)."""
    floor = flooring * cov_ctl
    ceil = ceiling * cov_ctl
    lo = cov <= floor
    hi = cov >= ceil
    cov = cov.copy()
    cov[lo] = floor[lo]
    cov[hi] = ceil[hi]
    return cov


def maximization(gmm, N, F, S, ceil_cov=10, floor_cov=1e-2):
    """M-step in place on `gmm` (sidekit/mixture.py:590-611)."""
    gmm.w = N / np.sum(N)
    gmm.mu = F / N[:, None]
    cov = S / N[:, None] - np.square(gmm.mu)
    cov = variance_control(cov, floor_cov, ceil_cov, gmm.cov_var_ctl)
    gmm.invcov = 1.0 / cov
    gmm.refresh()
    return gmm


def split(gmm):
    """C -> 2C along the largest-variance axis (sidekit/mixture.py:517-535)."""
    sigma = 1.0 / gmm.invcov
    top = np.max(sigma, axis=1)
    axis = np.argmax(sigma, axis=1)
    shift = np.zeros(gmm.mu.shape)
    shift[np.arange(axis.shape[0]), axis] = np.sqrt(top)
    gmm.mu = np.vstack((gmm.mu - shift, gmm.mu + shift))
    gmm.invcov = np.vstack((gmm.invcov, gmm.invcov))
    gmm.w = np.concatenate([gmm.w, gmm.w]) * 0.5
    gmm.cov_var_ctl = np.vstack((gmm.cov_var_ctl, gmm.cov_var_ctl))
    gmm.refresh()
    return gmm


def init_single(features):
    """1-Gaussian start; covariance is the raw second moment (sidekit/mixture.py:629-641)."""
    features = np.asarray(features, dtype=F64)
    mu = features.mean(0)
    cov = (features ** 2).mean(0)
    return GMM(np.asarray([1.0]), mu[None], 1.0 / cov[None])


def init_uniform(cep, distrib_nb):
    """EM_uniform start (sidekit/mixture.py:796-824) including its quirks.

    `end = max(start + 10, nb)` is always nb, rows i > 0 stack `cov` (not 1/cov) into
    invcov, and mu stays 1-D when distrib_nb == 1 (in which case the reference cannot
    run an E-step; we keep 2-D there because no caller reaches it).
    """
    cep = np.asarray(cep, dtype=F64)
    nb = cep.shape[0]
    w = np.full(distrib_nb, 1.0 / distrib_nb, "d")
    mus, ics = [], []
    for i in range(distrib_nb):
        start = nb // distrib_nb * i
        end = max(start + 10, nb)
        mus.append(np.mean(cep[start:end, :], axis=0))
        cov = (cep[start:end, :] ** 2).mean(0)
        ics.append(1.0 / cov if i == 0 else cov)
    return GMM(w, np.vstack(mus), np.vstack(ics))


def em_split(chunks, distrib_nb, iterations=(1, 2, 2, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8),
             ceil_cov=10, floor_cov=1e-2, trace=None, init_chunks=None):
    """Binary-split EM (sidekit/mixture.py:643-733), num_thread=1 semantics.

    `chunks` is the list of per-file frame blocks the features server returns to the E-steps, i.e.
    with keep_all_features = False (sidekit/mixture.py:585); `init_chunks` the blocks _init stacks
    with the server's own setting (sidekit/mixture.py:629) when that differs (VAD labels present and
    keep_all_features = True).  Returns (gmm, llk list).  `trace`, if a list, receives a copy of the
    model after every M-step.
    """
    gmm = init_single(np.vstack(chunks if init_chunks is None else init_chunks))
    llk = []
    for it in iterations[:int(np.log2(distrib_nb))]:
        split(gmm)
        for _ in range(it):
            N, F, S, l = expectation_chunked(gmm, chunks)
            llk.append(l / np.sum(N))
            maximization(gmm, N, F, S, ceil_cov=ceil_cov, floor_cov=floor_cov)
            if trace is not None:
                trace.append(gmm.copy())
    return gmm, llk


def em_uniform(cep, distrib_nb, iteration_min=3, iteration_max=10, llk_gain=0.01, gmm=None):
    """Plain EM from a uniform-slice start (sidekit/mixture.py:735-794)."""
    cep = np.asarray(cep, dtype=F64)
    if gmm is None:
        gmm = init_uniform(cep, distrib_nb)
    llk = []
    for i in range(iteration_max):
        N, F, S, l = expectation(gmm, cep)
        llk.append(l / cep.shape[0])
        maximization(gmm, N, F, S)
        if i > 0:
            gain = llk[-1] - llk[-2]
            if gain < llk_gain and i >= iteration_min:
                break
    return gmm, llk


# ----------------------------------------------------------------------------------------
# Baum-Welch statistics
# ----------------------------------------------------------------------------------------
def accumulate_stat(gmm, sessions):
    """Per-session zero / first order statistics (sidekit/statserver.py:633-667).

    `sessions` is a list of (T_s x D) arrays already cut to start:stop and VAD-selected.
    stat1 is component-major: index c*D + d.
    """
    C, D = gmm.C, gmm.D
    stat0 = np.zeros((len(sessions), C))
    stat1 = np.zeros((len(sessions), C * D))
    for i, data in enumerate(sessions):
        data = np.asarray(data, dtype=F64)
        if data.shape[1] != D:
            raise Exception('dimension of ubm and features differ: {:d} / {:d}'.format(D, data.shape[1]))
        pp, _ = logsumexp_rows(log_posterior(gmm, data))
        stat0[i] = pp.sum(0)
        stat1[i] = np.dot(data.T, pp).T.reshape(C * D)
    return stat0, stat1


def whiten_stat1(stat0, stat1, mu_sv, sigma_sv):
    """float64 centring + diagonal whitening (sidekit/statserver.py:690-697, 743-745)."""
    C = stat0.shape[1]
    D = stat1.shape[1] // C
    index_map = np.repeat(np.arange(C), D)
    out = stat1 - stat0[:, index_map] * mu_sv.astype(F64)
    return out / np.sqrt(sigma_sv.astype(F64))


# ----------------------------------------------------------------------------------------
# Total variability
# ----------------------------------------------------------------------------------------
def whiten_inplace_like_reference(stat0, stat1, gmm):
    """The in-place whitening at the top of e_on_batch (sidekit/factor_analyser.py:76-79).

    stat1 keeps ITS OWN dtype: with float32 inputs (statistics read back from the
    float32 HDF5 datasets, statserver.py:439-446) each of the two in-place updates is
    computed in float64 and rounded to float32.
    """
    C = stat0.shape[1]
    D = gmm.D
    index_map = np.repeat(np.arange(C), D)
    stat1 -= stat0[:, index_map] * gmm.mu.flatten()
    stat1 *= np.sqrt(gmm.invcov.flatten())
    return stat1


def triu_pack_index(R):
    return np.triu_indices(R)


def e_on_batch(stat0, stat1, gmm, Fm):
    """Posterior moments of the latent factor for a batch (sidekit/factor_analyser.py:52-92).

    stat1 is whitened IN PLACE (dtype preserved).  Returns e_h (B x R) and e_hh
    (B x R(R+1)/2, row-major upper triangle), both float64.
    """
    R = Fm.shape[1]
    C = stat0.shape[1]
    D = gmm.D
    index_map = np.repeat(np.arange(C), D)
    iu = np.triu_indices(R)
    B = stat0.shape[0]
    e_h = np.zeros((B, R))
    e_hh = np.zeros((B, R * (R + 1) // 2))
    whiten_inplace_like_reference(stat0, stat1, gmm)
    for i in range(B):
        inv_lambda = scipy.linalg.inv(np.eye(R) + (Fm.T * stat0[i, index_map]).dot(Fm))
        aux = Fm.T.dot(stat1[i, :])
        e_h[i] = np.dot(aux, inv_lambda)
        e_hh[i] = (inv_lambda + np.outer(e_h[i], e_h[i]))[iu]
    return e_h, e_hh


def batch_split(nb_sessions, batch_size):
    """Batch index lists (sidekit/factor_analyser.py:618-619)."""
    batch_nb = int(np.floor(nb_sessions / float(batch_size) + 0.999))
    return np.array_split(np.arange(nb_sessions), batch_nb)


def tv_accumulate(stat0, stat1, gmm, Fm, batch_size=300):
    """One TV E-step: accumulators _A (C x P), _C (R x CD), _R (P) (factor_analyser.py:604-651).

    stat0 / stat1 are the stored statistics (float32 when they come from a file); they are
    NOT modified (each batch is copied like an HDF5 read would).
    """
    R = Fm.shape[1]
    C = stat0.shape[1]
    P = R * (R + 1) // 2
    _A = np.zeros((C, P))
    _C = np.zeros((R, stat1.shape[1]))
    _R = np.zeros(P)
    for idx in batch_split(stat0.shape[0], batch_size):
        s0 = np.array(stat0[idx, :], copy=True)
        s1 = np.array(stat1[idx, :], copy=True)
        e_h, e_hh = e_on_batch(s0, s1, gmm, Fm)
        _A += s0.T.dot(e_hh)
        _C += e_h.T.dot(s1)
        _R += np.sum(e_hh, axis=0)
    _R /= stat0.shape[0]
    return _A, _C, _R


def tv_mstep(_A, _C, _R, D, min_div=True):
    """M-step + minimum divergence (sidekit/factor_analyser.py:653-665)."""
    C, P = _A.shape
    R = _C.shape[0]
    iu = np.triu_indices(R)
    Fm = np.zeros((C * D, R))
    tmp = np.zeros((R, R))
    for c in range(C):
        tmp[iu] = tmp.T[iu] = _A[c, :]
        Fm[c * D:(c + 1) * D, :] = scipy.linalg.solve(tmp, _C[:, c * D:(c + 1) * D]).T
    if min_div:
        rt = np.zeros((R, R))
        rt[iu] = rt.T[iu] = _R
        ch = scipy.linalg.cholesky(rt)          # upper factor (scipy default)
        Fm = Fm.dot(ch)
    return Fm


def total_variability(stat0, stat1, gmm, tv_rank, nb_iter=20, min_div=True, tv_init=None,
                      batch_size=300, trace=None):
    """TV matrix EM (sidekit/factor_analyser.py:535-672).  tv_init is required for parity
    (the reference draws from the unseeded global RNG otherwise, :587)."""
    assert tv_init is not None and tv_init.shape[1] == tv_rank
    Fm = np.array(tv_init, dtype=F64, copy=True)
    for _ in range(nb_iter):
        _A, _C, _R = tv_accumulate(stat0, stat1, gmm, Fm, batch_size)
        Fm = tv_mstep(_A, _C, _R, gmm.D, min_div)
        if trace is not None:
            trace.append(Fm.copy())
    return Fm


def extract_ivectors(stat0, stat1, gmm, Fm, batch_size=300):
    """i-vectors and diagonal uncertainty (sidekit/factor_analyser.py:131-164, 732-808)."""
    R = Fm.shape[1]
    S = stat0.shape[0]
    iv = np.zeros((S, R))
    iv_sigma = np.zeros((S, R))
    diag_idx = np.array([i * R - ((i * (i - 1)) // 2) for i in range(R)])
    for idx in batch_split(S, batch_size):
        s0 = np.array(stat0[idx, :], copy=True)
        s1 = np.array(stat1[idx, :], copy=True)
        e_h, e_hh = e_on_batch(s0, s1, gmm, Fm)
        iv[idx, :] = e_h
        iv_sigma[idx, :] = e_hh[:, diag_idx]
    return iv, iv_sigma


# ----------------------------------------------------------------------------------------
# Synthetic data shared by tests, golden generation and the bench
# ----------------------------------------------------------------------------------------
def synth_frames(n, dim=60, clusters=64, seed=1234, dtype=np.float32):
    """CMVN-like MFCC frames from a fixed random diagonal GMM."""
    rs = np.random.RandomState(seed)
    means = rs.randn(clusters, dim)
    std = np.sqrt(rs.uniform(0.3, 1.0, size=(clusters, dim)))
    lab = rs.randint(0, clusters, size=n)
    x = means[lab] + std[lab] * rs.randn(n, dim)
    return np.ascontiguousarray(x.astype(dtype))


def random_gmm(C, D, seed=7):
    """A valid random UBM: w ~ Dirichlet(5), mu ~ N(0,1), var ~ U(0.5,1.5)."""
    rs = np.random.RandomState(seed)
    w = rs.dirichlet(np.full(C, 5.0))
    mu = rs.randn(C, D)
    var = rs.uniform(0.5, 1.5, size=(C, D))
    return GMM(w, mu, 1.0 / var)


def raw_scales(dim, scale_seed=99):
    """Per-dimension offset (up to +-30, the first column -- c0 / log-energy -- the farthest out) and spread of
    the cluster means of un-normalised, raw-MFCC-like features."""
    rs = np.random.RandomState(scale_seed)
    off = rs.uniform(-30.0, 30.0, size=dim)
    off[0] = -30.0
    spread = rs.uniform(1.0, 6.0, size=dim)
    return off, spread


def synth_raw_frames(n, dim=60, clusters=16, seed=1234, dtype=np.float32, scale_seed=99):
    """Raw-MFCC-like frames, NO cmvn: per-dimension offsets up to +-30, cluster variances log-uniform in
    [1, 100] (the regime in which float32 logits of un-centred frames lose the 1e-4 bar)."""
    off, spread = raw_scales(dim, scale_seed)
    rs = np.random.RandomState(seed)
    means = off[None, :] + spread[None, :] * rs.randn(clusters, dim)
    std = np.sqrt(np.exp(rs.uniform(0.0, np.log(100.0), size=(clusters, dim))))
    lab = rs.randint(0, clusters, size=n)
    x = means[lab] + std[lab] * rs.randn(n, dim)
    return np.ascontiguousarray(x.astype(dtype))


def raw_gmm(C, D, seed=7, scale_seed=99):
    """A UBM for synth_raw_frames data: means around the same offsets, variances log-uniform in [1, 100]."""
    off, spread = raw_scales(D, scale_seed)
    rs = np.random.RandomState(seed)
    w = rs.dirichlet(np.full(C, 5.0))
    mu = off[None, :] + spread[None, :] * rs.randn(C, D)
    var = np.exp(rs.uniform(0.0, np.log(100.0), size=(C, D)))
    return GMM(w, mu, 1.0 / var)


# ----------------------------------------------------------------------------------------
# next row: MAP mean adaptation and GMM-UBM log-likelihood-ratio scoring
# ----------------------------------------------------------------------------------------
def adapt_mean_map_multisession(gmm, modelset, stat0, stat1, r=16.0):
    """MAP-adapted mean super-vectors, one per unique model, statistics summed over the model's sessions
    (sidekit/statserver.py:986-1028 with norm=False, sum_stat_per_model :1206-1226).
    Returns (unique model names, M[n_models x C*D])."""
    modelset = np.asarray(modelset)
    models = np.unique(modelset)
    C, D = gmm.C, gmm.D
    index_map = np.repeat(np.arange(C), D)
    s0 = np.stack([np.asarray(stat0, dtype=F64)[modelset == m].sum(0) for m in models])
    s1 = np.stack([np.asarray(stat1, dtype=F64)[modelset == m].sum(0) for m in models])
    alpha = s0 / (s0 + r)
    with np.errstate(divide="ignore", invalid="ignore"):
        M = s1 / s0[:, index_map]
    M[np.isnan(M)] = 0.0
    M = alpha[:, index_map] * M + (1.0 - alpha[:, index_map]) * np.tile(gmm.mu.reshape(-1), (M.shape[0], 1))
    return models, M


def gmm_scoring(gmm, enroll_modelset, enroll_means, ndx_modelset, ndx_segset, trialmask, segments):
    """Mean per-frame log-likelihood ratio of every trial (sidekit/gmm_scoring.py:56-119): model = UBM with the
    enrolled mean super-vector, denominator = the UBM.  `segments` maps a segment name to its T x D frames.
    Untested cells stay 0."""
    enroll_modelset = list(np.asarray(enroll_modelset))
    scores = np.zeros(np.asarray(trialmask).shape, dtype=F64)
    for ts, seg in enumerate(np.asarray(ndx_segset)):
        cep = np.asarray(segments[str(seg)], dtype=F64)
        ubm_llk = logsumexp_rows(log_posterior(gmm, cep))[1].mean()
        for mi, model in enumerate(np.asarray(ndx_modelset)):
            if not trialmask[mi, ts] or model not in enroll_modelset:
                continue
            mu = enroll_means[enroll_modelset.index(model)]
            scores[mi, ts] = logsumexp_rows(log_posterior(gmm, cep, mu=mu))[1].mean() - ubm_llk
    return scores


def cosine_scoring(enroll_iv, test_iv, wccn=None):
    """Score matrix of sidekit/iv_scoring.py:90-99: optional right-product by `wccn` (statserver.py:682-688),
    length normalisation with the norm clipped at 1e-8 (statserver.py:677-680), enroll . test'."""
    e = np.asarray(enroll_iv, dtype=F64)
    t = np.asarray(test_iv, dtype=F64)
    if wccn is not None:
        e, t = e.dot(wccn), t.dot(wccn)
    e = (e.T / np.clip(np.linalg.norm(e, axis=1), 1e-8, np.inf)).T
    t = (t.T / np.clip(np.linalg.norm(t, axis=1), 1e-8, np.inf)).T
    return e.dot(t.T)


# ----------------------------------------------------------------------------------------
# Feature post-processing in front of the path: FeaturesServer.post_processing
# ----------------------------------------------------------------------------------------
DELTA_FILTER = np.array([.25, .5, .25, 0, -.25, -.5, -.25])      # sidekit/features_server.py:129


def rasta_filt(x):
    """sidekit/frontend/normfeat.py:45-74.  FIR numerator arange(.2, -.3, -.1), one pole at 0.94.  The first
    four frames only prime the FIR state (their outputs are zero) and the recursion starts at frame 4 with
    no feedback from the past: y[n] = sum_k b_k x[n-k] + 0.94 y[n-1], y[3] = 0."""
    from scipy.signal import lfilter
    x = np.asarray(x).T
    numerator = np.arange(.2, -.3, -.1)
    denominator = np.array([1, -0.94])
    zf = np.zeros((x.shape[0], 4))
    for i in range(x.shape[0]):
        _, zf[i, :4] = lfilter(numerator, 1, x[i, :4], axis=-1, zi=[0, 0, 0, 0])
    y = np.zeros(x.shape)
    for i in range(y.shape[0]):
        y[i, 4:] = lfilter(numerator, denominator, x[i, 4:], axis=-1, zi=zf[i, :])[0]
    return y.T


def compute_delta(features, filt=DELTA_FILTER):
    """sidekit/frontend/features.py:136-169 ('filter' method).  The edge-padded copy the reference builds is
    never used: the result is the zero-padded 'same' convolution with the 7-tap filter, stored in a float32
    (PARAM_TYPE) array."""
    features = np.asarray(features)
    win = 3
    delta = np.zeros((features.shape[0] + 2 * win, features.shape[1]), dtype=np.float32)
    for i in range(features.shape[1]):
        delta[:, i] = np.convolve(features[:, i], filt)
    return delta[win:-win, :]


def post_processing(feat, label, mask=None, rasta=False, delta=False, double_delta=False, delta_filter=DELTA_FILTER,
                    feat_norm=None, keep_all_features=True, global_mean=None, global_std=None):
    """FeaturesServer.post_processing (sidekit/features_server.py:202-246) for the options the device path
    covers: mask (:248-258), RASTA (:308-324: the first two labels are copied from the third), delta /
    double delta (:291-306), label_fusion (frontend/vad.py:400-419: the identity on a 1-D label vector,
    whose elements it visits as 0-d arrays), 'cms' / 'cmvn' (frontend/normfeat.py:77-126; statistics over the
    labelled frames, applied to every frame; population standard deviation), frame selection (:241-244).
    Returns (features, labels)."""
    feat = np.array(feat)
    label = np.array(label, dtype=bool)
    if mask is not None:
        feat = feat[:, mask]
    if rasta:
        feat = rasta_filt(feat)
        feat[:2, :] = feat[2, :]
        label[:2] = label[2]
    if delta:
        d1 = compute_delta(feat, filt=delta_filter)
        feat = np.column_stack((feat, d1))
        if double_delta:
            feat = np.column_stack((feat, compute_delta(d1, filt=delta_filter)))
    if feat_norm == 'cms':
        mu = global_mean if global_mean is not None else np.mean(feat[label, :], axis=0)
        feat -= mu
    elif feat_norm == 'cmvn':
        if global_mean is not None and global_std is not None:
            feat -= global_mean
            feat /= global_std
        elif not label.sum() == 0:
            mu = np.mean(feat[label, :], axis=0)
            stdev = np.std(feat[label, :], axis=0)
            feat -= mu
            feat /= stdev
    elif feat_norm in ('cmvn_sliding', 'cms_sliding'):
        cep_sliding_norm(feat, label, win=301, reduce=(feat_norm == 'cmvn_sliding'))
    elif feat_norm is not None:
        raise NotImplementedError(feat_norm)
    if not keep_all_features:
        feat = feat[label]
        label = label[label]
    return feat, label


def cep_sliding_norm(feat, label, win=301, reduce=False):
    """frontend/normfeat.py:194-229 as called by features_server.py:282-287 (center=True): rolling mean and sample
    standard deviation (pandas, the reference's own dependency) over `win` LABELLED frames, centred, the first and last
    win // 2 positions take the nearest full window; only the labelled frames change.  With at most `win` labelled
    frames: whole-session cms / cmvn (every frame changes; normfeat.py:77-126)."""
    import pandas
    if np.sum(label) <= win:
        if reduce:
            if not label.sum() == 0:
                mu, sd = np.mean(feat[label, :], axis=0), np.std(feat[label, :], axis=0)
                feat -= mu
                feat /= sd
        else:
            feat -= np.mean(feat[label, :], axis=0)
        return
    d_win = win // 2
    r = pandas.DataFrame(feat[label, :]).rolling(window=win, center=True)
    mean, std = r.mean().values, r.std().values
    mean[0:d_win, :] = mean[d_win, :]
    mean[-d_win:, :] = mean[-d_win - 1, :]
    std[0:d_win, :] = std[d_win, :]
    std[-d_win:, :] = std[-d_win - 1, :]
    feat[label, :] -= mean
    if reduce:
        feat[label, :] /= std


# ------------------------------------------------------------------ i-vector back-end (next row)
def _classes(modelset):
    """Unique model names in numpy.unique order and the row indices of each."""
    modelset = np.asarray(modelset)
    uniq = np.unique(modelset)
    return uniq, [np.flatnonzero(modelset == u) for u in uniq]


def total_covariance(x):
    """sidekit/statserver.py:800-808."""
    c = x - x.mean(axis=0)
    return np.dot(c.T, c) / x.shape[0]


def within_covariance(modelset, x):
    """sidekit/statserver.py:810-826."""
    W = np.zeros((x.shape[1], x.shape[1]))
    for rows in _classes(modelset)[1]:
        c = x[rows] - x[rows].mean(axis=0)
        W += np.dot(c.T, c)
    return W / x.shape[0]


def between_covariance(modelset, x):
    """sidekit/statserver.py:828-848."""
    B = np.zeros((x.shape[1], x.shape[1]))
    mu = x.mean(axis=0)
    for rows in _classes(modelset)[1]:
        d = x[rows].mean(axis=0) - mu
        B += rows.size * np.outer(d, d)
    return B / x.shape[0]


def lda_matrix(modelset, x, rank):
    """sidekit/statserver.py:850-889 (scipy's eigh of the non-symmetric discrimination matrix reads its lower
    triangle only: kept)."""
    uniq, cls = _classes(modelset)
    mu = x.mean(axis=0)
    Sw = np.zeros((x.shape[1], x.shape[1]))
    means = np.zeros((len(cls), x.shape[1]))
    for k, rows in enumerate(cls):
        c = x[rows] - x[rows].mean(axis=0)
        Sw += np.dot(c.T, c) / rows.size
        means[k] = x[rows].mean(axis=0)
    means = means - mu
    Sb = np.dot(means.T, means)
    ev, evec = scipy.linalg.eigh(np.dot(Sb, scipy.linalg.inv(Sw)).T)
    return evec.real[:, ev.real.argsort()[-rank:][::-1]]


def wccn_choleski(modelset, x):
    """sidekit/statserver.py:901-925."""
    uniq, cls = _classes(modelset)
    W = np.zeros((x.shape[1], x.shape[1]))
    for rows in cls:
        c = x[rows] - x[rows].mean(axis=0)
        W += np.dot(c.T, c) / rows.size
    W = W / len(cls)
    return scipy.linalg.cholesky(scipy.linalg.inv(W)).T


def mean_per_model(modelset, x):
    """sidekit/statserver.py:1228-1245 (first-order part)."""
    uniq, cls = _classes(modelset)
    return uniq, np.array([x[rows].mean(axis=0) for rows in cls])


def whiten_full(x, mu, sigma, is_sqr_inv_sigma=False):
    """sidekit/statserver.py:747-764 for i-vectors (stat0 = 1)."""
    sq = sigma
    if not is_sqr_inv_sigma:
        ev, evec = scipy.linalg.eigh(sigma)
        ind = ev.real.argsort()[::-1]
        sq = np.dot(evec.real[:, ind], np.diag(1 / np.sqrt(ev.real[ind])))
    return np.dot(x - mu, sq)


def whiten_cholesky(x, mu, sigma):
    """sidekit/statserver.py:778-792."""
    return np.dot(x - mu, scipy.linalg.cholesky(scipy.linalg.inv(sigma)).T)


def mahalanobis_scores(enroll_iv, test_iv, m):
    """sidekit/iv_scoring.py:138-142."""
    s = np.zeros((enroll_iv.shape[0], test_iv.shape[0]))
    for i in range(enroll_iv.shape[0]):
        diff = enroll_iv[i] - test_iv
        s[i] = -0.5 * np.sum(np.dot(diff, m) * diff, axis=1)
    return s


def two_covariance_scores(enroll_iv, test_iv, W, B):
    """sidekit/iv_scoring.py:184-198."""
    iW, iB = scipy.linalg.inv(W), scipy.linalg.inv(B)
    G = iW.dot(scipy.linalg.inv(iB + 2 * iW)).dot(iW)
    H = iW.dot(scipy.linalg.inv(iB + iW)).dot(iW)
    s2 = np.sum(np.dot(enroll_iv, H) * enroll_iv, axis=1)
    s3 = np.sum(np.dot(test_iv, H) * test_iv, axis=1)
    S = np.zeros((enroll_iv.shape[0], test_iv.shape[0]))
    for i in range(enroll_iv.shape[0]):
        A = enroll_iv[i] + test_iv
        S[i] = np.sum(np.dot(A, G) * A, axis=1) - s3 - s2[i]
    return S


def fast_plda_scores(enroll_iv, test_iv, mu, F, Sigma, p_known=0.0, scaling_factor=1.0):
    """sidekit/iv_scoring.py:413-469 on already aligned, model-averaged i-vectors."""
    e, t = enroll_iv - mu, test_iv - mu
    inv_sigma = scipy.linalg.inv(Sigma)
    eye = np.eye(F.shape[1])
    K = F.T.dot(inv_sigma * scaling_factor).dot(F)
    K1 = scipy.linalg.inv(K + eye)
    K2 = scipy.linalg.inv(2 * K + eye)
    cst = np.linalg.slogdet(K2)[1] / 2.0 - np.linalg.slogdet(K1)[1]
    s_ac = F.dot(F.T)
    s_tot = s_ac + Sigma
    s_tot_inv = scipy.linalg.inv(s_tot)
    tmp = np.linalg.inv(s_tot - s_ac.dot(s_tot_inv).dot(s_ac))
    phi = s_tot_inv - tmp
    psi = s_tot_inv.dot(s_ac).dot(tmp)
    mp = 0.5 * np.einsum('ij, ji->i', e.dot(phi), e.T)
    sp = 0.5 * np.einsum('ij, ji->i', t.dot(phi), t.T)
    sc = (mp[:, None] + sp + cst + e.dot(psi).dot(t.T)) * scaling_factor
    if p_known != 0:
        N = sc.shape[0]
        ex = np.exp(sc)
        out = np.empty(sc.shape)
        for i in range(N):
            out[i] = sc[i] - np.log(p_known * ex[~(np.arange(N) == i)].sum(axis=0) / (N - 1) + (1 - p_known))
        sc = out
    return sc


def plda_train(modelset, x, rank_f, nb_iter=10, scaling_factor=1.0):
    """FactorAnalyser.plda (sidekit/factor_analyser.py:810-910) for i-vectors x (S x R, stat0 = 1): simplified PLDA, EM
    with minimum divergence.  Returns (mean, F, Sigma).  Per-class loop as fa_model_loop (:168-206, Sigma.ndim == 2
    branch: one inverse per distinct session count)."""
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=0)                                            # :832
    Sigma = total_covariance(x)                                      # :833
    uniq, rows = _classes(modelset)                                  # sum_stat_per_model (statserver.py:1206-1226)
    s0 = np.array([float(len(r)) for r in rows]) * scaling_factor    # :840
    s1 = np.array([x[r].sum(axis=0) for r in rows]) * scaling_factor # :841
    session_per_model = np.array([float(len(r)) for r in rows]) * scaling_factor   # :842
    sigma_obs = total_covariance(x)                                  # :845
    evals, evecs = scipy.linalg.eigh(sigma_obs)
    idx = np.argsort(evals)[::-1]
    F = evecs.real[:, idx[:rank_f]]                                  # :846-849
    K = len(rows)
    for _ in range(nb_iter):
        ev, evec = scipy.linalg.eigh(Sigma)                          # whiten_stat1 and :865-871 use the same factor
        ind = ev.real.argsort()[::-1]
        sqr_inv_sigma = np.dot(evec.real[:, ind], np.diag(1 / np.sqrt(ev.real[ind])))
        w1 = np.dot(s1 - s0[:, None] * mean, sqr_inv_sigma)          # statserver.py:690-697, 747-764
        F = sqr_inv_sigma.T.dot(F)                                   # :871
        A = F.T.dot(F)
        inv_lambda = {n: scipy.linalg.inv(n * A + np.eye(rank_f)) for n in np.unique(s0)}
        e_h = np.zeros((K, rank_f))
        e_hh = np.zeros((K, rank_f, rank_f))
        for i in range(K):
            il = inv_lambda[s0[i]]
            e_h[i] = F.T.dot(w1[i]).dot(il)
            e_hh[i] = il + np.outer(e_h[i], e_h[i])
        _R = e_hh.sum(axis=0) / session_per_model.shape[0]           # :891
        _C = e_h.T.dot(w1).dot(scipy.linalg.inv(sqr_inv_sigma))      # :893
        _A = np.einsum('ijk,i->jk', e_hh, s0)                        # :894
        F = scipy.linalg.solve(_A, _C).T                             # :897
        Sigma = sigma_obs - F.dot(_C) / session_per_model.sum()      # :900
        F = F.dot(scipy.linalg.cholesky(_R))                         # :903
    return mean, F, Sigma


def full_plda_scores(enroll_iv, test_iv, mu, F, G, Sigma, p_known=0.0, scaling_factor=1.0):
    """full_PLDA_scoring (sidekit/iv_scoring.py:266-361) on aligned i-vectors: rows of enroll_iv are the models, rows of
    test_iv the segments of the score matrix."""
    e = np.asarray(enroll_iv, dtype=np.float64) - mu
    t = np.asarray(test_iv, dtype=np.float64) - mu
    invSigma = scipy.linalg.inv(Sigma)
    A = np.linalg.inv(G.T.dot(invSigma * scaling_factor).dot(G) + np.eye(G.shape[1]))
    B = F.T.dot(invSigma * scaling_factor).dot(np.eye(mu.shape[0]) - G.dot(A).dot(G.T).dot(invSigma * scaling_factor))
    K = B.dot(F)
    K1 = scipy.linalg.inv(K + np.eye(F.shape[1]))
    K2 = scipy.linalg.inv(2 * K + np.eye(F.shape[1]))
    constant = np.linalg.slogdet(K2)[1] / 2.0 - np.linalg.slogdet(K1)[1]
    tt, et = B.dot(t.T), B.dot(e.T)
    S1 = np.array([tt[:, i].dot(K1).dot(tt[:, i]) / 2. for i in range(tt.shape[1])])
    score = np.zeros((et.shape[1], tt.shape[1]))
    S2 = np.zeros(et.shape[1])
    for m in range(et.shape[1]):
        both = tt + et[:, m:m + 1]
        S2[m] = et[:, m].dot(K1).dot(et[:, m]) / 2.
        score[m] = np.einsum("ij, ji->i", both.T.dot(K2), both) / 2.
    score += constant - (S1 + S2[:, None])
    score *= scaling_factor
    if p_known != 0:
        N = score.shape[0]
        ex = np.exp(score)
        out = np.empty(score.shape)
        for ii in range(N):
            out[ii] = score[ii] - np.log(p_known * ex[~(np.arange(N) == ii)].sum(axis=0) / (N - 1) + (1 - p_known))
        score = out
    return score
