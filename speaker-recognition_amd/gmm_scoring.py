"""GMM-UBM log-likelihood-ratio scoring with the surface of sidekit/gmm_scoring.py (SURVEY.md 8f rank 1:
the first "next" row after the statistics path -- it re-uses the log-likelihood kernel with the `mu=`
override of Mixture.compute_log_posterior_probabilities, mixture.py:474-479).

Reference loop (gmm_scoring.py:83-119): for every test segment, for every enrolled model of its trials,
a T x C log-probability matrix, a row log-sum-exp and a mean; then the same for the UBM.  Here every test
segment's frames sit in HBM once (one float32 matrix + an offset table), and for the UBM and for each
enrolled model ONE pass of libskm's log-likelihood kernel (skm_gmm_loglik: MFMA logits + row log-sum-exp,
the T x C matrix is never stored) runs over all of them, followed by a per-segment float64 mean
(skm_segment_mean).  The score matrix leaves the device once.
"""
import numpy as np

from . import dist
from .idmap import Ndx, Scores
from .mixture import Mixture
from .statserver import StatServer


def _segment_frames(feature_server, segset):
    blocks, lens = [], []
    for seg in segset:
        cep = feature_server.load(seg)[0]
        if type(cep).__module__.startswith("torch"):
            cep = cep.detach().cpu().numpy()
        cep = np.asarray(cep, dtype=np.float32)
        blocks.append(cep)
        lens.append(cep.shape[0])
    return blocks, np.asarray(lens, dtype=np.int64)


def gmm_scoring_singleThread(ubm, enroll, ndx, feature_server, score_mat, seg_idx=None):
    """Fills score_mat[m, ts] for the trials of the listed test segments (gmm_scoring.py:56-119)."""
    assert isinstance(ubm, Mixture), 'First parameter should be a Mixture'
    assert isinstance(enroll, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be a Ndx'
    assert hasattr(feature_server, "load"), 'Fourth parameter should be a FeatureServer'
    assert ubm.invcov.ndim == 2, "only diagonal covariance is on this path"
    if seg_idx is None:
        seg_idx = range(ndx.segset.shape[0])
    seg_idx = [int(i) for i in seg_idx]
    if not seg_idx:
        return
    eng = ubm.engine
    if getattr(feature_server, "has_post_processing", lambda: False)() and hasattr(feature_server, "load_batch"):
        # raw frames are post-processed on the GPU in one launch and stay there (what load() returns per segment)
        X, seg_off, _ = feature_server.load_batch(list(ndx.segset[seg_idx]))
        if X.shape[0] and X.shape[1] != ubm.dim():
            raise Exception('dimension of ubm and features differ: {:d} / {:d}'.format(ubm.dim(), X.shape[1]))
        if seg_off[-1] == 0:
            return
    else:
        blocks, lens = _segment_frames(feature_server, ndx.segset[seg_idx])
        for b in blocks:
            if b.shape[1] != ubm.dim():
                raise Exception('dimension of ubm and features differ: {:d} / {:d}'.format(ubm.dim(), b.shape[1]))
        seg_off = np.concatenate([[0], np.cumsum(lens)])
        if seg_off[-1] == 0:
            return
        X = eng.upload_frames(np.concatenate(blocks, axis=0))
    seg_off_dev = eng.torch.as_tensor(seg_off).to(eng.device)
    # un-normalised features: every model is packed around the UBM's mean and the frames are centred once
    # (log-likelihoods are translation invariant; Mixture._auto_shift)
    shift = ubm._auto_shift()
    if shift is not None:
        X = eng.center_frames(X, shift)
    ubm_mean = eng.segment_mean(eng.loglik(X, ubm._pack(shift=shift, stats=False), centered=True), seg_off_dev)

    # models that take part in a trial of these segments AND are enrolled (gmm_scoring.py:87-93)
    mask = np.asarray(ndx.trialmask)[:, seg_idx]
    enr_index = {k: i for i, k in enumerate(enroll.modelset)}
    rows, llr = [], []
    for m, model in enumerate(ndx.modelset):
        if not mask[m].any() or model not in enr_index:
            continue
        packed = eng.pack(ubm.w, np.asarray(enroll.stat1[enr_index[model], :], dtype=np.float64).reshape(ubm.mu.shape),
                          ubm.invcov, shift=shift)
        rows.append(m)
        llr.append(eng.segment_mean(eng.loglik(X, packed, centered=True), seg_off_dev) - ubm_mean)
    if not rows:
        return
    llr = eng.to_host(eng.torch.stack(llr))             # [models x segments]
    for r, m in enumerate(rows):
        cols = np.flatnonzero(mask[m])
        score_mat[m, np.asarray(seg_idx)[cols]] = llr[r, cols]


def gmm_scoring(ubm, enroll, ndx, feature_server, num_thread=1):
    """Log-likelihood ratios of all the trials of `ndx` (gmm_scoring.py:122-185); returns a Scores object.
    `num_thread` is accepted and ignored.  With torch.distributed initialised the test segments are dealt
    to the ranks like numpy.array_split (the reference's thread split, gmm_scoring.py:166) and the score
    matrix is summed over ranks (every column has one owner)."""
    assert isinstance(ubm, Mixture), 'First parameter should be a Mixture'
    assert isinstance(enroll, StatServer), 'Second parameter should be a StatServer'
    assert isinstance(ndx, Ndx), 'Third parameter should be a Ndx'
    assert hasattr(feature_server, "load"), 'Fourth parameter should be a FeatureServer'
    # keep the enrolled models and the test segments the server can deliver (gmm_scoring.py:146-155)
    have = getattr(feature_server, "features", None)
    segs = ndx.segset if have is None else np.array([s for s in ndx.segset if s in have], dtype=ndx.segset.dtype)
    clean_ndx = ndx.filter(enroll.modelset, segs, True)

    s = np.zeros(clean_ndx.trialmask.shape)
    n_seg = clean_ndx.segset.shape[0]
    a, b = dist.shard_bounds(n_seg) if dist.is_dist() else (0, n_seg)
    gmm_scoring_singleThread(ubm, enroll, clean_ndx, feature_server, s, range(a, b))
    if dist.is_dist():
        t = ubm.engine.torch.as_tensor(s).to(getattr(ubm.engine, "device", "cpu"))
        dist.allreduce_sum_(t)
        s = t.cpu().numpy()

    score = Scores()
    score.scoremat = s
    score.modelset = clean_ndx.modelset
    score.segset = clean_ndx.segset
    score.scoremask = clean_ndx.trialmask
    return score
