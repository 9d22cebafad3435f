"""Mixture: diagonal-covariance GMM with the method surface of sidekit.Mixture
(reference: sidekit/mixture.py), E-step on MI355X.

What runs where
  * E-step (`_expectation`, `_expectation_list`, and therefore `EM_split` / `EM_uniform`):
    hand-written gfx950 kernels through libskm (csrc/skm_gmm.hip); frames live in HBM as float32
    for the whole training run instead of being re-read from disk every iteration
    (mixture.py:586-588).
  * M-step, variance control, split, `_compute_all`: float64 NumPy on the host.  They are C x D
    element-wise updates on already-reduced statistics and the split takes a
    discrete arg-max that must be bit-robust, so they stay in float64; on a multi-GPU job every
    rank runs them identically on the all-reduced statistics (no broadcast, replaces
    sidekit_mpi.py:401,514).
  * Multi-GPU: one process per GPU; frames are cut like numpy.array_split over ranks
    (sidekit_mpi.py:408-410) unless the features server says it is already sharded; the packed
    [N | F | S | llk] float64 buffer is summed with ONE all-reduce (RCCL) per iteration.

Only the diagonal-covariance path exists (north star); full-covariance branches of the
reference (mixture.py:441-459, 563-565, 604-610, 826-909) are out of scope.
"""
import copy
import logging
import os

import numpy as np

from . import dist
from .team import fanout
from . import engine as _engine


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _row_logsumexp(lp, engine=None):
    """(pp, log_lk[T]) of a T x C matrix of log-probabilities, float64, on the GPU (skm_row_logsumexp)."""
    eng = engine or _engine.default_engine()
    lp = np.asarray(lp, dtype=np.float64)
    if lp.ndim != 2:
        raise ValueError("sum_log_probabilities wants a T x C matrix")
    dev = eng.torch.from_numpy(np.ascontiguousarray(lp)).to(eng.device)
    pp, log_lk = eng.row_logsumexp(dev)
    return eng.to_host(pp), eng.to_host(log_lk)


def sum_log_probabilities(lp, engine=None):
    """Posteriors and the SUMMED log-likelihood of a matrix of log-probabilities (sidekit/mixture.py:52-64):
    returns (pp, llk).  The training path never forms `lp` (the E-step kernels fuse this); this free function is for
    callers that import it by name."""
    pp, log_lk = _row_logsumexp(lp, engine)
    return pp, log_lk.sum()


class Mixture(object):
    """Gaussian mixture model storage + EM training (sidekit/mixture.py:67-197)."""

    def __init__(self, mixture_file_name='', name='empty', engine=None):
        self.w = np.array([])
        self.mu = np.array([])
        self.invcov = np.array([])
        self.invchol = np.array([])
        self.cov_var_ctl = np.array([])
        self.cst = np.array([])
        self.det = np.array([])
        self.name = name
        self.A = 0
        self._engine = engine
        if mixture_file_name != '':
            self.read(mixture_file_name)

    # ------------------------------------------------------------------ plumbing
    @property
    def engine(self):
        if self._engine is None:
            self._engine = _engine.default_engine()
        return self._engine

    def __deepcopy__(self, memo):
        new = Mixture.__new__(Mixture)
        for k, v in self.__dict__.items():
            setattr(new, k, v if k == "_engine" else copy.deepcopy(v, memo))
        return new

    def __add__(self, other):
        """Weights, means and inverse covariances are added (mixture.py:203-212)."""
        new_mixture = Mixture(engine=self._engine)
        new_mixture.w = self.w + other.w
        new_mixture.mu = self.mu + other.mu
        new_mixture.invcov = self.invcov + other.invcov
        return new_mixture

    def _serialize(self):
        """Shared-memory serialisation of the reference (mixture.py:236-271) is not needed: the
        accumulators live in HBM and are combined by all-reduce."""
        return None

    def get_distrib_nb(self):
        return self.w.shape[0]

    def distrib_nb(self):
        return self.w.shape[0]

    def dim(self):
        return self.mu.shape[1]

    def sv_size(self):
        return self.mu.shape[1] * self.w.shape[0]

    def get_mean_super_vector(self):
        return self.mu.flatten()

    def get_invcov_super_vector(self):
        assert self.invcov.ndim == 2, 'Must be diagonal co-variance.'
        return self.invcov.flatten()

    def validate(self):
        """Shape consistency (mixture.py:398-422), diagonal case."""
        ok = (self.w.ndim == 1)
        ok &= (self.det.ndim == 1)
        ok &= (self.cst.ndim == 1)
        ok &= (self.mu.ndim == 2)
        ok &= (self.invcov.ndim == 2)
        if not ok:
            return False
        ok &= (self.w.shape[0] == self.mu.shape[0])
        ok &= (self.w.shape[0] == self.cst.shape[0])
        ok &= (self.w.shape[0] == self.det.shape[0])
        ok &= (self.invcov.shape == self.mu.shape)
        return bool(ok)

    # ------------------------------------------------------------------ I/O (npz; HDF5 when h5py exists)
    _FIELDS = ("w", "mu", "invcov", "invchol", "cov_var_ctl", "cst", "det")

    def write(self, mixture_file_name, prefix='', mode='w'):
        """Same dataset names as mixture.py:325-359.  `.h5` goes through h5py or libhdf5 (_h5), anything else is an .npz."""
        d = os.path.dirname(mixture_file_name)
        if d and not os.path.exists(d):
            os.makedirs(d)
        data = {prefix + k: np.asarray(getattr(self, k), dtype="d") for k in self._FIELDS}
        data[prefix + "a"] = np.asarray(self.A, dtype="d")
        if mixture_file_name.endswith((".h5", ".hdf5")):
            h5py = _h5()
            with h5py.File(mixture_file_name, mode) as f:
                for k, v in data.items():
                    f.create_dataset(k, v.shape, "d", v, compression="gzip", fletcher32=True)
        else:
            with open(mixture_file_name, "wb") as f:
                np.savez(f, **data)

    def read(self, mixture_file_name, prefix=''):
        if mixture_file_name.endswith((".h5", ".hdf5")):
            h5py = _h5()
            with h5py.File(mixture_file_name, 'r') as f:
                get = lambda k: np.asarray(f[prefix + k][()])
                vals = {k: get(k) for k in self._FIELDS + ("a",)}
        else:
            z = np.load(mixture_file_name, allow_pickle=False)
            vals = {k: z[prefix + k] for k in self._FIELDS + ("a",)}
        for k in self._FIELDS:
            setattr(self, k, vals[k])
        self.w = self.w.reshape(-1)
        self.A = vals["a"]

    # ------------------------------------------------------------------ parameters
    def _compute_all(self):
        """det, cst, A (mixture.py:382-396)."""
        assert self.invcov.ndim == 2, "only diagonal covariance is supported on this path"
        self.det = 1.0 / np.prod(self.invcov, axis=1)
        self.cst = 1.0 / (np.sqrt(self.det) * (2.0 * np.pi) ** (self.dim() / 2.0))
        self.A = (np.square(self.mu) * self.invcov).sum(1) - 2.0 * (np.log(self.w) + np.log(self.cst))

    def _reset(self):
        """All values to zero (mixture.py:508-515)."""
        self.cst.fill(0.0)
        self.det.fill(0.0)
        self.w.fill(0.0)
        self.mu.fill(0.0)
        self.invcov.fill(0.0)
        self.A = 0.0

    @staticmethod
    def variance_control(cov, flooring, ceiling, cov_ctl):
        """Inclusive floor / ceiling (mixture.py:489-506); modifies and returns `cov`."""
        floor = flooring * cov_ctl
        ceil = ceiling * cov_ctl
        to_floor = np.less_equal(cov, floor)
        to_ceil = np.greater_equal(cov, ceil)
        cov[to_floor] = floor[to_floor]
        cov[to_ceil] = ceil[to_ceil]
        return cov

    def _split_ditribution(self):
        """Each Gaussian becomes two, shifted by +-sigma along its largest-variance axis
        (mixture.py:517-535; the reference's spelling of the name is kept)."""
        sigma = 1.0 / self.invcov
        sig_max = np.max(sigma, axis=1)
        arg_max = np.argmax(sigma, axis=1)
        shift = np.zeros(self.mu.shape)
        shift[np.arange(arg_max.shape[0]), arg_max] = np.sqrt(sig_max)
        self.mu = np.vstack((self.mu - shift, self.mu + shift))
        self.invcov = np.vstack((self.invcov, self.invcov))
        self.w = np.concatenate([self.w, self.w]) * 0.5
        self.cst = np.zeros(self.w.shape)
        self.det = np.zeros(self.w.shape)
        self.cov_var_ctl = np.vstack((self.cov_var_ctl, self.cov_var_ctl))
        self._compute_all()

    def _maximization(self, accum, ceil_cov=10, floor_cov=1e-2):
        """M-step from the statistics in `accum` (mixture.py:590-611)."""
        self.w = accum.w / np.sum(accum.w)
        self.mu = accum.mu / accum.w[:, np.newaxis]
        cov = accum.invcov / accum.w[:, np.newaxis] - np.square(self.mu)
        cov = Mixture.variance_control(cov, floor_cov, ceil_cov, self.cov_var_ctl)
        self.invcov = 1.0 / cov
        self._compute_all()

    # ------------------------------------------------------------------ E-step on the GPU
    def _frame_model(self):
        """(shift, ref_var) that condition the float32 logits of this model (the reference forms them in float64,
        mixture.py:482-486; both leave posteriors and statistics unchanged, see include/skm.h):
        shift    offset to centre the frames on when the model sits far from the origin (un-normalised features),
                 else None: the model's own mean sum_c w_c mu_c against its total standard deviation per
                 dimension (engine.centering_offset);
        ref_var  per-dimension second moment of the whole model about that offset -- the variance of the
                 reference Gaussian the statistics kernels' image is packed against (skm_gmm_pack_ref)."""
        off, ref = _engine.stats_conditioning(self.w, self.mu, self.invcov)
        return (None if off is None else self.engine.frame_shift(off)), ref

    def _auto_shift(self):
        return self._frame_model()[0]

    def _pack(self, shift="auto", ref_var="auto", stats=True):
        """Device image of this model.  stats=True: for the statistics kernels (E-step, Baum-Welch), packed against
        the reference Gaussian; stats=False: for per-frame log-likelihoods / log-probabilities (plain)."""
        if isinstance(shift, str) or (stats and isinstance(ref_var, str)):
            auto_shift, auto_ref = self._frame_model()
            if isinstance(shift, str):
                shift = auto_shift
            if isinstance(ref_var, str):
                ref_var = auto_ref
        return self.engine.pack(self.w, self.mu, self.invcov, shift=shift, ref_var=ref_var if stats else None)

    def _device_frames(self, cep):
        """float32 device view of `cep` (uploads a host array; device tensors pass through)."""
        return self.engine.upload_frames(cep)

    def _estep_device(self, X_dev, packed=None):
        """[N | F | S | llk] of the frames resident in HBM, as a float64 device tensor."""
        if packed is None:
            packed = self._pack()
        return self.engine.estep(X_dev, packed)

    def _unpack_stats(self, flat):
        C, D = self.distrib_nb(), self.dim()
        flat = np.asarray(flat, dtype=np.float64)
        N = flat[:C]
        F = flat[C:C + C * D].reshape(C, D)
        S = flat[C + C * D:C + 2 * C * D].reshape(C, D)
        return N, F, S, float(flat[-1])

    def _expectation(self, accum, cep):
        """E-step over one block of frames; statistics are ADDED to `accum`; returns the summed
        log-likelihood (mixture.py:537-568)."""
        if not _is_torch(cep) and np.asarray(cep).ndim == 1:
            cep = np.asarray(cep)[:, np.newaxis]
        X = self._device_frames(cep)
        flat = self.engine.to_host(self._estep_device(X))
        N, F, S, llk = self._unpack_stats(flat)
        accum.w += N
        accum.mu += F
        accum.invcov += S
        return llk

    def _expectation_list(self, stat_acc, feature_list, feature_server, llk_acc=np.zeros(1), num_thread=1):
        """E-step over a list of feature files (mixture.py:570-588).  `num_thread` is accepted for
        compatibility and ignored: the parallelism is the GPU's, and the reference's multi-process
        version is racy.  One device accumulation over all files, one
        device->host copy."""
        stat_acc._reset()
        feature_server.keep_all_features = False
        packed = self._pack()
        out = None
        for feat in feature_list:
            X = self._device_frames(feature_server.load(feat)[0])
            out = self.engine.estep(X, packed, out=out, accumulate=out is not None, unshift=False)
        if out is None:
            return
        if packed.shift is not None:        # the correction is linear: once, on the sum over the files
            self.engine.stats_unshift_(out, packed.C, packed.D, packed.shift)
        N, F, S, llk = self._unpack_stats(self.engine.to_host(out))
        stat_acc.w += N
        stat_acc.mu += F
        stat_acc.invcov += S
        llk_acc[0] += llk

    # ------------------------------------------------------------------ training drivers
    def _resident_frames(self, features_server, feature_list):
        """All training frames of THIS rank as one float32 tensor in HBM.  In a multi-GPU job the rows of the
        stacked matrix are cut like numpy.array_split (sidekit_mpi.py:408-410); a server that can tell the frame
        count of a file without loading it (`frame_counts`) lets every rank load only the files that overlap its
        row range instead of stacking everything on every host."""
        if not dist.is_dist() or getattr(features_server, "is_sharded", False):
            # host files go straight into the pinned staging buffers (engine.upload_sessions): no np.vstack of the
            # whole training set in pageable memory in front of the transfer
            if hasattr(self.engine, "upload_sessions") and hasattr(features_server, "has_post_processing") \
                    and not features_server.has_post_processing():
                features_server._blocks_wanted = True
                try:
                    blocks = features_server.stack_features_parallel(feature_list, num_thread=1)
                finally:
                    features_server._blocks_wanted = False
                if isinstance(blocks, list) and blocks and all(isinstance(b, np.ndarray) for b in blocks):
                    return self.engine.upload_sessions(blocks)
                if isinstance(blocks, list):
                    torch = self.engine.torch
                    blocks = [self._device_frames(b) for b in blocks]
                    return torch.cat(blocks, dim=0) if len(blocks) > 1 else blocks[0]
                return self._device_frames(blocks)
            return self._device_frames(features_server.stack_features_parallel(feature_list, num_thread=1))
        feature_list = list(feature_list)
        counts = features_server.frame_counts(feature_list) if hasattr(features_server, "frame_counts") else None
        if counts is None:
            feats = features_server.stack_features_parallel(feature_list, num_thread=1)
            a, b = dist.shard_bounds(feats.shape[0])
            return self._device_frames(feats[a:b])
        off = np.concatenate([[0], np.cumsum(np.asarray(counts, dtype=np.int64))])
        a, b = dist.shard_bounds(int(off[-1]))
        first = int(np.searchsorted(off, a, side="right")) - 1
        last = int(np.searchsorted(off, b, side="left"))          # files [first, last) overlap rows [a, b)
        if b <= a:
            first, last = 0, 1                                    # an empty shard still needs the frame dimension
        feats = features_server.stack_features_parallel(feature_list[first:last], num_thread=1)
        return self._device_frames(feats[a - int(off[first]):max(b, a) - int(off[first])] if b > a else feats[:0])

    def _global_stats(self, X_dev, packed=None):
        """E-step on the local shard + all-reduce -> host (N, F, S, llk)."""
        out = self._estep_device(X_dev, packed)
        dist.allreduce_sum_(out)
        return self._unpack_stats(self.engine.to_host(out))

    def _init(self, features_server, feature_list, num_thread=1, X_dev=None):
        """Single Gaussian whose 'covariance' is the raw second moment E[x^2] -- a quirk of the
        reference that is kept (mixture.py:613-641)."""
        if X_dev is None:
            X_dev = self._resident_frames(features_server, feature_list)
        mom = self.engine.moments(X_dev)
        mom = self.engine.torch.cat([mom, self.engine.torch.tensor([float(X_dev.shape[0])], dtype=mom.dtype,
                                                                   device=mom.device)])
        dist.allreduce_sum_(mom)
        mom = self.engine.to_host(mom)
        D = X_dev.shape[1]
        n_frames = mom[-1]
        mu = mom[:D] / n_frames
        cov = mom[D:2 * D] / n_frames
        self.mu = mu[None]
        self.invcov = 1. / cov[None]
        self.w = np.asarray([1.0])
        self.cst = np.zeros(self.w.shape)
        self.det = np.zeros(self.w.shape)
        self.cov_var_ctl = 1.0 / copy.deepcopy(self.invcov)
        self._compute_all()
        # un-normalised features: the EM loop runs on frames centred on the global mean (identical on every rank:
        # it comes from the all-reduced moments)
        off = _engine.centering_offset(mu, np.sqrt(np.maximum(cov - np.square(mu), 0.0)))
        self._em_shift = None if off is None else self.engine.frame_shift(off)
        sh = 0.0 if self._em_shift is None else self._em_shift.host
        self._em_ref_var = _ref_var(cov - 2.0 * sh * mu + np.square(sh))   # E[(x - shift)^2] of all the frames
        return X_dev

    @fanout
    def EM_split(self,
                 features_server,
                 feature_list,
                 distrib_nb,
                 iterations=(1, 2, 2, 4, 4, 4, 4, 8, 8, 8, 8, 8, 8),
                 num_thread=1,
                 llk_gain=0.01,
                 save_partial=False,
                 output_file_name="ubm",
                 ceil_cov=10,
                 floor_cov=1e-2):
        """Binary-split EM (mixture.py:643-733).  Returns the per-iteration mean log-likelihoods.

        `num_thread` and `llk_gain` are accepted and unused, as in the reference (its gain test is
        commented out, mixture.py:711-731)."""
        llk = []
        X_dev = self._init(features_server, feature_list, num_thread)
        shift, ref_var = self._em_shift, self._em_ref_var
        if shift is not None:
            X_dev = self.engine.center_frames(X_dev, shift)
        selected = False

        for it in iterations[:int(np.log2(distrib_nb))]:
            if save_partial and dist.rank() == 0:
                ext = ".h5" if _has_h5py() else ".npz"
                self.write('{}_{}g{}'.format(output_file_name, self.get_distrib_nb(), ext), prefix='')
            self._split_ditribution()
            if not selected:
                # _init stacks the frames with the server's own keep_all_features setting, but the E-steps of the
                # reference load every file with keep_all_features = False (mixture.py:585) and leave it False:
                # with VAD labels they see the labelled frames only
                selected = True
                if getattr(features_server, "keep_all_features", False):
                    features_server.keep_all_features = False
                    same = getattr(features_server, "selection_is_identity", None)
                    if not (same is not None and same(feature_list)):
                        del X_dev
                        X_dev = self._resident_frames(features_server, feature_list)
                        if shift is not None:
                            X_dev = self.engine.center_frames(X_dev, shift)
            llk.extend(self._em_iterations_device(X_dev, it, ceil_cov, floor_cov, shift=shift, ref_var=ref_var))
        return llk

    def _em_iterations_device(self, X_dev, it, ceil_cov=10, floor_cov=1e-2, shift=None, ref_var="auto"):
        """`it` EM iterations of the loop body mixture.py:686-710 with the model resident on the GPU: E-step,
        all-reduce of the packed statistics, M-step + variance control + re-pack (skm_gmm_mstep_pack) -- no host
        round trip inside the loop.  The parameters and the per-iteration mean log-likelihoods come back once."""
        eng = self.engine
        loop = _DeviceEM(self, X_dev, ceil_cov, floor_cov, shift=shift, ref_var=ref_var)
        llk_dev = eng.torch.empty(it, dtype=eng.torch.float64, device=eng.device)
        for i in range(it):
            loop.step(llk_mean=llk_dev[i:i + 1])
        loop.download()
        return [float(v) for v in eng.to_host(llk_dev)]

    @fanout
    def EM_uniform(self, cep, distrib_nb, iteration_min=3, iteration_max=10, llk_gain=0.01, do_init=True):
        """EM from a uniform-slice start with an early stop on the gain (mixture.py:735-794)."""
        llk = []
        n_total = cep.shape[0]
        if do_init:
            self._init_uniform(cep, distrib_nb)
        local = cep
        if dist.is_dist():
            a, b = dist.shard_bounds(n_total)
            local = cep[a:b]
        X_dev = self._device_frames(local)
        shift, ref_var = self._frame_model()          # from the initial model: the same on every rank
        if shift is not None:
            X_dev = self.engine.center_frames(X_dev, shift)
        # the model stays on the GPU; only the summed log-likelihood (8 bytes) is read per iteration for the gain test
        loop = _DeviceEM(self, X_dev, 10, 1e-2, shift=shift, ref_var=ref_var)
        for i in range(0, iteration_max):
            stats = loop.step()
            llk.append(float(self.engine.to_host(stats[-1:])[0]) / n_total)
            if i > 0:
                gain = llk[-1] - llk[-2]
                if gain < llk_gain and i >= iteration_min:
                    logging.debug('EM (break) distrib_nb: %d %i/%d gain: %f -- %s, %d',
                                  self.mu.shape[0], i + 1, iteration_max, gain, self.name, n_total)
                    break
        loop.download()
        return llk

    def _init_uniform(self, cep, distrib_nb):
        """Start of EM_uniform with the reference's quirks kept (mixture.py:796-824): every slice
        ends at the last frame, rows after the first store the covariance (not its inverse) in
        `invcov`, and `mu` is 1-D when distrib_nb == 1.  One pass over <= distrib_nb suffixes of the
        data; done on the host in float64 (initialisation, not on the iterated path)."""
        if _is_torch(cep):
            cep = cep.detach().cpu().numpy()
        cep = np.asarray(cep, dtype=np.float64)
        nb = cep.shape[0]
        self.w = np.full(distrib_nb, 1.0 / distrib_nb, "d")
        self.cst = np.zeros(distrib_nb, "d")
        self.det = np.zeros(distrib_nb, "d")
        for i in range(0, distrib_nb):
            start = nb // distrib_nb * i
            end = max(start + 10, nb)
            mean = np.mean(cep[start:end, :], axis=0)
            cov = (cep[start:end, :] ** 2).mean(0)
            if i == 0:
                self.mu = mean
                self.invcov = 1. / cov[None]
            else:
                self.mu = np.vstack((self.mu, mean))
                self.invcov = np.vstack((self.invcov, cov))
        self.cov_var_ctl = 1.0 / copy.deepcopy(self.invcov)
        self._compute_all()

    # ------------------------------------------------------------------ T x C matrix (mixture.py:461-487)
    def compute_log_posterior_probabilities(self, cep, mu=None):
        """ln(w_c N(x_t; mu_c, Sigma_c)) for every frame and component, T x C float64, computed on the
        GPU (float32 MFMA, widened on store).  `mu` replaces the means (a C x D array or a C*D
        super-vector, the MAP-adapted model of gmm_scoring), exactly as mixture.py:474-479 does by
        recomputing the data-independent term A from it."""
        cep = np.asarray(cep) if not _is_torch(cep) else cep
        if cep.ndim == 1:
            cep = cep[None, :]
        if mu is None:
            packed = self._pack(stats=False)
        else:
            packed = self.engine.pack(self.w, np.asarray(mu, dtype=np.float64).reshape(self.mu.shape), self.invcov,
                                      shift=self._auto_shift())
        X = self._device_frames(cep)
        return self.engine.to_host(self.engine.logprob(X, packed))

    # ------------------------------------------------------------------ scoring helper
    def log_likelihood(self, cep):
        """Per-frame log sum_c w_c N(x_t) on the GPU: the `log_lk` of sum_log_probabilities
        (statserver.py:83-95) without materialising the T x C matrix."""
        X = self._device_frames(cep)
        return self.engine.to_host(self.engine.loglik(X, self._pack(stats=False))).astype(np.float64)



class _DeviceEM(object):
    """EM iterations with the model resident on the GPU (Mixture.EM_split / EM_uniform): step() = E-step over the
    rank's frames, all-reduce of the packed statistics, M-step + variance control + re-pack on the device
    (skm_gmm_mstep_pack); download() brings the parameters back and refreshes det / cst / A."""

    def __init__(self, mixture, X_dev, ceil_cov, floor_cov, shift=None, ref_var="auto"):
        """`shift`: X_dev holds frames centred on it (engine.center_frames); the loop then lives in the centred
        coordinates -- the M-step is translation invariant (mu' = F'/N, cov = S'/N - mu'^2, same w, same llk) -- and
        download() moves the means back."""
        eng = mixture.engine
        self.m, self.eng, self.X = mixture, eng, X_dev
        self.ceil, self.floor = ceil_cov, floor_cov
        self.shift = shift
        self.packed = mixture._pack(shift=shift, ref_var=ref_var)
        self.ctl = eng.torch.from_numpy(np.ascontiguousarray(mixture.cov_var_ctl, dtype=np.float64)).to(eng.device)
        self.params = self.stats = None

    def step(self, llk_mean=None):
        """One iteration; returns the all-reduced statistics [N | F | S | llk] it was made from (device)."""
        self.stats = self.eng.estep(self.X, self.packed, out=self.stats, centered=True, unshift=False)
        dist.allreduce_sum_(self.stats)
        self.params = self.eng.mstep_pack(self.stats, self.ctl, self.packed, self.ceil, self.floor,
                                          params=self.params, llk_mean=llk_mean)
        return self.stats

    def download(self):
        if self.params is None:
            return
        m = self.m
        C, D = m.distrib_nb(), m.dim()
        flat = self.eng.to_host(self.params)
        m.w = flat[:C].copy()
        m.mu = flat[C:C + C * D].reshape(C, D).copy()
        if self.shift is not None:
            m.mu += self.shift.host[None, :]
        m.invcov = flat[C + C * D:C + 2 * C * D].reshape(C, D).copy()
        m._compute_all()


def _ref_var(second_moment):
    """Variance vector of the reference Gaussian, or None when SKM_REF=0 switches the reference packing off (A/B)."""
    if os.environ.get("SKM_REF", "1") == "0":
        return None
    return np.asarray(second_moment, dtype=np.float64)


def _h5():
    """The HDF5 module of this process: h5py when it is installed, else the same calls on libhdf5 through ctypes
    (h5lite).  Raises ImportError when neither exists: use .npz file names there."""
    try:
        import h5py
        return h5py
    except ImportError:
        from . import h5lite
        if not h5lite.available():
            raise ImportError("neither h5py nor libhdf5 is available: use .npz file names")
        return h5lite


def _has_h5py():
    """True when .h5 products can be written and read (h5py, or libhdf5 through h5lite)."""
    try:
        _h5()
        return True
    except ImportError:
        return False
