"""StatServer: container of per-session zero / first order statistics with the method surface of
sidekit.StatServer that the hot path uses (reference: sidekit/statserver.py:207-326, 377-476,
609-697, 732-745).

`accumulate_stat` runs on MI355X: the sessions of a batch are concatenated into one float32
frame matrix in HBM with an offset table, and libskm's Baum-Welch kernel (csrc/skm_gmm.hip,
k_stats in session mode) writes each session's N (C) and F (C*D, component-major) rows in float64.
Back-end algebra of the reference class (LDA / WCCN / NAP / PLDA, statserver.py:800-1820) is out
of scope of this path.
"""
import copy
import logging
import os

import numpy as np

from . import dist
from . import engine as _engine
from . import team as _team
from .idmap import IdMap
from .mixture import Mixture, _h5

STAT_TYPE = np.float64


def sum_log_probabilities(lp, engine=None):
    """Posteriors and the PER-FRAME log-likelihood of a matrix of log-probabilities (sidekit/statserver.py:83-95,
    used by gmm_scoring.py:102-111): returns (pp, log_lk[T])."""
    from .mixture import _row_logsumexp
    return _row_logsumexp(lp, engine)


class StatServer(object):
    def __init__(self, statserver_file_name=None, distrib_nb=0, feature_size=0, index=None, ubm=None,
                 engine=None):
        """Empty, shaped like an IdMap, or read from a file (statserver.py:207-294)."""
        self.modelset = np.empty(0, dtype="|O")
        self.segset = np.empty(0, dtype="|O")
        self.start = np.empty(0, dtype="|O")
        self.stop = np.empty(0, dtype="|O")
        self.stat0 = np.array([], dtype=STAT_TYPE)
        self.stat1 = np.array([], dtype=STAT_TYPE)
        self._engine = engine
        self._resident = None      # device copy of this rank's rows left by accumulate_stat(keep_on_device=True)

        if ubm is not None:
            distrib_nb = ubm.w.shape[0]
            feature_size = ubm.mu.shape[1]

        if statserver_file_name is None:
            pass
        elif isinstance(statserver_file_name, IdMap) or _looks_like_idmap(statserver_file_name):
            self.modelset = statserver_file_name.leftids
            self.segset = statserver_file_name.rightids
            self.start = statserver_file_name.start
            self.stop = statserver_file_name.stop
            self.stat0 = np.empty((self.segset.shape[0], distrib_nb), dtype=STAT_TYPE)
            self.stat1 = np.empty((self.segset.shape[0], distrib_nb * feature_size), dtype=STAT_TYPE)
        elif isinstance(statserver_file_name, str):
            tmp = StatServer.read(statserver_file_name) if index is None \
                else StatServer.read_subset(statserver_file_name, index)
            self.modelset = tmp.modelset
            self.segset = tmp.segset
            self.start = tmp.start
            self.stop = tmp.stop
            self.stat0 = tmp.stat0.astype(STAT_TYPE)
            self.stat1 = tmp.stat1.astype(STAT_TYPE)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = _engine.default_engine()
        return self._engine

    def __deepcopy__(self, memo):
        new = StatServer.__new__(StatServer)
        for k, v in self.__dict__.items():
            setattr(new, k, v if k == "_engine" else (None if k == "_resident" else copy.deepcopy(v, memo)))
        return new

    def __repr__(self):
        ch = '-' * 30 + '\n'
        ch += 'modelset: ' + self.modelset.__repr__() + '\n'
        ch += 'segset: ' + self.segset.__repr__() + '\n'
        ch += 'seg start:' + self.start.__repr__() + '\n'
        ch += 'seg stop:' + self.stop.__repr__() + '\n'
        ch += 'stat0:' + self.stat0.__repr__() + '\n'
        ch += 'stat1:' + self.stat1.__repr__() + '\n'
        ch += '-' * 30 + '\n'
        return ch

    def validate(self, warn=False):
        """statserver.py:308-326."""
        ok = self.modelset.ndim == 1 \
            and (self.modelset.shape == self.segset.shape == self.start.shape == self.stop.shape) \
            and (self.stat0.shape[0] == self.stat1.shape[0] == self.modelset.shape[0]) \
            and (not bool(self.stat1.shape[1] % self.stat0.shape[1]))
        if warn and (self.segset.shape != np.unique(self.segset).shape):
            logging.warning('Duplicated segments in StatServer')
        return bool(ok)

    # ------------------------------------------------------------------ I/O
    # Same dataset names and the same float32 storage of stat0/stat1 as statserver.py:431-455: the
    # float32 rounding is part of the TV input contract.  `.h5` needs h5py; otherwise .npz.
    def write(self, output_file_name, prefix='', mode='w'):
        assert self.validate(), "Error: wrong StatServer format"
        d = os.path.dirname(output_file_name)
        if d and not os.path.exists(d):
            os.makedirs(d)
        start = np.array([-1 if (v is None or (isinstance(v, float) and np.isnan(v))) else int(v)
                          for v in self.start], dtype=np.int64)
        stop = np.array([-1 if (v is None or (isinstance(v, float) and np.isnan(v))) else int(v)
                         for v in self.stop], dtype=np.int64)
        # The reference stores both as int8 (statserver.py:420,424), which wraps frame indices above 127.  Same
        # type whenever it holds the values (always for the all-None start / stop of the experiment scripts), the
        # IdMap file's int32 (bosaris/idmap.py:104,108) otherwise: readers only compare with -1.
        fits = all(a.size == 0 or (a.min() >= -128 and a.max() <= 127) for a in (start, stop))
        start, stop = (a.astype(np.int8 if fits else np.int32) for a in (start, stop))
        data = {prefix + "modelset": np.asarray(self.modelset).astype('S'),
                prefix + "segset": np.asarray(self.segset).astype('S'),
                prefix + "stat0": self.stat0.astype(np.float32),
                prefix + "stat1": self.stat1.astype(np.float32),
                prefix + "start": start, prefix + "stop": stop}
        # mode 'a' on a file that exists: every dataset grows and the new rows go behind the old ones, no check for
        # duplicated entries (statserver.py:408-476)
        grow = mode != "w" and os.path.exists(output_file_name)
        if output_file_name.endswith((".h5", ".hdf5")):
            h5py = _h5()
            with h5py.File(output_file_name, mode) as f:
                if grow and (prefix + "modelset") in f:
                    previous_size = f[prefix + "modelset"].shape[0]
                    for k, v in data.items():
                        f[k].resize((previous_size + v.shape[0],) + v.shape[1:])
                    for k, v in data.items():
                        f[k][previous_size:] = v
                else:
                    for k, v in data.items():
                        f.create_dataset(k, data=v, maxshape=(None,) + v.shape[1:], compression="gzip",
                                         fletcher32=True)
        else:
            if grow:
                z = np.load(output_file_name, allow_pickle=False)
                old = {k: z[k] for k in z.files}
                for k, v in data.items():
                    if k in old:
                        w = max(old[k].dtype.itemsize, v.dtype.itemsize) if v.dtype.kind == "S" else None
                        data[k] = np.concatenate([old.pop(k).astype("S%d" % w), v.astype("S%d" % w)]) if w else \
                            np.concatenate([old.pop(k), v])
                data.update(old)
            with open(output_file_name, "wb") as f:
                np.savez(f, **data)

    @staticmethod
    def _load_arrays(statserver_file_name, prefix=''):
        keys = ("modelset", "segset", "start", "stop", "stat0", "stat1")
        if statserver_file_name.endswith((".h5", ".hdf5")):
            h5py = _h5()
            with h5py.File(statserver_file_name, "r") as f:
                return {k: np.asarray(f[prefix + k][()]) for k in keys}
        z = np.load(statserver_file_name, allow_pickle=False)
        return {k: z[prefix + k] for k in keys}

    @staticmethod
    def read(statserver_file_name, prefix=''):
        a = StatServer._load_arrays(statserver_file_name, prefix)
        ss = StatServer()
        ss.modelset = a["modelset"].astype('U', copy=False)
        ss.segset = a["segset"].astype('U', copy=False)
        ss.start = np.empty(a["start"].shape, '|O')
        ss.stop = np.empty(a["stop"].shape, '|O')
        ss.start[a["start"] != -1] = a["start"][a["start"] != -1]
        ss.stop[a["stop"] != -1] = a["stop"][a["stop"] != -1]
        ss.stat0 = a["stat0"].astype(dtype=STAT_TYPE)
        ss.stat1 = a["stat1"].astype(dtype=STAT_TYPE)
        assert ss.validate(), "Error: wrong StatServer format"
        return ss

    @staticmethod
    def read_subset(statserver_filename, index, prefix=''):
        full = StatServer.read(statserver_filename, prefix)
        if isinstance(index, IdMap) or _looks_like_idmap(index):
            keep = np.array([np.argwhere(full.segset == s)[0][0] for s in index.rightids])
        else:
            keep = np.asarray(index)
        ss = StatServer()
        ss.modelset, ss.segset = full.modelset[keep], full.segset[keep]
        ss.start, ss.stop = full.start[keep], full.stop[keep]
        ss.stat0, ss.stat1 = full.stat0[keep, :], full.stat1[keep, :]
        return ss

    # ------------------------------------------------------------------ Baum-Welch statistics
    def accumulate_stat(self, ubm, feature_server, seg_indices=None, channel_extension=("", "_b"), num_thread=1,
                        max_batch_frames=8 * 1024 * 1024, gather=True, keep_on_device=False):
        """Zero and first order statistics of the listed sessions (statserver.py:609-667).

        Host side: per session, load frames, cut to [start, stop), keep VAD-selected rows
        (statserver.py:645-652).  Device side: everything else.  `num_thread` is accepted and ignored.

        Multi-GPU (torch.distributed initialised): the sessions are dealt to the ranks like numpy.array_split and
        every rank computes and stores ITS rows only -- there is no reduction on this path (each row has one
        owner).  `gather=True` (default: every rank returns the reference's full host arrays) then all-gathers the
        row blocks, each rank sending its own rows; `gather=False` leaves the other ranks' rows untouched
        (`self.local_rows` lists the rows this rank filled).

        `keep_on_device=True` also keeps this rank's rows in HBM, rounded to float32 as StatServer.write stores
        them (statserver.py:439-446), so that FactorAnalyser.total_variability / extract_ivectors given this
        object start from them without a device->host->device round trip (49 GB at 50k sessions x 2048 x 60).
        """
        assert isinstance(ubm, Mixture), 'First parameter has to be a Mixture'
        assert hasattr(feature_server, "load"), 'Second parameter has to be a FeaturesServer'
        team = _team._active_team()
        if team is not None:
            return self._accumulate_stat_team(team, ubm, feature_server, seg_indices, channel_extension, num_thread,
                                              max_batch_frames, keep_on_device)

        if (seg_indices is None) \
                or (self.stat0.shape[0] != self.segset.shape[0]) \
                or (self.stat1.shape[0] != self.segset.shape[0]):
            self.stat0 = np.zeros((self.segset.shape[0], ubm.distrib_nb()), dtype=STAT_TYPE)
            self.stat1 = np.zeros((self.segset.shape[0], ubm.sv_size()), dtype=STAT_TYPE)
            seg_indices = range(self.segset.shape[0])
        feature_server.keep_all_features = True

        seg_indices = [int(i) for i in seg_indices]
        mine = seg_indices
        if dist.is_dist():
            a, b = dist.shard_bounds(len(seg_indices))
            mine = seg_indices[a:b]
        self.local_rows = np.asarray(mine, dtype=np.int64)
        self._resident = None

        eng = self.engine
        torch = eng.torch
        packed = ubm._pack()              # centred on the UBM's mean when it sits far from the origin (raw cepstra)
        C, D = ubm.distrib_nb(), ubm.dim()
        kept = []                         # (rows, stat0 f32, stat1 f32) device blocks when keep_on_device

        # The copy off the device is a pipeline stage of its own: a worker thread drains finished batches on a second
        # stream (engine.to_host_rows: pinned double buffers + threaded host copy) while this thread uploads and
        # launches the next batch.  Host float64 statistics are 2 KB per frame at C = 2048: the API is bound by that
        # transfer, so it must never wait behind, or make wait, the kernels.
        pipelined = hasattr(eng, "to_host_rows") and hasattr(eng, "copy_stream") and str(eng.device).startswith("cuda")
        pending = []
        if pipelined:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1)
            cstream = eng.copy_stream()

        def drain(rows, s0, s1, ready):
            with torch.cuda.stream(cstream):
                cstream.wait_event(ready)
                eng.to_host_rows(s0, self.stat0, rows)
                eng.to_host_rows(s1, self.stat1, rows)
                cstream.synchronize()

        def store(rows, s0, s1):
            if keep_on_device:
                kept.append((list(rows), s0.to(torch.float32), s1.to(torch.float32)))
            if pipelined:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(eng.device))
                s0.record_stream(cstream)
                s1.record_stream(cstream)
                while len(pending) >= 2:                    # at most two batches of statistics wait in HBM
                    pending.pop(0).result()
                pending.append(pool.submit(drain, list(rows), s0, s1, ready))
            elif hasattr(eng, "to_host_rows"):      # pinned, pipelined copy off the device (engine.to_host_rows)
                eng.to_host_rows(s0, self.stat0, rows)
                eng.to_host_rows(s1, self.stat1, rows)
            else:
                self.stat0[rows, :] = eng.to_host(s0)
                self.stat1[rows, :] = eng.to_host(s1)

        def finish_stores(abort=False):
            """Waits for the drain jobs (re-raising what a worker raised) and stops the worker thread.  `abort`: the
            main loop failed -- jobs that have not started are cancelled, the running one is awaited so that nothing
            writes into stat0 / stat1 after this method has returned, and their own errors are dropped in favour
            of the one being raised."""
            try:
                while pending:
                    job = pending.pop(0)
                    if abort:
                        job.cancel()
                        try:
                            job.result()
                        except BaseException:
                            pass
                    else:
                        job.result()
            finally:
                if pipelined:
                    for job in pending:
                        job.cancel()
                    pool.shutdown(wait=True)

        batch, batch_idx, nframes = [], [], 0

        def run_batch():
            if not batch_idx:
                return
            lens = np.array([b.shape[0] for b in batch], dtype=np.int64)
            seg_off = np.concatenate([[0], np.cumsum(lens)])
            if hasattr(eng, "upload_sessions") and not any(type(b).__module__.startswith("torch") for b in batch):
                X = eng.upload_sessions(batch)          # pinned, pipelined; no host concatenation
            else:
                X = eng.upload_frames(np.concatenate(batch, axis=0) if len(batch) > 1 else batch[0])
            s0, s1 = eng.bw_stats(X, seg_off, packed, own_frames=True)
            store(batch_idx, s0, s1)

        def show_of(idx):
            show = self.segset[idx]
            channel = 0
            if feature_server.features_extractor is not None and show.endswith(channel_extension[1]):
                channel = 1
            return show[:show.rfind(channel_extension[channel])], channel

        try:
            # Device data plane: a features server that post-processes on the GPU (skm_feat_postproc) hands over
            # whole batches of sessions already VAD-selected in HBM -- same result as load() + cep[vad]
            # (statserver.py:645-652: statistics over the labelled frames of the fully normalised session).
            if getattr(feature_server, "has_post_processing", lambda: False)() and hasattr(feature_server, "load_batch") \
                    and all(self.start[i] is None and self.stop[i] is None for i in mine):
                todo = list(mine)
                while todo:
                    take, nfr = [], 0
                    while todo and (not take or nfr < max_batch_frames):
                        i = todo.pop(0)
                        take.append(i)
                        nfr += feature_server.features[show_of(i)[0]].shape[0]
                    X, seg_off, _ = feature_server.load_batch([show_of(i)[0] for i in take], keep_all_features=False)
                    if not ubm.dim() == X.shape[1]:
                        raise Exception('dimension of ubm and features differ: {:d} / {:d}'.format(ubm.dim(), X.shape[1]))
                    s0, s1 = eng.bw_stats(X, seg_off, packed)
                    store(take, s0, s1)
                mine_loop = []
            else:
                mine_loop = mine

            for idx in mine_loop:
                show, channel = show_of(idx)
                cep, vad = feature_server.load(show, channel=channel,
                                               input_feature_filename=feature_server.feature_filename_structure)
                if type(cep).__module__.startswith("torch"):
                    cep = cep.detach().cpu().numpy()
                stop = vad.shape[0] if self.stop[idx] is None else min(self.stop[idx], vad.shape[0])
                data = cep[self.start[idx]:stop, :]
                sel = vad[self.start[idx]:stop]
                if not sel.all():               # the reference's data[vad] copies even when every frame is kept
                    data = data[sel, :]
                if not ubm.dim() == data.shape[1]:
                    raise Exception('dimension of ubm and features differ: {:d} / {:d}'.format(ubm.dim(), data.shape[1]))
                batch.append(np.asarray(data, dtype=np.float32))
                batch_idx.append(idx)
                nframes += data.shape[0]
                if nframes >= (min(max_batch_frames, 1 << 20) if pipelined else max_batch_frames):
                    run_batch()             # pipelined: batches of about a million frames keep all three stages busy
                    batch, batch_idx, nframes = [], [], 0
            run_batch()
        except BaseException:
            finish_stores(abort=True)
            raise
        finish_stores()

        if keep_on_device and kept:
            rows = [r for blk in kept for r in blk[0]]
            self._resident = {"rows": np.asarray(rows, dtype=np.int64),
                              "stat0": torch.cat([blk[1] for blk in kept], 0) if len(kept) > 1 else kept[0][1],
                              "stat1": torch.cat([blk[2] for blk in kept], 0) if len(kept) > 1 else kept[0][2],
                              "host": (self.stat0, self.stat1)}
            self._resident["mark"] = self._fingerprint(self._resident["rows"])
        del kept

        if dist.is_dist() and gather:
            self._gather_rows(seg_indices)

    def _accumulate_stat_team(self, team, ubm, feature_server, seg_indices, channel_extension, num_thread,
                              max_batch_frames, keep_on_device):
        """One process, several GPUs (team.py): the host arrays are allocated once, on the caller's object, and every
        rank thread fills ITS rows of them through its own GPU -- disjoint rows of shared arrays, so the all-gather of
        the multi-process path has nothing to do."""
        if (seg_indices is None) \
                or (self.stat0.shape[0] != self.segset.shape[0]) \
                or (self.stat1.shape[0] != self.segset.shape[0]):
            self.stat0 = np.zeros((self.segset.shape[0], ubm.distrib_nb()), dtype=STAT_TYPE)
            self.stat1 = np.zeros((self.segset.shape[0], ubm.sv_size()), dtype=STAT_TYPE)
            seg_indices = range(self.segset.shape[0])
        seg_indices = [int(i) for i in seg_indices]
        keep = self._engine

        def body(rank):
            me = self if rank == 0 else copy.copy(self)
            me._engine = team.engine(rank)
            me.accumulate_stat(_team.private_copy(ubm, team, rank), _team.private_copy(feature_server, team, rank),
                               seg_indices, channel_extension, num_thread, max_batch_frames, gather=False,
                               keep_on_device=keep_on_device)
        try:
            team.run(body)
        finally:
            self._engine = keep
        feature_server.keep_all_features = True
        self.local_rows = np.asarray(seg_indices, dtype=np.int64)      # the caller's object holds every row

    def _gather_rows(self, seg_indices, chunk_bytes=1 << 30):
        """All-gather of the row blocks the ranks own (each rank sends ITS rows, in chunks of about chunk_bytes so
        the staging buffers stay small); afterwards every rank holds every requested row on the host."""
        eng = self.engine
        torch = eng.torch
        world = dist.world_size()
        bounds = [dist.shard_bounds(len(seg_indices), world, r) for r in range(world)]
        a, b = bounds[dist.rank()]
        dev = getattr(eng, "device", "cpu")
        for name in ("stat0", "stat1"):
            arr = getattr(self, name)
            per = max(1, int(chunk_bytes // max(1, arr.shape[1] * arr.itemsize)))
            longest = max(hi - lo for lo, hi in bounds)
            for k0 in range(0, longest, per):
                spans = [(min(lo + k0, hi), min(lo + k0 + per, hi)) for lo, hi in bounds]
                la, lb = spans[dist.rank()]
                local = torch.from_numpy(np.ascontiguousarray(arr[seg_indices[la:lb], :])).to(dev)
                full = dist.allgather_rows(local, [hi - lo for lo, hi in spans]).cpu().numpy()
                pos = 0
                for r, (lo, hi) in enumerate(spans):
                    if r != dist.rank() and hi > lo:
                        arr[seg_indices[lo:hi], :] = full[pos:pos + hi - lo]
                    pos += hi - lo

    def _fingerprint(self, rows):
        """Per-row mark of the host statistics behind a resident copy: the row sums of stat0 and of ~256 evenly spaced
        columns of stat1.  An in-place edit of the host arrays (ss.stat1 -= m, ss.stat1[i] = ..., ss.stat0 *= k) keeps
        their identity but not this."""
        step = max(1, self.stat1.shape[1] // 256)
        return np.asarray(self.stat0[rows].sum(axis=1) + self.stat1[rows, ::step].sum(axis=1))

    def resident_rows(self):
        """(rows, stat0 float32, stat1 float32) device copy of this rank's rows left by
        accumulate_stat(keep_on_device=True), or None when there is none or the host arrays were replaced or
        edited in place since (the device copy is then dropped: the host arrays are what the caller sees)."""
        r = self._resident
        if r is None:
            return None
        if r["host"][0] is not self.stat0 or r["host"][1] is not self.stat1 \
                or not np.array_equal(r["mark"], self._fingerprint(r["rows"]), equal_nan=True):
            self._resident = None
            return None
        return r["rows"], r["stat0"], r["stat1"]

    def release_device(self):
        self._resident = None

    # ------------------------------------------------------------------ small algebra used around the TV path
    def get_mean_stat1(self):
        return np.mean(self.stat1, axis=0)

    def norm_stat1(self):
        vect_norm = np.clip(np.linalg.norm(self.stat1, axis=1), 1e-08, np.inf)
        self.stat1 = (self.stat1.transpose() / vect_norm).transpose()

    def center_stat1(self, mu):
        """statserver.py:690-697."""
        dim = self.stat1.shape[1] // self.stat0.shape[1]
        index_map = np.repeat(np.arange(self.stat0.shape[1]), dim)
        self.stat1 = self.stat1 - (self.stat0[:, index_map] * mu.astype(STAT_TYPE))

    def whiten_stat1(self, mu, sigma, isSqrInvSigma=False):
        """Diagonal (statserver.py:743-745) and single full-covariance (:747-764) whitening; the R x R eigen-decomposition
        is host LAPACK like the reference's, the rotation of the S rows a device GEMM."""
        if sigma.ndim == 1:
            self.center_stat1(mu)
            self.stat1 = self.stat1 / np.sqrt(sigma.astype(STAT_TYPE))
        elif sigma.ndim == 2:
            sqr_inv_sigma = sigma
            if not isSqrInvSigma:
                import scipy.linalg
                eigen_values, eigen_vectors = scipy.linalg.eigh(sigma)
                ind = eigen_values.real.argsort()[::-1]
                sqr_inv_sigma = np.dot(eigen_vectors.real[:, ind], np.diag(1 / np.sqrt(eigen_values.real[ind])))
            self.center_stat1(mu)
            self.rotate_stat1(sqr_inv_sigma)
        else:
            raise Exception('full-covariance UBM whitening (sigma.ndim == 3) is not on this path')

    def whiten_cholesky_stat1(self, mu, sigma):
        """statserver.py:778-798."""
        if sigma.ndim == 2:
            import scipy.linalg
            chol_invcov = scipy.linalg.cholesky(scipy.linalg.inv(sigma)).T
            self.center_stat1(mu)
            self.rotate_stat1(chol_invcov)
        elif sigma.ndim == 1:
            self.center_stat1(mu)
            self.stat1 = self.stat1 / np.sqrt(sigma)
        else:
            raise Exception('Wrong dimension of Sigma, must be 1 or 2')

    # ------------------------------------------------------------------ i-vector back-end
    # Every scatter matrix below is ONE float64 MFMA GEMM Xc' Xc over centred (and 1/sqrt(n)-scaled) rows; class
    # means and the centring are skm_class_sums / skm_rows_center over the sessions sorted by speaker.  The R x R
    # inverse / Cholesky / eigen-decomposition that follow stay host LAPACK, like the reference's.
    def _class_layout(self):
        uniq, inv = np.unique(self.modelset, return_inverse=True)
        order = np.argsort(inv, kind="stable").astype(np.int64)
        counts = np.bincount(inv, minlength=uniq.shape[0]).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        return uniq, inv[order].astype(np.int64), order, counts, off

    def _class_centred(self, weighted):
        """(device X, class means K x R, rows sorted by class minus their class mean [times 1/sqrt(n_k)], counts)."""
        eng = self.engine
        uniq, cls_sorted, order, counts, off = self._class_layout()
        X = eng._f64_dev(self.stat1)
        means = eng.class_sums(X, off, src=order, mean=True)
        scale = (1.0 / np.sqrt(counts.astype(np.float64)))[cls_sorted] if weighted else None
        return X, means, eng.rows_center(X, Mu=means, cls=cls_sorted, src=order, scale=scale), counts

    def get_total_covariance_stat1(self):
        """statserver.py:800-808."""
        eng = self.engine
        X = eng._f64_dev(self.stat1)
        Xc = eng.rows_center(X, Mu=eng.col_mean(X))
        return eng.to_host(eng.gram(Xc, alpha=1.0 / self.stat1.shape[0]))

    def get_within_covariance_stat1(self):
        """statserver.py:810-826."""
        eng = self.engine
        Xc = self._class_centred(False)[2]
        return eng.to_host(eng.gram(Xc, alpha=1.0 / self.stat1.shape[0]))

    def get_between_covariance_stat1(self):
        """statserver.py:828-848: sum_k n_k (m_k - mu)(m_k - mu)' / S."""
        eng = self.engine
        uniq, cls_sorted, order, counts, off = self._class_layout()
        X = eng._f64_dev(self.stat1)
        means = eng.class_sums(X, off, src=order, mean=True)
        Mc = eng.rows_center(means, Mu=eng.col_mean(X), scale=np.sqrt(counts.astype(np.float64)))
        return eng.to_host(eng.gram(Mc, alpha=1.0 / self.stat1.shape[0]))

    def get_lda_matrix_stat1(self, rank):
        """statserver.py:850-889 (scipy's eigh on the non-symmetric discrimination matrix, as the reference)."""
        import scipy.linalg
        eng = self.engine
        X, means, Xc, counts = self._class_centred(True)
        Sw = eng.to_host(eng.gram(Xc))
        Mc = eng.rows_center(means, Mu=eng.col_mean(X))
        Sb = eng.to_host(eng.gram(Mc))
        eigen_values, eigen_vectors = scipy.linalg.eigh(np.dot(Sb, scipy.linalg.inv(Sw)).transpose())
        idx = eigen_values.real.argsort()[-rank:][::-1]
        return eigen_vectors.real[:, idx]

    def get_mahalanobis_matrix_stat1(self):
        """statserver.py:891-899."""
        import scipy.linalg
        return scipy.linalg.inv(self.get_within_covariance_stat1())

    def get_wccn_choleski_stat1(self):
        """statserver.py:901-925."""
        import scipy.linalg
        eng = self.engine
        X, means, Xc, counts = self._class_centred(True)
        WCCN = eng.to_host(eng.gram(Xc, alpha=1.0 / counts.shape[0]))
        return scipy.linalg.cholesky(scipy.linalg.inv(WCCN)).T

    def mean_stat_per_model(self):
        """statserver.py:1228-1245."""
        eng = self.engine
        uniq, cls_sorted, order, counts, off = self._class_layout()
        out = StatServer(engine=self._engine)
        out.modelset = uniq
        out.segset = out.modelset
        out.start = np.empty(out.segset.shape, '|O')
        out.stop = np.empty(out.segset.shape, '|O')
        out.stat0 = eng.to_host(eng.class_sums(eng._f64_dev(self.stat0), off, src=order, mean=True))
        out.stat1 = eng.to_host(eng.class_sums(eng._f64_dev(self.stat1), off, src=order, mean=True))
        return out

    def _take(self, indx):
        self.segset = self.segset[indx]
        self.modelset = self.modelset[indx]
        self.start = self.start[indx]
        self.stop = self.stop[indx]
        self.stat0 = self.stat0[indx, :]
        self.stat1 = self.stat1[indx, :]

    def align_segments(self, segment_list):
        """Rows re-ordered (and possibly dropped) to follow `segment_list` (statserver.py:579-592)."""
        self._take(np.array([np.argwhere(self.segset == v)[0][0] for v in segment_list], dtype=np.int64))

    def align_models(self, model_list):
        """Rows re-ordered to follow `model_list`, first session of each model (statserver.py:594-607)."""
        self._take(np.array([np.argwhere(self.modelset == v)[0][0] for v in model_list], dtype=np.int64))

    def rotate_stat1(self, R):
        """stat1 <- stat1 . R (statserver.py:682-688), a float64 device GEMM."""
        eng = self.engine
        self.stat1 = eng.to_host(eng.gemm(eng._f64_dev(self.stat1), eng._f64_dev(R)))

    # ------------------------------------------------------------------ MAP adaptation (feeds gmm_scoring)
    # C*D element-wise float64 updates on already-reduced statistics: host, like the GMM M-step; the per-model sums are
    # device class sums.
    def get_model_stat0(self, mod_id):
        return self.stat0[self.modelset == mod_id, :]

    def get_model_stat1(self, mod_id):
        return self.stat1[self.modelset == mod_id, :]

    def sum_stat_per_model(self):
        """Statistics summed over the sessions of every unique model (statserver.py:1206-1226)."""
        out = StatServer(engine=self._engine)
        out.modelset = np.unique(self.modelset)
        out.segset = out.modelset
        out.stat0 = np.zeros((out.modelset.shape[0], self.stat0.shape[1]), dtype=STAT_TYPE)
        out.stat1 = np.zeros((out.modelset.shape[0], self.stat1.shape[1]), dtype=STAT_TYPE)
        out.start = np.empty(out.segset.shape, '|O')
        out.stop = np.empty(out.segset.shape, '|O')
        # one launch per statistic over the sessions sorted by model (skm_class_sums) instead of a masked sum per model
        eng = self.engine
        uniq, cls_sorted, order, counts, off = self._class_layout()
        out.stat0[...] = eng.to_host(eng.class_sums(eng._f64_dev(self.stat0), off, src=order))
        out.stat1[...] = eng.to_host(eng.class_sums(eng._f64_dev(self.stat1), off, src=order))
        session_per_model = counts.astype(np.float64)
        return out, session_per_model

    def _map_means(self, ubm, stat0, stat1, alpha, norm):
        index_map = np.repeat(np.arange(ubm.distrib_nb()), ubm.dim())
        with np.errstate(divide="ignore", invalid="ignore"):
            M = stat1 / stat0[:, index_map]
        M[np.isnan(M)] = 0  # empty components keep the UBM mean
        M = alpha[:, index_map] * M + (1 - alpha[:, index_map]) * np.tile(ubm.get_mean_super_vector(), (M.shape[0], 1))
        if norm:
            # Kullback-Leibler normalisation, diagonal covariance (statserver.py:974-980)
            M = M * np.sqrt(np.repeat(ubm.w, ubm.dim()) * ubm.get_invcov_super_vector())
        return M.astype(dtype=STAT_TYPE)

    def adapt_mean_map(self, ubm, r=16, norm=False):
        """One MAP-adapted mean super-vector per session (statserver.py:946-984)."""
        out = StatServer(engine=self._engine)
        out.modelset, out.segset, out.start, out.stop = self.modelset, self.segset, self.start, self.stop
        out.stat0 = np.ones((self.segset.shape[0], 1), dtype=STAT_TYPE)
        eps = np.finfo(np.float32).eps
        alpha = (self.stat0 + eps) / (self.stat0 + eps + r)
        out.stat1 = self._map_means(ubm, self.stat0, self.stat1, alpha, norm)
        return out

    def adapt_mean_map_multisession(self, ubm, r=16, norm=False):
        """One MAP-adapted mean super-vector per unique model, statistics summed over its sessions
        (statserver.py:986-1028)."""
        models = np.unique(self.modelset)
        out = StatServer(engine=self._engine)
        out.modelset = models
        out.segset = np.unique(self.modelset)
        out.start = np.empty(models.shape, dtype="|O")
        out.stop = np.empty(models.shape, dtype="|O")
        out.stat0 = np.ones((models.shape[0], 1), dtype=STAT_TYPE)
        per_model = self.sum_stat_per_model()[0]
        alpha = per_model.stat0 / (per_model.stat0 + r)
        out.stat1 = self._map_means(ubm, per_model.stat0, per_model.stat1, alpha, norm)
        return out


def _looks_like_idmap(obj):
    return all(hasattr(obj, k) for k in ("leftids", "rightids", "start", "stop"))
