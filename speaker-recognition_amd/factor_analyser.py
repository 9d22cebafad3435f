"""FactorAnalyser: total-variability training and i-vector extraction with the method surface of
sidekit.FactorAnalyser (reference: sidekit/factor_analyser.py:209-808), on MI355X.

The reference streams batches of sessions from an HDF5 StatServer file through a process pool
(e_worker / e_gather, factor_analyser.py:95-128, 621-649).  Here the whitened statistics of this
rank's sessions live in HBM for the whole training run and every EM iteration is a handful of
batched MFMA GEMMs (csrc/skm_tv.hip).  With torch.distributed initialised the sessions are cut
over the ranks like numpy.array_split (sidekit_mpi.py:118-123), the accumulators _A, _C, _R are
summed with all-reduce (replaces the Reduce calls of sidekit_mpi.py:186-214) and every rank runs
the same M-step.  PLDA training (factor_analyser.py:810-910), the step after extraction, is `plda` below.
"""
import copy
import os

import numpy as np

from . import dist
from . import engine as _engine
from .team import fanout
from .mixture import Mixture, _has_h5py, _h5
from .statserver import StatServer, STAT_TYPE


def _load_stat_arrays(src, prefix=''):
    """(modelset, segset, start, stop, stat0, stat1) with stat0/stat1 as STORED (float32 for files,
    which is what the reference's TV code consumes, statserver.py:439-446)."""
    if isinstance(src, StatServer):
        # float64 in memory; the rounding to float32 that a file would apply happens where the rows are uploaded
        # (engine.upload_rows(..., float32): on the device, and only for the rank's rows)
        return (src.modelset, src.segset, src.start, src.stop, np.asarray(src.stat0), np.asarray(src.stat1))
    a = StatServer._load_arrays(src, prefix)
    start = np.empty(a["start"].shape, '|O')
    stop = np.empty(a["stop"].shape, '|O')
    start[a["start"] != -1] = a["start"][a["start"] != -1]
    stop[a["stop"] != -1] = a["stop"][a["stop"] != -1]
    return (a["modelset"], a["segset"], start, stop,
            np.asarray(a["stat0"], dtype=np.float32), np.asarray(a["stat1"], dtype=np.float32))


def _upload_f32(eng, a):
    """Rows of stored statistics on the device as float32 (the storage type of a StatServer file)."""
    if hasattr(eng, "upload_rows"):
        return eng.upload_rows(a, eng.torch.float32)
    return eng.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(eng.device)


# Sessions per device batch of extract_ivectors: no e_hh / A accumulation there, so the batch is bounded by scratch only
# (10.5 GB of float32 matrices at R = 400) and larger batches mean fewer, fuller launches of the per-session chain
# (bench tv leg, 50 000 sessions: 2.15e5 i-vectors/s at 4096, 2.21e5 at 16384).  Training stays at 4096: the e_hh buffer
# of a batch is one operand of the A accumulation and must stay inside the 32-bit extent of a buffer descriptor.
EXTRACT_BATCH = 16384


class _FtPlanes(object):
    """f~ of a set of sessions as the rows image of its bf16 x 3 planes (engine.tv_whiten_planes): what
    extract_ivectors keeps instead of the float32 matrix."""

    def __init__(self, planes):
        self.planes = planes


class FactorAnalyser(object):
    def __init__(self, input_file_name=None, mean=None, F=None, G=None, H=None, Sigma=None, engine=None):
        self.mean = None
        self.F = None
        self.G = None
        self.H = None
        self.Sigma = None
        self._engine = engine
        # precision of the TV E-step: "auto" (a bound on cond(L_s) decides per call, _estep_chain), "f32", "f64"
        # (float64 Cholesky / inverse chain only) or "precise" (Gram matrices, L build, f~ T and chain in float64)
        self.chain_precision = os.environ.get("SKM_TV_PRECISION", "auto")
        if input_file_name is not None:
            fa = FactorAnalyser.read(input_file_name)
            self.mean, self.F, self.G, self.H, self.Sigma = fa.mean, fa.F, fa.G, fa.H, fa.Sigma
        if mean is not None:
            self.mean = mean
        if F is not None:
            self.F = F
        if G is not None:
            self.G = G
        if H is not None:
            self.H = H
        if Sigma is not None:
            self.Sigma = Sigma

    @property
    def engine(self):
        if self._engine is None:
            self._engine = _engine.default_engine()
        return self._engine

    # ------------------------------------------------------------------ I/O (fa/* layout of factor_analyser.py:264-324)
    _KEYS = (("mean", "fa/mean"), ("F", "fa/f"), ("G", "fa/g"), ("H", "fa/h"), ("Sigma", "fa/sigma"))

    def write(self, output_file_name):
        d = os.path.dirname(output_file_name)
        if d and not os.path.exists(d):
            os.makedirs(d)
        kind = np.zeros(5, dtype="int16")
        data = {}
        for i, (attr, key) in enumerate(self._KEYS):
            v = getattr(self, attr)
            if v is not None:
                kind[i] = 1
                data[key] = np.asarray(v)
        data["fa/kind"] = kind
        if output_file_name.endswith((".h5", ".hdf5")):
            h5py = _h5()
            with h5py.File(output_file_name, "w") as fh:
                for k, v in data.items():
                    fh.create_dataset(k, data=v, compression="gzip", fletcher32=True)
        else:
            with open(output_file_name, "wb") as f:
                np.savez(f, **{k.replace("/", "__"): v for k, v in data.items()})

    @staticmethod
    def read(input_filename):
        fa = FactorAnalyser()
        if input_filename.endswith((".h5", ".hdf5")):
            h5py = _h5()
            with h5py.File(input_filename, "r") as fh:
                kind = np.asarray(fh["fa/kind"][()])
                for i, (attr, key) in enumerate(FactorAnalyser._KEYS):
                    if kind[i] != 0:
                        setattr(fa, attr, np.asarray(fh[key][()]))
        else:
            z = np.load(input_filename, allow_pickle=False)
            kind = z["fa__kind"]
            for i, (attr, key) in enumerate(FactorAnalyser._KEYS):
                if kind[i] != 0:
                    setattr(fa, attr, z[key.replace("/", "__")])
        return fa

    @staticmethod
    def _ext():
        return ".h5" if _has_h5py() else ".npz"

    # ------------------------------------------------------------------ device helpers
    def _resident_stats(self, sources, ubm, prefix='', rows_planes=False):
        """Whitened statistics of THIS rank's sessions in HBM: (ftilde, naug, total sessions, local slice).
        rows_planes (extraction from ONE source): ftilde comes back as _FtPlanes -- the rows image of its bf16 x 3
        planes, written by the whitening itself (engine.tv_whiten_planes) -- when the engine offers that."""
        eng = self.engine
        torch = eng.torch
        rows_planes = rows_planes and len(sources) == 1 and hasattr(eng, "tv_whiten_planes")

        def whiten(d0, d1):
            if rows_planes:
                planes, naug = eng.tv_whiten_planes(d0, d1, ubm.get_mean_super_vector(), ubm.get_invcov_super_vector())
                if planes is not None:
                    return _FtPlanes(planes), naug
            return eng.tv_whiten(d0, d1, ubm.get_mean_super_vector(), ubm.get_invcov_super_vector())
        fts, naugs, total, local = [], [], 0, []
        for src in sources:
            res = src.resident_rows() if isinstance(src, StatServer) else None
            if res is not None:
                # statistics left in HBM by accumulate_stat(keep_on_device=True), already rounded to float32 as the
                # file would store them: no host round trip, provided they are exactly this rank's shard
                n = src.stat0.shape[0]
                a, b = dist.shard_bounds(n) if dist.is_dist() else (0, n)
                if np.array_equal(res[0], np.arange(a, b)):
                    local.append((total + a, total + b))
                    total += n
                    if b > a:
                        ft, naug = whiten(res[1], res[2])
                        fts.append(ft)
                        naugs.append(naug)
                    continue
            _, _, _, _, s0, s1 = _load_stat_arrays(src, prefix)
            n = s0.shape[0]
            a, b = dist.shard_bounds(n) if dist.is_dist() else (0, n)
            local.append((total + a, total + b))
            total += n
            if b > a:
                d0 = _upload_f32(eng, s0[a:b])
                d1 = _upload_f32(eng, s1[a:b])
                ft, naug = whiten(d0, d1)
                del d0, d1
                fts.append(ft)
                naugs.append(naug)
        if not fts:
            return None, None, total, local
        ft = fts[0] if len(fts) == 1 else torch.cat(fts, 0)
        naug = naugs[0] if len(naugs) == 1 else torch.cat(naugs, 0)
        return ft, naug, total, local

    # cond(L_s) above which the float32 products of the E-step no longer give cosine >= 0.9999 with room to spare
    # (tools/tv_long_sessions.py: e_h is off by ~cond * 1e-8 normwise whatever the chain's type; cond 3.6e5 -> 1 - cos
    # = 9e-6, cond 3e6 -> 7e-4).  The bound below over-estimates cond(L_s) when the components' loadings do not align.
    PRECISE_ABOVE = 3.0e5

    @staticmethod
    def chain_cond_bound(eng, T_dev, naug, C, D):
        """Upper bound of cond(L_s) over the sessions of `naug`: L_s = I + sum_c N_sc T_c'T_c has eigenvalues >= 1 and
        lambda_max(L_s) <= 1 + sum_c N_sc lambda_max(T_c T_c') -- the D x D Gram matrices' largest eigenvalues by 20
        power iterations (a few small batched float64 products), then one matrix-vector product with N."""
        torch = eng.torch
        if not hasattr(torch, "bmm") or naug is None:
            return 0.0
        Tc = T_dev.view(C, D, -1)
        B = torch.bmm(Tc, Tc.transpose(1, 2))                       # C x D x D
        v = torch.ones((C, D, 1), dtype=B.dtype, device=B.device)
        for _ in range(20):
            v = torch.bmm(B, v)
            v = v / v.norm(dim=1, keepdim=True).clamp_min(1e-300)
        lam = 1.1 * torch.bmm(v.transpose(1, 2), torch.bmm(B, v)).reshape(C)      # Rayleigh quotient (+10 %)
        n = naug[:, :C].to(torch.float64)
        return 1.0 + float((n @ lam).max().item())

    def _estep_chain(self, T_dev, naug, C, D):
        """(chain, T64) arguments of engine.tv_estep for this call.  "auto": identical on every rank of a multi-GPU
        job (the bound is max-reduced), because the ranks' accumulators are summed."""
        mode = self.chain_precision
        if mode == "auto":
            eng = self.engine
            bound = self.chain_cond_bound(eng, T_dev, naug, C, D) if hasattr(eng, "lib") else 0.0
            if dist.is_dist():
                t = eng.torch.tensor([bound], dtype=eng.torch.float64, device=getattr(eng, "device", "cpu"))
                dist.allreduce_max_(t)
                bound = float(t.item())
            mode = "precise" if bound > self.PRECISE_ABOVE else None
        if mode == "precise":
            return "precise", T_dev
        return (mode if mode in ("f32", "f64") else None), None

    def _tv_iteration(self, ft, naug, T_dev, C, D, R, total_sessions, min_div, batch, ft_planes=None):
        """One EM iteration on the device; returns the new loading matrix (device, float64).  ft_planes: the
        three-term bf16 planes of ft (engine.tv_ftilde_planes), built once per training run, or None."""
        eng = self.engine
        torch = eng.torch
        T32, G = eng.tv_prepare(T_dev, C, D)
        A_aug = torch.zeros((C + 1, R * R), dtype=torch.float64, device=eng.device)
        Cacc = torch.zeros((R, C * D), dtype=torch.float64, device=eng.device)
        chain, T64 = self._estep_chain(T_dev, naug, C, D)
        if ft is not None:
            more = {"ft_planes": ft_planes} if ft_planes is not None else {}
            eng.tv_estep(naug, ft, T32, G, C, D, batch=batch, A_aug=A_aug, Cacc=Cacc, chain=chain, T64=T64, **more)
        del G
        world = dist.world_size()
        sharded = dist.is_dist() and C % world == 0
        if sharded:
            # component-sharded M-step: reduce-scatter the accumulators so that rank g owns
            # components [g*C/W, (g+1)*C/W), all-reduce only _R (R*R doubles), solve locally, all-gather F rows.
            Cl = C // world
            r_dev = A_aug[C].clone()
            dist.allreduce_sum_(r_dev)
            r_sum = eng.to_host(r_dev).reshape(R, R)
            A_loc = dist.reduce_scatter_rows(A_aug[:C])
            del A_aug
            C_loc = dist.reduce_scatter_rows(
                Cacc.view(R, world, Cl * D).permute(1, 0, 2).contiguous().view(world * R, Cl * D))
            del Cacc
            F_new = eng.tv_mstep(A_loc, C_loc, Cl, D, R)
        else:
            dist.allreduce_sum_(A_aug)
            dist.allreduce_sum_(Cacc)
            r_sum = eng.to_host(A_aug[C]).reshape(R, R)
            F_new = eng.tv_mstep(A_aug, Cacc, C, D, R)
        if min_div:
            # _R / nb_sessions, symmetric from its upper triangle; upper Cholesky factor
            # (scipy.linalg.cholesky default, factor_analyser.py:661-665).  One R x R matrix per iteration.
            up = np.triu(r_sum)
            r_bar = (up + np.triu(r_sum, 1).T) / float(total_sessions)
            ch = np.linalg.cholesky(r_bar).T
            F_new = eng.gemm(F_new, torch.from_numpy(np.ascontiguousarray(ch)).to(eng.device))
        if sharded:
            F_new = dist.allgather_rows(F_new, [F_new.shape[0]] * world)
        return F_new

    def _train(self, sources, ubm, tv_rank, nb_iter, min_div, tv_init, batch_size, save_init, output_file_name,
               always_write, resident=None):
        assert (isinstance(ubm, Mixture) and ubm.validate()), "Second argument must be a proper Mixture"
        assert (isinstance(nb_iter, int) and (0 < nb_iter)), "nb_iter must be a positive integer"
        eng = self.engine
        torch = eng.torch
        C, D = ubm.distrib_nb(), ubm.dim()
        sv_size = C * D
        self.mean = np.zeros(sv_size, dtype=STAT_TYPE)
        self.Sigma = np.zeros(sv_size, dtype=STAT_TYPE)
        if tv_init is None:
            # unseeded global RNG, as the reference.  With several ranks only rank 0 draws (rank threads of one process
            # share that RNG: a seeded run must not depend on which thread gets there first) and its matrix is the one
            # (the reference's ranks only agree from the bcast at the end of their first iteration on,
            # sidekit_mpi.py:120-123, 241)
            if dist.is_dist() and dist.rank() != 0:
                self.F = np.zeros((sv_size, tv_rank), dtype=STAT_TYPE)
            else:
                self.F = np.random.randn(sv_size, tv_rank).astype(STAT_TYPE)
            if dist.is_dist():
                t = torch.from_numpy(self.F).to(getattr(eng, "device", "cpu"))
                dist.allreduce_sum_(t)
                self.F = eng.to_host(t)
        else:
            self.F = tv_init
        if output_file_name is None and always_write:
            output_file_name = "temporary_factor_analyser"
        if save_init and dist.rank() == 0:
            self.write((output_file_name or "temporary_factor_analyser") + "_init" + self._ext())

        if resident is not None:
            ft, naug, total = resident
        else:
            ft, naug, total, _ = self._resident_stats(sources, ubm)
        batch = 4096   # device batch (14 GB of scratch at R = 400): results do not depend on it, the reference
                       # batches only to bound host memory; 4096 measured 2.5 % faster than 2048, 9 % faster than 1024
        T_dev = torch.from_numpy(np.ascontiguousarray(np.asarray(self.F, dtype=np.float64))).to(eng.device)
        # f~ does not change over the iterations: split it once for the two products that read it (about the time one
        # iteration saves, so from two iterations on; 12 bytes per element, skipped when that does not fit)
        planes = None
        if nb_iter >= 2 and ft is not None and hasattr(eng, "tv_ftilde_planes"):
            planes = eng.tv_ftilde_planes(ft, C, D)
        for it in range(nb_iter):
            T_dev = self._tv_iteration(ft, naug, T_dev, C, D, tv_rank, total, min_div, batch, ft_planes=planes)
            self.F = eng.to_host(T_dev)
            if output_file_name is not None and dist.rank() == 0:
                if it < nb_iter - 1:
                    self.write(output_file_name + "_it-{}".format(it) + self._ext())
                else:
                    self.write(output_file_name + self._ext())

    # ------------------------------------------------------------------ training drivers
    @fanout
    def total_variability(self, stat_server_filename, ubm, tv_rank, nb_iter=20, min_div=True, tv_init=None,
                          batch_size=300, save_init=False, output_file_name=None, num_thread=1):
        """TV matrix EM (factor_analyser.py:535-672).  `stat_server_filename` is a StatServer file name
        (.npz, or .h5 when h5py is installed), a list of them, or an in-memory StatServer (its statistics
        are rounded to float32 as a file would store them).  `num_thread` is ignored."""
        if not isinstance(stat_server_filename, list):
            stat_server_filename = [stat_server_filename]
        self._train(stat_server_filename, ubm, tv_rank, nb_iter, min_div, tv_init, batch_size, save_init,
                    output_file_name, always_write=True)

    @fanout
    def total_variability_single(self, stat_server_filename, ubm, tv_rank, nb_iter=20, min_div=True, tv_init=None,
                                 batch_size=300, save_init=False, output_file_name=None):
        """Same arithmetic as total_variability (factor_analyser.py:430-533); files are only written when
        output_file_name is given (:529-533)."""
        self._train([stat_server_filename], ubm, tv_rank, nb_iter, min_div, tv_init, batch_size, save_init,
                    output_file_name, always_write=False)

    def total_variability_raw(self, stat_server, ubm, tv_rank, nb_iter=20, min_div=True, tv_init=None,
                              save_init=False, output_file_name=None):
        """In-memory variant (factor_analyser.py:326-428): the StatServer is whitened in float64 (and,
        like the reference, left whitened), the same EM follows; files are always written
        ("temporary_factor_analyser" when no name is given, :381-382)."""
        assert (isinstance(stat_server, StatServer) and stat_server.validate()), \
            "First argument must be a proper StatServer"
        assert (isinstance(ubm, Mixture) and ubm.validate()), "Second argument must be a proper Mixture"
        assert (isinstance(tv_rank, int) and (0 < tv_rank <= min(stat_server.stat1.shape))), \
            "tv_rank must be a positive integer less than the dimension of the statistics"
        eng = self.engine
        torch = eng.torch
        n = stat_server.stat1.shape[0]
        a, b = dist.shard_bounds(n) if dist.is_dist() else (0, n)
        ft = naug = None
        if b > a:
            d0 = torch.from_numpy(np.ascontiguousarray(stat_server.stat0[a:b], dtype=np.float64)).to(eng.device)
            d1 = torch.from_numpy(np.ascontiguousarray(stat_server.stat1[a:b], dtype=np.float64)).to(eng.device)
            ft, naug = eng.tv_whiten(d0, d1, ubm.get_mean_super_vector(), ubm.get_invcov_super_vector())
        stat_server.whiten_stat1(ubm.get_mean_super_vector(), 1. / ubm.get_invcov_super_vector())
        self._train(None, ubm, tv_rank, nb_iter, min_div, tv_init, None, save_init, output_file_name,
                    always_write=True, resident=(ft, naug, n))

    # ------------------------------------------------------------------ PLDA (next row: the step after extraction)
    def plda(self, stat_server, rank_f, nb_iter=10, scaling_factor=1., output_file_name=None, save_partial=False):
        """Simplified PLDA (between-class subspace F, full residual covariance Sigma) by EM with minimum divergence --
        sidekit/factor_analyser.py:810-910 with its per-class loop fa_model_loop (:168-206).

        The reference walks the classes in Python: one rank x rank `inv` per distinct session count, then per class two
        small `dot`s and an outer product into a K x rank x rank array.  Here the classes are sorted by session count
        once; per iteration the K x R whitening and projection are two float64 MFMA GEMMs, e_h is one GEMM per distinct
        count on a contiguous row block, and the sums over classes of e_hh (weighted and unweighted) never form the
        K x rank x rank array: sum_i e_hh[i] = sum_u n_u inv_lambda_u + e_h'e_h.  The R x R eigen-decompositions,
        inverses and the rank x rank solve / Cholesky are single host LAPACK calls, as in the reference.  Every rank of
        a distributed job computes the same (small) model."""
        import scipy.linalg
        eng = self.engine
        torch = eng.torch
        self.mean = stat_server.get_mean_stat1()
        self.Sigma = stat_server.get_total_covariance_stat1()
        model_shifted_stat, session_per_model = stat_server.sum_stat_per_model()
        model_shifted_stat.stat0 *= scaling_factor
        model_shifted_stat.stat1 *= scaling_factor
        session_per_model *= scaling_factor
        sigma_obs = self.Sigma.copy()                       # get_total_covariance_stat1 again in the reference (:845)
        evals, evecs = scipy.linalg.eigh(sigma_obs)
        idx = np.argsort(evals)[::-1]
        self.F = evecs.real[:, idx[:rank_f]]

        n = model_shifted_stat.stat0[:, 0].astype(np.float64)
        order = np.argsort(n, kind="stable")
        n_sorted = n[order]
        uniq, first = np.unique(n_sorted, return_index=True)
        bounds = list(first) + [n.shape[0]]
        S1 = eng.rows_center(eng._f64_dev(model_shifted_stat.stat1), src=order)      # classes sorted by count
        n_col = eng._f64_dev(n_sorted.reshape(-1, 1))
        K, R = S1.shape
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(eng.device)

        for it in range(nb_iter):
            eigen_values, eigen_vectors = scipy.linalg.eigh(self.Sigma)
            ind = eigen_values.real.argsort()[::-1]
            sqr_inv_sigma = np.dot(eigen_vectors.real[:, ind], np.diag(1 / np.sqrt(eigen_values.real[ind])))
            # whitened class statistics (stat1 - stat0 mean) Q  (statserver.py:690-697, 747-764)
            W1 = eng.gemm(S1, dev(sqr_inv_sigma))
            eng.gemm(n_col, dev(self.mean.reshape(1, -1).dot(sqr_inv_sigma)), C=W1, alpha=-1.0, beta=1.0)
            self.F = sqr_inv_sigma.T.dot(self.F)
            A = self.F.T.dot(self.F)
            aux = eng.gemm(W1, dev(self.F))                                          # K x rank
            e_h = torch.empty((K, rank_f), dtype=torch.float64, device=eng.device)
            sum_il = np.zeros((rank_f, rank_f))
            sum_n_il = np.zeros((rank_f, rank_f))
            EhEh = torch.zeros((rank_f, rank_f), dtype=torch.float64, device=eng.device)
            nEhEh = torch.zeros((rank_f, rank_f), dtype=torch.float64, device=eng.device)
            for u, a, b in zip(uniq, bounds[:-1], bounds[1:]):
                inv_lambda = scipy.linalg.inv(u * A + np.eye(rank_f))
                eng.gemm(aux[a:b], dev(inv_lambda), C=e_h[a:b])
                sum_il += (b - a) * inv_lambda
                sum_n_il += (b - a) * u * inv_lambda
                eng.gemm(e_h[a:b], e_h[a:b], C=EhEh, transA=True, beta=1.0)
                eng.gemm(e_h[a:b], e_h[a:b], C=nEhEh, transA=True, alpha=float(u), beta=1.0)
            _R = (sum_il + eng.to_host(EhEh)) / session_per_model.shape[0]
            _C = eng.to_host(eng.gemm(e_h, W1, transA=True)).dot(scipy.linalg.inv(sqr_inv_sigma))
            _A = sum_n_il + eng.to_host(nEhEh)
            self.F = scipy.linalg.solve(_A, _C).T
            self.Sigma = sigma_obs - self.F.dot(_C) / session_per_model.sum()
            self.F = self.F.dot(scipy.linalg.cholesky(_R))
            if output_file_name is None:
                output_file_name = "temporary_plda"
            if dist.rank() == 0:
                if save_partial and it < nb_iter - 1:
                    self.write(output_file_name + "_it-{}".format(it) + self._ext())
                elif it == nb_iter - 1:
                    self.write(output_file_name + self._ext())

    # ------------------------------------------------------------------ extraction
    def _extract(self, ft, naug, ubm, batch, uncertainty, again=None):
        """again(): the plain float32 f~ of the same sessions, for the one case the planes cannot serve -- the
        all-float64 route that long sessions on trained loadings switch to (_estep_chain)."""
        eng = self.engine
        torch = eng.torch
        C, D = ubm.distrib_nb(), ubm.dim()
        T_dev = torch.from_numpy(np.ascontiguousarray(np.asarray(self.F, dtype=np.float64))).to(eng.device)
        T32, G = eng.tv_prepare(T_dev, C, D)
        # without `uncertainty` the device solves L x = aux from the Cholesky factor and never forms U^-1
        chain, T64 = self._estep_chain(T_dev, naug, C, D)
        more = {}
        if isinstance(ft, _FtPlanes):
            if chain == "precise":
                ft = again()
            else:
                ft, more = None, {"ft_planes": ft.planes}
        e_h, sig = eng.tv_estep(naug, ft, T32, G, C, D, batch=batch, uncertainty=bool(uncertainty), chain=chain, T64=T64,
                                **more)
        return e_h, sig

    @fanout
    def extract_ivectors(self, ubm, stat_server_filename, prefix='', batch_size=300, uncertainty=False,
                         num_thread=1):
        """i-vectors of every session of a StatServer file (factor_analyser.py:732-808).  Returns a
        StatServer with stat0 = ones (S x 1) and stat1 = i-vectors (S x R) [, diagonal uncertainty S x R]."""
        assert (isinstance(ubm, Mixture) and ubm.validate()), "Second argument must be a proper Mixture"
        eng = self.engine
        tv_rank = self.F.shape[1]
        if isinstance(stat_server_filename, StatServer):
            src = stat_server_filename
            modelset, segset, start, stop = src.modelset, src.segset, src.start, src.stop
        else:
            modelset, segset, start, stop, _, _ = _load_stat_arrays(stat_server_filename, prefix)
        nb_sessions = modelset.shape[0]
        # the whitening writes f~ as the planes the aux product reads (no float32 f~) unless the all-float64 route is
        # asked for outright; should the bound on cond(L_s) pick that route for this call, f~ is whitened again
        ft, naug, total, local = self._resident_stats([stat_server_filename], ubm, prefix,
                                                      rows_planes=self.chain_precision != "precise")
        a, b = local[0]
        torch = eng.torch
        width = tv_rank * (2 if uncertainty else 1)
        if ft is not None:
            e_h, sig = self._extract(ft, naug, ubm, EXTRACT_BATCH, uncertainty,
                                     again=lambda: self._resident_stats([stat_server_filename], ubm, prefix)[0])
            mine = torch.cat([e_h, sig], dim=1) if sig is not None else e_h
        else:
            mine = torch.zeros((0, width), dtype=torch.float64, device=getattr(eng, "device", "cpu"))
        if dist.is_dist():
            # each rank sends the i-vectors of ITS sessions (the Gatherv of sidekit_mpi.py:313-314), S x R doubles
            # in total; nothing is summed
            world = dist.world_size()
            counts = [hi - lo for lo, hi in (dist.shard_bounds(nb_sessions, world, r) for r in range(world))]
            mine = dist.allgather_rows(mine, counts)
        both = eng.to_host(mine)
        iv = np.ascontiguousarray(both[:, :tv_rank])
        iv_sigma = np.ascontiguousarray(both[:, tv_rank:]) if uncertainty else np.zeros((nb_sessions, tv_rank), dtype=STAT_TYPE)
        iv_server = StatServer()
        iv_server.modelset = np.asarray(modelset)       # stays as stored (bytes), factor_analyser.py:760-761
        iv_server.segset = np.asarray(segset)
        iv_server.start = start
        iv_server.stop = stop
        iv_server.stat0 = np.ones((nb_sessions, 1), dtype=STAT_TYPE)
        iv_server.stat1 = iv
        if uncertainty:
            return iv_server, iv_sigma
        return iv_server

    def extract_ivectors_single(self, ubm, stat_server, uncertainty=False):
        """In-memory variant (factor_analyser.py:674-730): whitening in float64; like the reference it
        leaves `stat_server` whitened."""
        assert (isinstance(stat_server, StatServer) and stat_server.validate()), \
            "First argument must be a proper StatServer"
        assert (isinstance(ubm, Mixture) and ubm.validate()), "Second argument must be a proper Mixture"
        eng = self.engine
        torch = eng.torch
        d0 = torch.from_numpy(np.ascontiguousarray(stat_server.stat0, dtype=np.float64)).to(eng.device)
        d1 = torch.from_numpy(np.ascontiguousarray(stat_server.stat1, dtype=np.float64)).to(eng.device)
        ft, naug = eng.tv_whiten(d0, d1, ubm.get_mean_super_vector(), ubm.get_invcov_super_vector())
        stat_server.whiten_stat1(ubm.get_mean_super_vector(), 1. / ubm.get_invcov_super_vector())
        e_h, sig = self._extract(ft, naug, ubm, EXTRACT_BATCH, uncertainty)
        iv_stat_server = StatServer()
        iv_stat_server.modelset = copy.deepcopy(stat_server.modelset)
        iv_stat_server.segset = copy.deepcopy(stat_server.segset)
        iv_stat_server.start = copy.deepcopy(stat_server.start)
        iv_stat_server.stop = copy.deepcopy(stat_server.stop)
        iv_stat_server.stat0 = np.ones((stat_server.modelset.shape[0], 1))
        iv_stat_server.stat1 = eng.to_host(e_h)
        if uncertainty:
            return iv_stat_server, eng.to_host(sig)
        return iv_stat_server
