"""ctypes binding of libskm.so (include/skm.h) + the device-side plumbing of the hot path.

PyTorch is used for device memory, streams and torch.distributed only; every number on the
path is produced by the hand-written gfx950 kernels in csrc/.  There is NO CPU fallback: if
the library or a GPU is missing, constructing a :class:`HipEngine` raises.
"""
import atexit
import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKM_LIB") or os.path.join(_HERE, "libskm.so")

_c_void_p = ctypes.c_void_p
_c_size_t = ctypes.c_size_t
_c_int = ctypes.c_int
_c_long = ctypes.c_long
_dp = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); exactly the declarations of include/skm.h
SIGNATURES = {
    "skm_version": (_c_int, []),
    "skm_last_error": (ctypes.c_char_p, []),
    "skm_ctx_create": (_c_int, [_c_int, ctypes.POINTER(_c_void_p)]),
    "skm_ctx_destroy": (_c_int, [_c_void_p]),
    "skm_ctx_cu_count": (_c_int, [_c_void_p]),
    "skm_prof_enable": (_c_int, [_c_void_p, _c_int]),
    "skm_prof_read": (_c_long, [_c_void_p, ctypes.c_char_p, _c_size_t]),
    "skm_gmm_packed_bytes": (_c_size_t, [_c_int, _c_int]),
    "skm_gmm_pack": (_c_int, [_c_void_p, _dp, _dp, _dp, _c_int, _c_int, _c_void_p, _c_void_p]),
    "skm_gmm_pack_ref": (_c_int, [_c_void_p, _dp, _dp, _dp, _dp, _c_int, _c_int, _c_void_p, _c_void_p]),
    "skm_gmm_llk_ref": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_long, _c_void_p, _c_void_p]),
    "skm_gmm_mstep_pack": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, ctypes.c_double, ctypes.c_double,
                                    _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "skm_gmm_workspace_bytes": (_c_size_t, [_c_void_p, _c_long, _c_int, _c_int]),
    "skm_gmm_estep": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p, _c_int,
                               _c_void_p, _c_size_t, _c_void_p]),
    "skm_gmm_estep_noalloc_max_bytes": (_c_size_t, []),
    "skm_bw_team_count": (_c_int, [_c_void_p, _c_int, _c_int]),
    "skm_gmm_estep_noalloc": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p, _c_int,
                                       _c_void_p]),
    "skm_gmm_loglik": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p,
                                _c_void_p, _c_size_t, _c_void_p]),
    "skm_gmm_logprob": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                 _c_void_p, _c_size_t, _c_void_p]),
    "skm_row_logsumexp": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "skm_bw_stats": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_int, _c_int, _c_int, _c_void_p,
                              _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "skm_frames_center": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "skm_gmm_stats_unshift": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p]),
    "skm_bw_unshift": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p]),
    "skm_segment_mean": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p]),
    "skm_row_normalize": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_void_p]),
    "skm_class_sums": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                                ctypes.c_double, _c_void_p, _c_void_p]),
    "skm_rows_center": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long,
                                 _c_int, _c_void_p, _c_void_p]),
    "skm_row_dot": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p]),
    "skm_pair_combine": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_long, ctypes.c_double, _c_void_p, ctypes.c_double,
                                  _c_void_p, ctypes.c_double, _c_void_p]),
    "skm_gemm": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_long,
                          _c_long, _c_long, _c_int, _c_int, ctypes.c_double, ctypes.c_double, _c_int, _c_long,
                          _c_long, _c_long, _c_void_p]),
    "skm_gemm_split_workspace_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "skm_gemm_split": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                _c_void_p, _c_size_t, _c_void_p]),
    "skm_tv_naug_stride": (_c_int, [_c_int]),
    "skm_tv_workspace_bytes": (_c_size_t, [_c_void_p, _c_int, _c_int, _c_int, _c_long, _c_int, _c_int]),
    "skm_tv_whiten": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_long, _c_int, _c_int, _c_void_p,
                               _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "skm_tv_prepare": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p]),
    "skm_tv_estep": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_int, _c_void_p,
                              _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p,
                              _c_void_p, _c_size_t, _c_void_p]),
    "skm_tv_ftilde_planes_bytes": (_c_size_t, [_c_long, _c_int, _c_int]),
    "skm_tv_ftilde_rows_bytes": (_c_size_t, [_c_long, _c_int, _c_int]),
    "skm_tv_whiten_planes": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_long, _c_int, _c_int, _c_void_p,
                                      _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "skm_tv_ftilde_planes": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_void_p, _c_void_p]),
    "skm_tv_estep_planes": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_long, _c_int, _c_int, _c_int, _c_void_p,
                                     _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p,
                                     _c_void_p, _c_size_t, _c_void_p]),
    "skm_tv_mstep": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p,
                              _c_void_p, _c_size_t, _c_void_p]),
    "skm_moments_workspace_bytes": (_c_size_t, [_c_void_p, _c_int]),
    "skm_moments": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_int, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    "skm_feat_workspace_bytes": (_c_size_t, [_c_long, _c_int, _c_int, _c_int]),
    "skm_feat_sliding_norm": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                       _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p]),
    "skm_feat_sliding_lds_chunk": (_c_int, [_c_int]),
    "skm_feat_sliding_norm_lds": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p,
                                           _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "skm_label_index_workspace_bytes": (_c_size_t, [_c_long]),
    "skm_label_index": (_c_int, [_c_void_p, _c_void_p, _c_long, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                 _c_size_t, _c_void_p]),
    "skm_feat_postproc": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_long, _c_void_p, _c_int, _c_void_p, _c_void_p,
                                   _c_int, _c_long, _c_int, _c_int, _dp, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p,
                                   _c_int, _c_void_p, _c_void_p, _c_void_p, _c_size_t, _c_void_p]),
    # one process, several GPUs (team.py)
    "skm_comm_init_all": (_c_int, [_c_int, ctypes.POINTER(_c_int), ctypes.POINTER(_c_void_p)]),
    "skm_comm_size": (_c_int, [_c_void_p]),
    "skm_comm_destroy": (_c_int, [_c_void_p]),
    "skm_allreduce_f64": (_c_int, [_c_void_p, ctypes.POINTER(_c_void_p), _c_size_t, _c_int, ctypes.POINTER(_c_void_p)]),
    "skm_reduce_scatter_f64": (_c_int, [_c_void_p, ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _c_size_t,
                                        ctypes.POINTER(_c_void_p)]),
    "skm_allgather_bytes": (_c_int, [_c_void_p, ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p), _c_size_t,
                                     ctypes.POINTER(_c_void_p)]),
}

_lib = None
_lib_lock = threading.Lock()
_live_engines = weakref.WeakSet()


def _close_engines():
    for e in list(_live_engines):
        e.close()


atexit.register(_close_engines)


class SkmError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libskm.so and attach the prototypes.  Raises if the library is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise SkmError("libskm.so not found at %s -- build it with speaker-recognition_amd/csrc/build.sh "
                           "(there is no CPU fallback)" % p)
        lib = ctypes.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


def _as_f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class HipEngine(object):
    """One engine per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device=None):
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise SkmError("no GPU visible: the sidekit hot path runs on MI355X only (no CPU fallback)")
        if device is None:
            device = torch.cuda.current_device()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        ctx = _c_void_p()
        self._check(self.lib.skm_ctx_create(self.device_index, ctypes.byref(ctx)))
        self.ctx = ctx
        self._ws = None
        self._ws_bytes = 0

        _live_engines.add(self)

    def close(self):
        """Release the context, its pending timing events and the workspace.  Registered with atexit so that it
        happens while the HIP runtime is still up (not from __del__ during interpreter shutdown)."""
        ctx = getattr(self, "ctx", None)
        if ctx:
            try:
                self.torch.cuda.synchronize(self.device)
                self.lib.skm_prof_enable(ctx, 0)
                buf = ctypes.create_string_buffer(16384)
                self.lib.skm_prof_read(ctx, buf, 16384)        # destroys the recorded events
                self.lib.skm_ctx_destroy(ctx)
            except Exception:
                pass
            self.ctx = None
        self._ws = None
        self._ws_bytes = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, rc):
        if rc != 0:
            raise SkmError("libskm error %d: %s" % (rc, self.lib.skm_last_error().decode("utf-8", "replace")))

    def _stream(self):
        return _c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, nbytes):
        if nbytes > self._ws_bytes:
            self._ws = None
            self._ws = self.torch.empty(int(nbytes) + 256, dtype=self.torch.uint8, device=self.device)
            self._ws_bytes = int(nbytes)
        base = self._ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        return aligned

    def cu_count(self):
        return self.lib.skm_ctx_cu_count(self.ctx)

    def upload_frames(self, x):
        """host (T x D) array -> float32 device tensor (the kernels' input layout)."""
        torch = self.torch
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=torch.float32)
        else:
            a = np.asarray(x)
            if a.ndim == 1:
                a = a[:, None]
            if a.nbytes >= (32 << 20) and a.dtype in (np.float32, np.float64):
                return self.upload_rows(a, dtype=torch.float32)      # pinned double buffers, rounding on the device
            a = np.ascontiguousarray(a, dtype=np.float32)
            t = torch.from_numpy(a).to(self.device)
        return t.contiguous()

    # ------------------------------------------------------------------ GMM
    def frame_shift(self, m):
        """A FrameShift for the per-dimension offset m (None passes through)."""
        if m is None or isinstance(m, FrameShift):
            return m
        return FrameShift(self, m)

    def center_frames(self, X, shift, inplace=False):
        """X - shift as float32 (skm_frames_center); `inplace` overwrites X (the caller owns it)."""
        torch = self.torch
        assert X.dim() == 2 and X.dtype == torch.float32 and X.is_contiguous() and X.shape[1] == shift.D
        out = X if inplace else torch.empty_like(X)
        self._check(self.lib.skm_frames_center(self.ctx, _c_void_p(X.data_ptr()), X.shape[0], X.shape[1],
                                               _c_void_p(shift.dev.data_ptr()), _c_void_p(out.data_ptr()),
                                               self._stream()))
        return out

    def stats_unshift_(self, flat, C, D, shift):
        """[N | F' | S' | llk] of centred frames -> the statistics of the frames themselves, in place."""
        assert flat.dtype == self.torch.float64 and flat.numel() == 2 * C * D + C + 1 and flat.is_contiguous()
        self._check(self.lib.skm_gmm_stats_unshift(self.ctx, _c_void_p(flat.data_ptr()), C, D,
                                                   _c_void_p(shift.dev.data_ptr()), self._stream()))
        return flat

    def bw_unshift_(self, stat0, stat1, D, shift):
        S, C = stat0.shape
        assert stat1.shape == (S, C * D) and stat0.is_contiguous() and stat1.is_contiguous()
        self._check(self.lib.skm_bw_unshift(self.ctx, _c_void_p(stat0.data_ptr()), _c_void_p(stat1.data_ptr()), S, C, D,
                                            _c_void_p(shift.dev.data_ptr()), self._stream()))
        return stat1

    def pack(self, w, mu, invcov, shift=None, ref_var=None):
        """Device image of the model.
        `shift` (a FrameShift or a D-vector): the image is the model of the CENTRED frames x - shift (means
        mu - shift); estep / bw_stats / loglik / logprob then centre the frames they are given (or take them as
        already centred, `centered=True`) and map the statistics back.
        `ref_var` (D-vector): the image is relative to the reference Gaussian N(0, diag(ref_var)) of those
        coordinates (skm_gmm_pack_ref): same posteriors and statistics from much smaller float32 partial sums;
        estep adds the reference's log-likelihood back.  For estep / bw_stats only (loglik / logprob return
        per-frame values)."""
        torch = self.torch
        w = _as_f64(w)
        mu = _as_f64(mu)
        invcov = _as_f64(invcov)
        C, D = mu.shape
        assert w.shape == (C,) and invcov.shape == (C, D)
        shift = self.frame_shift(shift)
        if shift is not None:
            assert shift.D == D
            mu = mu - shift.host[None, :]
        ref = None
        if ref_var is not None:
            ref = ref_var if isinstance(ref_var, FrameRef) else FrameRef(self, ref_var)
            assert ref.D == D
        nbytes = self.lib.skm_gmm_packed_bytes(C, D)
        packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._check(self.lib.skm_gmm_pack_ref(self.ctx, w.ctypes.data_as(_dp), mu.ctypes.data_as(_dp),
                                              invcov.ctypes.data_as(_dp),
                                              ref.host.ctypes.data_as(_dp) if ref is not None else None, C, D,
                                              _c_void_p(packed.data_ptr()), self._stream()))
        return PackedGMM(packed, C, D, shift, ref)

    def pack_stats(self, w, mu, invcov):
        """pack() for the statistics kernels (estep / bw_stats) as the drop-in classes do it: frames centred when the
        model sits far from the origin, image relative to the model's own reference Gaussian (stats_conditioning)."""
        off, ref = stats_conditioning(w, mu, invcov)
        return self.pack(w, mu, invcov, shift=off, ref_var=ref)

    def mstep_pack(self, stats, cov_var_ctl, gmm, ceil_cov, floor_cov, params=None, llk_mean=None):
        """M-step + re-pack on the device (skm_gmm_mstep_pack): `gmm` (a PackedGMM) is overwritten with the new
        model, `params` = [w | mu | invcov | sum N] float64 device tensor; `llk_mean` a 1-element float64 view."""
        torch = self.torch
        C, D = gmm.C, gmm.D
        assert stats.dtype == torch.float64 and stats.numel() == 2 * C * D + C + 1 and stats.is_contiguous()
        assert cov_var_ctl.dtype == torch.float64 and cov_var_ctl.numel() == C * D and cov_var_ctl.is_contiguous()
        if params is None:
            params = torch.empty(2 * C * D + C + 1, dtype=torch.float64, device=self.device)
        self._check(self.lib.skm_gmm_mstep_pack(
            self.ctx, _c_void_p(stats.data_ptr()), _c_void_p(cov_var_ctl.data_ptr()), C, D, float(ceil_cov),
            float(floor_cov), _c_void_p(params.data_ptr()), _c_void_p(gmm.packed.data_ptr()),
            _c_void_p(llk_mean.data_ptr()) if llk_mean is not None else None,
            _c_void_p(gmm.ref.dev.data_ptr()) if gmm.ref is not None else None, self._stream()))
        return params

    def estep(self, X, gmm, out=None, accumulate=False, centered=False, unshift=True):
        """[N | F | S | llk] float64 device tensor for the frames X (T x D float32 device tensor).
        A model packed with a shift: X is centred here unless `centered` says it already is, and the statistics
        are mapped back to those of the frames themselves unless `unshift` is False (the device EM loop runs its
        M-step in the centred coordinates)."""
        torch = self.torch
        T, D = X.shape
        assert D == gmm.D and X.dtype == torch.float32 and X.is_contiguous()
        C = gmm.C
        n = 2 * C * D + C + 1
        if gmm.shift is not None or gmm.ref is not None:
            if gmm.shift is not None and not centered:
                X = self.center_frames(X, gmm.shift)
            # the corrections are linear in the statistics of THIS call: they go on a fresh result, then it is added
            fresh = out if (out is not None and not accumulate) else None
            part = self.estep(X, _plain(gmm), out=fresh)
            if gmm.ref is not None:
                self._check(self.lib.skm_gmm_llk_ref(self.ctx, _c_void_p(part.data_ptr()), C, D, T,
                                                     _c_void_p(gmm.ref.dev.data_ptr()), self._stream()))
            if gmm.shift is not None and unshift:
                self.stats_unshift_(part, C, D, gmm.shift)
            if out is not None and accumulate:
                out += part
                return out
            return part
        if out is None:
            out = torch.empty(n, dtype=torch.float64, device=self.device)
            accumulate = False
        assert out.numel() == n and out.dtype == torch.float64
        need = self.lib.skm_gmm_workspace_bytes(self.ctx, T, C, D)
        ws = self.workspace(need)
        self._check(self.lib.skm_gmm_estep(self.ctx, _c_void_p(X.data_ptr()), T, C, D,
                                           _c_void_p(gmm.packed.data_ptr()), _c_void_p(out.data_ptr()),
                                           1 if accumulate else 0, _c_void_p(ws), need, self._stream()))
        return out

    def estep_noalloc(self, X, gmm, out, accumulate=False):
        """skm_gmm_estep_noalloc: the E-step on the library's built-in scratch; `out` must exist (nothing is
        allocated here either)."""
        torch = self.torch
        T, D = X.shape
        assert D == gmm.D and X.dtype == torch.float32 and X.is_contiguous()
        assert out.numel() == 2 * gmm.C * D + gmm.C + 1 and out.dtype == torch.float64
        self._check(self.lib.skm_gmm_estep_noalloc(self.ctx, _c_void_p(X.data_ptr()), T, gmm.C, D,
                                                   _c_void_p(gmm.packed.data_ptr()), _c_void_p(out.data_ptr()),
                                                   1 if accumulate else 0, self._stream()))
        return out

    def loglik(self, X, gmm, centered=False):
        torch = self.torch
        T, D = X.shape
        assert D == gmm.D
        assert gmm.ref is None, "per-frame log-likelihoods want an image packed without a reference Gaussian"
        if gmm.shift is not None and not centered:
            X = self.center_frames(X, gmm.shift)
        out = torch.empty(T, dtype=torch.float32, device=self.device)
        need = self.lib.skm_gmm_workspace_bytes(self.ctx, T, gmm.C, D)
        ws = self.workspace(need)
        self._check(self.lib.skm_gmm_loglik(self.ctx, _c_void_p(X.data_ptr()), T, gmm.C, D,
                                            _c_void_p(gmm.packed.data_ptr()), _c_void_p(out.data_ptr()),
                                            _c_void_p(ws), need, self._stream()))
        return out

    def logprob(self, X, gmm, with_loglik=False, centered=False):
        """T x C float64 device matrix ln(w_c N(x_t; c)) [and the T float32 row log-sum-exp]."""
        torch = self.torch
        T, D = X.shape
        assert D == gmm.D and X.dtype == torch.float32 and X.is_contiguous()
        assert gmm.ref is None, "log-probabilities want an image packed without a reference Gaussian"
        if gmm.shift is not None and not centered and T > 0:
            X = self.center_frames(X, gmm.shift)
        lp = torch.empty((T, gmm.C), dtype=torch.float64, device=self.device)
        llk = torch.empty(T, dtype=torch.float32, device=self.device) if with_loglik else None
        if T == 0:
            return (lp, llk) if with_loglik else lp
        need = self.lib.skm_gmm_workspace_bytes(self.ctx, T, gmm.C, D)
        ws = self.workspace(need)
        self._check(self.lib.skm_gmm_logprob(self.ctx, _c_void_p(X.data_ptr()), T, gmm.C, D,
                                             _c_void_p(gmm.packed.data_ptr()), _c_void_p(lp.data_ptr()),
                                             _c_void_p(llk.data_ptr()) if with_loglik else None,
                                             _c_void_p(ws), need, self._stream()))
        return (lp, llk) if with_loglik else lp

    def row_logsumexp(self, lp, want_pp=True):
        """(pp, log_lk) of a T x C float64 device matrix of log-probabilities (skm_row_logsumexp)."""
        torch = self.torch
        assert lp.dim() == 2 and lp.dtype == torch.float64 and lp.is_contiguous()
        T, C = lp.shape
        pp = torch.empty_like(lp) if want_pp else None
        llk = torch.empty(T, dtype=torch.float64, device=self.device)
        self._check(self.lib.skm_row_logsumexp(self.ctx, _c_void_p(lp.data_ptr()), T, C,
                                               _c_void_p(pp.data_ptr()) if want_pp else None,
                                               _c_void_p(llk.data_ptr()), self._stream()))
        return pp, llk

    def _bw_split_plan(self, seg_host, C, D=60):
        """Few long sessions leave teams of the Baum-Welch kernel idle (whole sessions are dealt to the teams of
        workgroups that hold one copy of the model: 200 x 20 000 frames ran at 86 TF against 108 TF for 2000 x 2000).
        Returns (piece offsets, owner of each piece) when cutting the sessions into 32-frame-aligned pieces pays,
        else None."""
        S = seg_host.size - 1
        if S == 0:
            return None
        teams = int(self.lib.skm_bw_team_count(self.ctx, int(C), int(D)))
        if teams <= 0:
            return None
        total = int(seg_host[-1] - seg_host[0])
        if S >= 4 * teams or total < 2048 * S:
            return None
        target = max(1024, -(-total // (8 * teams)))            # frames per piece
        target = (target + 31) // 32 * 32
        offs, owner = [int(seg_host[0])], []
        for s in range(S):
            a, b = int(seg_host[s]), int(seg_host[s + 1])
            n = max(1, -(-(b - a) // target))
            step = (-(-(b - a) // n) + 31) // 32 * 32            # 32-frame aligned: no partial tile inside a session
            for k in range(1, n):
                if a + k * step < b:
                    offs.append(a + k * step)
                    owner.append(s)
            offs.append(b)
            owner.append(s)
        if len(owner) <= S:
            return None
        return np.asarray(offs, dtype=np.int64), np.asarray(owner, dtype=np.int64)

    def bw_stats(self, X, seg_off, gmm, stat0=None, stat1=None, split_long=True, centered=False, own_frames=False):
        """Per-session N (S x C) and F (S x C*D, component-major), float64 device tensors.  A model packed with a
        shift: X is centred first (in place when the caller gives it up, `own_frames`) unless `centered`, and
        stat1 is mapped back (stat1 += stat0 (x) shift)."""
        torch = self.torch
        T, D = X.shape
        assert D == gmm.D and X.dtype == torch.float32 and X.is_contiguous()
        if gmm.shift is not None:
            if not centered and T > 0:
                X = self.center_frames(X, gmm.shift, inplace=own_frames)
            stat0, stat1 = self.bw_stats(X, seg_off, _plain(gmm), stat0, stat1, split_long)
            self.bw_unshift_(stat0, stat1, D, gmm.shift)
            return stat0, stat1
        if split_long and not isinstance(seg_off, torch.Tensor):
            plan = self._bw_split_plan(np.asarray(seg_off, dtype=np.int64), gmm.C, gmm.D)
            if plan is not None:
                # statistics of the pieces, then the rows of a session's pieces are added: a float64 MFMA product
                # with the 0/1 ownership matrix (exact weights, fixed summation order)
                offs, owner = plan
                p0, p1 = self.bw_stats(X, offs, gmm, split_long=False)
                S = len(seg_off) - 1
                own = np.zeros((S, owner.size), dtype=np.float64)
                own[owner, np.arange(owner.size)] = 1.0
                own_dev = torch.from_numpy(own).to(self.device)
                stat0 = self.gemm(own_dev, p0, C=stat0)
                stat1 = self.gemm(own_dev, p1, C=stat1)
                return stat0, stat1
        if not isinstance(seg_off, torch.Tensor):
            seg_off = torch.as_tensor(np.asarray(seg_off, dtype=np.int64))
        seg_off = seg_off.to(device=self.device, dtype=torch.int64).contiguous()
        S = seg_off.numel() - 1
        C = gmm.C
        if stat0 is None:
            stat0 = torch.empty((S, C), dtype=torch.float64, device=self.device)
        if stat1 is None:
            stat1 = torch.empty((S, C * D), dtype=torch.float64, device=self.device)
        assert stat0.shape == (S, C) and stat1.shape == (S, C * D)
        assert stat0.is_contiguous() and stat1.is_contiguous()
        need = self.lib.skm_gmm_workspace_bytes(self.ctx, T, C, D)
        ws = self.workspace(need)
        self._check(self.lib.skm_bw_stats(self.ctx, _c_void_p(X.data_ptr()), T, _c_void_p(seg_off.data_ptr()), S,
                                          C, D, _c_void_p(gmm.packed.data_ptr()), _c_void_p(stat0.data_ptr()),
                                          _c_void_p(stat1.data_ptr()), _c_void_p(ws), need, self._stream()))
        return stat0, stat1

    def segment_mean(self, v, seg_off):
        """float64 mean of the float32 device vector v over each [seg_off[s], seg_off[s+1])."""
        torch = self.torch
        if not isinstance(seg_off, torch.Tensor):
            seg_off = torch.as_tensor(np.asarray(seg_off, dtype=np.int64))
        seg_off = seg_off.to(device=self.device, dtype=torch.int64).contiguous()
        S = seg_off.numel() - 1
        assert v.dtype == torch.float32 and v.is_contiguous()
        out = torch.empty(S, dtype=torch.float64, device=self.device)
        self._check(self.lib.skm_segment_mean(self.ctx, _c_void_p(v.data_ptr()), _c_void_p(seg_off.data_ptr()), S,
                                              _c_void_p(out.data_ptr()), self._stream()))
        return out

    def moments(self, X):
        """[sum x | sum x^2] over the rows of X, float64 device tensor of 2*D."""
        torch = self.torch
        T, D = X.shape
        out = torch.empty(2 * D, dtype=torch.float64, device=self.device)
        need = self.lib.skm_moments_workspace_bytes(self.ctx, D)
        ws = self.workspace(need)
        self._check(self.lib.skm_moments(self.ctx, _c_void_p(X.data_ptr()), T, D, _c_void_p(out.data_ptr()),
                                         _c_void_p(ws), need, self._stream()))
        return out

    # ------------------------------------------------------------------ feature post-processing
    def feat_postproc(self, X, seg_off, label=None, cols=None, rasta=False, ndelta=0, delta_filter=None, norm=0,
                      global_mean=None, global_std=None, keep_all=True, out_dtype=None):
        """FeaturesServer.post_processing of a batch of sessions resident in HBM (skm_feat_postproc).
        X: (T x Draw) float32/float64 device tensor, session s = rows [seg_off[s], seg_off[s+1]).
        Returns (out [rows x Dout], label_out uint8 [rows], out_off int64 [S+1]) device tensors; with
        keep_all=False the rows are the labelled frames only (one host read of out_off[S] sizes the result)."""
        torch = self.torch
        assert X.dim() == 2 and X.dtype in (torch.float32, torch.float64) and X.stride(1) == 1
        if not isinstance(seg_off, torch.Tensor):
            seg_off = torch.as_tensor(np.asarray(seg_off, dtype=np.int64))
        seg_off = seg_off.to(device=self.device, dtype=torch.int64).contiguous()
        S = seg_off.numel() - 1
        D0 = X.shape[1] if cols is None else len(cols)
        cols_dev = None
        if cols is not None:
            cols = np.asarray(cols, dtype=np.int64)
            if cols.size == 0:
                raise Exception('filter list is empty')                        # features_server.py:255-256
            if cols.min() < -X.shape[1] or cols.max() >= X.shape[1]:
                raise IndexError("mask index out of range for %d columns" % X.shape[1])
            cols_dev = torch.as_tensor(np.where(cols < 0, cols + X.shape[1], cols).astype(np.int32)).to(self.device)
        dout = D0 * (ndelta + 1)
        if label is not None:
            if not isinstance(label, torch.Tensor):
                label = torch.as_tensor(np.ascontiguousarray(np.asarray(label, dtype=np.uint8)))
            label = label.to(device=self.device, dtype=torch.uint8).contiguous()
            assert label.numel() == X.shape[0]
        out_dtype = out_dtype or torch.float32
        assert out_dtype in (torch.float32, torch.float64)
        out = torch.empty((X.shape[0], dout), dtype=out_dtype, device=self.device)
        lab_out = torch.empty(X.shape[0], dtype=torch.uint8, device=self.device)
        out_off = torch.empty(S + 1, dtype=torch.int64, device=self.device)
        gm = gi = None
        if norm and global_mean is not None and (norm == 1 or global_std is not None):
            gm = torch.as_tensor(_as_f64(global_mean)).to(self.device)
            assert gm.numel() == dout
            if norm == 2:
                with np.errstate(divide="ignore"):
                    gi = torch.as_tensor(1.0 / _as_f64(global_std)).to(self.device)
                assert gi.numel() == dout
        filt = None
        if delta_filter is not None:
            filt = _as_f64(delta_filter)
            if filt.shape != (7,):
                # numpy.convolve output would not fit the reference's (T + 6)-row array (features.py:154,167)
                raise ValueError("could not broadcast: delta_filter must have 7 taps")
        need = self.lib.skm_feat_workspace_bytes(X.shape[0], S, D0, ndelta)
        ws = self.workspace(need)
        self._check(self.lib.skm_feat_postproc(
            self.ctx, _c_void_p(X.data_ptr()), 1 if X.dtype == torch.float64 else 0, X.stride(0),
            _c_void_p(cols_dev.data_ptr()) if cols_dev is not None else None, D0,
            _c_void_p(label.data_ptr()) if label is not None else None, _c_void_p(seg_off.data_ptr()), S,
            X.shape[0], 1 if rasta else 0, int(ndelta), filt.ctypes.data_as(_dp) if filt is not None else None, int(norm),
            _c_void_p(gm.data_ptr()) if gm is not None else None, _c_void_p(gi.data_ptr()) if gi is not None else None,
            1 if keep_all else 0, _c_void_p(out.data_ptr()), 1 if out_dtype == torch.float64 else 0,
            _c_void_p(lab_out.data_ptr()), _c_void_p(out_off.data_ptr()), _c_void_p(ws), need, self._stream()))
        if not keep_all:
            rows = int(out_off[-1].item())
            out, lab_out = out[:rows], lab_out[:rows]
        return out, lab_out, out_off

    # ------------------------------------------------------------------ total variability
    GEMM_F32_F32, GEMM_F32_F64ACC, GEMM_F64_F64, GEMM_F64_F32OUT, GEMM_F32_F64OUT, GEMM_F32S_F32, GEMM_F32S_F64OUT = 0, 1, 2, 3, 4, 5, 6

    def gemm(self, A, B, C=None, transA=False, transB=False, alpha=1.0, beta=0.0, kind=None):
        """C = alpha op(A) op(B) + beta C on 2-D contiguous device tensors (float32 or float64)."""
        torch = self.torch
        assert A.dim() == 2 and B.dim() == 2 and A.is_contiguous() and B.is_contiguous()
        M, K = (A.shape[1], A.shape[0]) if transA else A.shape
        K2, N = (B.shape[1], B.shape[0]) if transB else B.shape
        assert K == K2, (A.shape, B.shape, transA, transB)
        if kind is None:
            kind = self.GEMM_F64_F64 if A.dtype == torch.float64 else self.GEMM_F32_F32
        out_dtype = torch.float32 if kind in (self.GEMM_F32_F32, self.GEMM_F64_F32OUT, self.GEMM_F32S_F32) else torch.float64
        if C is None:
            C = torch.empty((M, N), dtype=out_dtype, device=self.device)
            beta = 0.0
        assert C.shape == (M, N) and C.dtype == out_dtype and C.is_contiguous()
        self._check(self.lib.skm_gemm(self.ctx, kind, _c_void_p(A.data_ptr()), _c_void_p(B.data_ptr()),
                                      _c_void_p(C.data_ptr()), M, N, K, A.shape[1], B.shape[1], N,
                                      1 if transA else 0, 1 if transB else 0, float(alpha), float(beta), 1, 0, 0, 0,
                                      self._stream()))
        return C

    def gemm_split(self, A, B, C=None, out_f64=False):
        """C (+)= A B for float32 device matrices through the bf16 x 3 split kernel (skm_gemm_split)."""
        torch = self.torch
        assert A.dtype == B.dtype == torch.float32 and A.is_contiguous() and B.is_contiguous()
        M, K = A.shape
        K2, N = B.shape
        assert K == K2
        acc = C is not None
        if C is None:
            C = torch.empty((M, N), dtype=torch.float64 if out_f64 else torch.float32, device=self.device)
        assert C.shape == (M, N) and C.is_contiguous()
        need = self.lib.skm_gemm_split_workspace_bytes(M, N, K)
        ws = self.workspace(need)
        self._check(self.lib.skm_gemm_split(self.ctx, self._ptr(A), self._ptr(B), self._ptr(C), M, N, K,
                                            1 if C.dtype == torch.float64 else 0, 1 if acc else 0, _c_void_p(ws), need,
                                            self._stream()))
        return C

    def feat_sliding_norm(self, Y, Yn, label, seg_host, win=301, reduce=False, compact=False, out_dtype=None):
        """cep_sliding_norm over a batch (skm_feat_sliding_norm).  Y: un-normalised frames of all sessions (float64, or float32
        for a float32 result: half the bytes of the window walks),
        Yn: whole-session cms / cmvn of the same frames, or a callable that returns it (called only when a session
        has at most `win` labelled frames), or None; label: uint8 / bool per row (device tensor or host array), seg_host: int64[S+1].
        The index bookkeeping (rows of the labelled frames, labelled frames before each session) stays on the
        device; S + 1 counts come to the host to size the work list.
        compact=True writes only the labelled frames (row coff[s] + p), in `out_dtype` (float32 default).
        Returns (out, idx device int64 rows of the labelled frames, coff int64[S+1] host)."""
        torch = self.torch
        assert Y.dim() == 2 and Y.dtype in (torch.float32, torch.float64) and Y.is_contiguous()
        seg = np.asarray(seg_host, dtype=np.int64)
        if not isinstance(label, torch.Tensor):
            label = torch.from_numpy(np.ascontiguousarray(np.asarray(label, dtype=np.uint8)))
        lab = label.to(self.device)
        if lab.dtype != torch.uint8:
            lab = lab.ne(0).to(torch.uint8)
        lab = lab.contiguous()
        assert lab.numel() == Y.shape[0] and seg[0] == 0 and seg[-1] <= lab.numel()
        # rows of the labelled frames and labelled frames before each session (skm_label_index); S + 1 counts come to
        # the host to size the work lists
        T_lab = int(seg[-1])
        seg_dev = self._i64_dev(seg)
        idx_dev = torch.empty(max(T_lab, 1), dtype=torch.int64, device=self.device)
        coff_dev = torch.empty(seg.size, dtype=torch.int64, device=self.device)
        if T_lab > 0:
            need = self.lib.skm_label_index_workspace_bytes(T_lab)
            iws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
            self._check(self.lib.skm_label_index(self.ctx, self._ptr(lab), T_lab, self._ptr(seg_dev), seg.size - 1,
                                                 self._ptr(idx_dev), self._ptr(coff_dev), self._ptr(iws), need, self._stream()))
        else:
            coff_dev.zero_()
        coff = coff_dev.cpu().numpy()
        idx_dev = idx_dev[:int(coff[-1])]
        n = np.diff(coff)
        short = n <= win
        if callable(Yn):
            Yn = Yn() if short.any() else None
        assert Yn is not None or not short.any()
        assert Yn is None or (Yn.dtype == Y.dtype and Yn.shape == Y.shape and Yn.is_contiguous())
        if compact:
            out = torch.empty((int(coff[-1]), Y.shape[1]), dtype=out_dtype or torch.float32, device=self.device)
            assert Y.dtype == torch.float64 or out.dtype == torch.float32
        else:
            out = Y.clone()
        in_f64, out_f64 = 1 if Y.dtype == torch.float64 else 0, 1 if out.dtype == torch.float64 else 0

        def work_list(span, chunk_len):
            per = (span + chunk_len - 1) // chunk_len
            chunk_sess = np.repeat(np.arange(n.size, dtype=np.int32), per)
            first = np.concatenate([[0], np.cumsum(per)])[:-1]
            chunk_p0 = ((np.arange(chunk_sess.size, dtype=np.int64) - np.repeat(first, per)) * chunk_len).astype(np.int32)
            return torch.from_numpy(chunk_sess).to(self.device), torch.from_numpy(chunk_p0).to(self.device)
        # sessions with more than `win` labelled frames: window in LDS, every row read once per chunk (+ win rows of halo)
        if (~short).any() and win <= 601 and os.environ.get("SKM_SLIDING_LDS", "1") != "0":
            cs, cp = work_list(np.where(short, 0, n), self.lib.skm_feat_sliding_lds_chunk(in_f64))
            self._check(self.lib.skm_feat_sliding_norm_lds(self.ctx, self._ptr(Y), in_f64, self._ptr(out), out_f64,
                                                           1 if compact else 0, Y.shape[1], self._ptr(idx_dev),
                                                           self._ptr(coff_dev), self._ptr(cs), self._ptr(cp),
                                                           int(cs.numel()), int(win), 1 if reduce else 0, self._stream()))
            todo = short
        else:
            todo = np.ones_like(short)
        # short sessions (whole-session result copied from Yn; frames when not compact), or every session when the
        # window does not fit the LDS: lane-per-column walk
        span = np.where(todo, n if compact else np.where(short, np.diff(seg), n), 0)
        if span.any():
            chunk_len = 512
            cs, cp = work_list(span, chunk_len)
            self._check(self.lib.skm_feat_sliding_norm(self.ctx, self._ptr(Y), self._ptr(Yn), in_f64, self._ptr(out), out_f64,
                                                       1 if compact else 0, Y.shape[1], self._ptr(idx_dev),
                                                       self._ptr(coff_dev), self._ptr(seg_dev), self._ptr(cs), self._ptr(cp),
                                                       int(cs.numel()), chunk_len, int(win), 1 if reduce else 0,
                                                       self._stream()))
        return out, idx_dev, coff

    # ------------------------------------------------------------------ i-vector back-end reductions (float64)
    def _f64_dev(self, x):
        torch = self.torch
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=torch.float64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(self.device)

    def _i64_dev(self, x):
        torch = self.torch
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=torch.int64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.int64)).to(self.device)

    @staticmethod
    def _ptr(t):
        return _c_void_p(t.data_ptr()) if t is not None else None

    def class_sums(self, X, cls_off, src=None, mean=False, alpha=1.0):
        """K x R sums (or means) of the rows X[src[cls_off[k]:cls_off[k+1]]] of a float64 device matrix."""
        assert X.dim() == 2 and X.dtype == self.torch.float64 and X.is_contiguous()
        cls_off, src = self._i64_dev(cls_off), self._i64_dev(src)
        K = cls_off.numel() - 1
        out = self.torch.empty((K, X.shape[1]), dtype=self.torch.float64, device=self.device)
        self._check(self.lib.skm_class_sums(self.ctx, self._ptr(X), X.shape[1], self._ptr(src), self._ptr(cls_off), K,
                                            X.shape[1], 1 if mean else 0, float(alpha), self._ptr(out), self._stream()))
        return out

    def col_mean(self, X):
        """1 x R column mean of a float64 device matrix, two levels of fixed-order sums (chunks of 128 rows)."""
        S = X.shape[0]
        off = np.append(np.arange(0, S, 128, dtype=np.int64), S)
        part = self.class_sums(X, off)
        return self.class_sums(part, np.array([0, part.shape[0]], dtype=np.int64), alpha=1.0 / max(S, 1))

    def rows_center(self, X, Mu=None, cls=None, src=None, scale=None):
        """(X[src[i]] - Mu[cls[i]]) * scale[i] as a new S x R float64 device matrix."""
        torch = self.torch
        assert X.dim() == 2 and X.dtype == torch.float64 and X.is_contiguous()
        src, cls = self._i64_dev(src), self._i64_dev(cls)
        S = X.shape[0] if src is None else src.numel()
        R = X.shape[1]
        if Mu is not None:
            Mu = self._f64_dev(Mu).reshape(-1, R)
        if scale is not None:
            scale = self._f64_dev(scale)
            assert scale.numel() == S
        assert cls is None or (Mu is not None and cls.numel() == S)
        out = torch.empty((S, R), dtype=torch.float64, device=self.device)
        self._check(self.lib.skm_rows_center(self.ctx, self._ptr(X), R, self._ptr(src), self._ptr(cls), self._ptr(Mu),
                                             self._ptr(scale), S, R, self._ptr(out), self._stream()))
        return out

    def row_dot(self, A, B):
        assert A.shape == B.shape and A.dtype == B.dtype == self.torch.float64 and A.is_contiguous() and B.is_contiguous()
        out = self.torch.empty((A.shape[0],), dtype=self.torch.float64, device=self.device)
        self._check(self.lib.skm_row_dot(self.ctx, self._ptr(A), self._ptr(B), A.shape[0], A.shape[1], self._ptr(out),
                                         self._stream()))
        return out

    def pair_combine_(self, Sc, a, b, alpha=1.0, ca=1.0, cb=1.0):
        """In place: Sc[i, j] = alpha Sc[i, j] + ca a[i] + cb b[j]."""
        assert Sc.dim() == 2 and Sc.dtype == self.torch.float64 and Sc.is_contiguous()
        assert a.numel() == Sc.shape[0] and b.numel() == Sc.shape[1]
        self._check(self.lib.skm_pair_combine(self.ctx, self._ptr(Sc), Sc.shape[0], Sc.shape[1], float(alpha),
                                              self._ptr(a), float(ca), self._ptr(b), float(cb), self._stream()))
        return Sc

    def gram(self, X, alpha=1.0):
        """alpha X'X (R x R float64) of an S x R float64 device matrix.  R x R is a handful of output tiles (49 at
        R = 400) while K = S is long, so the rows are cut into slices that run as one batched GEMM (about 1024
        workgroups) and the slices' products are added in a fixed order (skm_class_sums over the slabs)."""
        torch = self.torch
        assert X.dim() == 2 and X.dtype == torch.float64 and X.is_contiguous()
        S, R = X.shape
        tiles = ((R + 63) // 64) ** 2
        Z = min(-(-1024 // tiles), S // 1024)
        if Z < 2:
            return self.gemm(X, X, transA=True, alpha=alpha)
        kc = S // Z
        slabs = torch.empty((Z, R * R), dtype=torch.float64, device=self.device)
        self._check(self.lib.skm_gemm(self.ctx, self.GEMM_F64_F64, self._ptr(X), self._ptr(X), self._ptr(slabs),
                                      R, R, kc, R, R, R, 1, 0, 1.0, 0.0, Z, kc * R, kc * R, R * R, self._stream()))
        if Z * kc < S:
            rest = X[Z * kc:]
            self.gemm(rest, rest, C=slabs[0].view(R, R), transA=True, beta=1.0)
        return self.class_sums(slabs, np.array([0, Z], dtype=np.int64), alpha=alpha).view(R, R)

    def row_normalize_(self, X):
        """In-place length normalisation of the rows of a float64 device matrix (norm clipped at 1e-8)."""
        assert X.dim() == 2 and X.dtype == self.torch.float64 and X.is_contiguous()
        self._check(self.lib.skm_row_normalize(self.ctx, _c_void_p(X.data_ptr()), X.shape[0], X.shape[1],
                                               self._stream()))
        return X

    def tv_whiten(self, stat0, stat1, mu_sv, invcov_sv):
        """(ftilde float32 [S x CD], naug float32 [S x (C+1)]) from device statistics (float32 or float64)."""
        torch = self.torch
        S, C = stat0.shape
        CD = stat1.shape[1]
        D = CD // C
        assert stat0.dtype == stat1.dtype and stat0.dtype in (torch.float32, torch.float64)
        assert stat0.is_contiguous() and stat1.is_contiguous()
        mu = torch.as_tensor(_as_f64(mu_sv)).to(self.device)
        sq = torch.as_tensor(np.sqrt(_as_f64(invcov_sv))).to(self.device)
        ft = torch.empty((S, CD), dtype=torch.float32, device=self.device)
        naug = torch.empty((S, self.lib.skm_tv_naug_stride(C)), dtype=torch.float32, device=self.device)
        self._check(self.lib.skm_tv_whiten(self.ctx, _c_void_p(stat0.data_ptr()), _c_void_p(stat1.data_ptr()),
                                           1 if stat0.dtype == torch.float64 else 0, S, C, D,
                                           _c_void_p(mu.data_ptr()), _c_void_p(sq.data_ptr()),
                                           _c_void_p(ft.data_ptr()), _c_void_p(naug.data_ptr()), self._stream()))
        return ft, naug

    def tv_whiten_planes(self, stat0, stat1, mu_sv, invcov_sv):
        """tv_whiten for extract_ivectors: (rows image of the bf16 x 3 planes of ftilde, naug) -- the float32 ftilde
        is never stored; tv_estep(naug, None, ..., ft_planes=planes) takes the aux product from the planes.  None
        instead of the planes when the split products are switched off (SKM_TV_SPLIT=0 / SKM_TV_FT_PLANES=0)."""
        import os
        torch = self.torch
        if os.environ.get("SKM_TV_FT_PLANES", "1") == "0" or os.environ.get("SKM_TV_SPLIT", "1") == "0":
            return None, None
        S, C = stat0.shape
        D = stat1.shape[1] // C
        assert stat0.dtype == stat1.dtype and stat0.dtype in (torch.float32, torch.float64)
        assert stat0.is_contiguous() and stat1.is_contiguous()
        mu = torch.as_tensor(_as_f64(mu_sv)).to(self.device)
        sq = torch.as_tensor(np.sqrt(_as_f64(invcov_sv))).to(self.device)
        planes = torch.empty(int(self.lib.skm_tv_ftilde_rows_bytes(S, C, D)), dtype=torch.uint8, device=self.device)
        naug = torch.empty((S, self.lib.skm_tv_naug_stride(C)), dtype=torch.float32, device=self.device)
        self._check(self.lib.skm_tv_whiten_planes(self.ctx, _c_void_p(stat0.data_ptr()), _c_void_p(stat1.data_ptr()),
                                                  1 if stat0.dtype == torch.float64 else 0, S, C, D,
                                                  _c_void_p(mu.data_ptr()), _c_void_p(sq.data_ptr()),
                                                  _c_void_p(planes.data_ptr()), _c_void_p(naug.data_ptr()), self._stream()))
        return planes, naug

    def tv_prepare(self, T, C, D):
        """(T32, G) for the loading matrix T (CD x R float64 device tensor)."""
        torch = self.torch
        CD, R = T.shape
        assert CD == C * D and T.dtype == torch.float64 and T.is_contiguous()
        T32 = torch.empty((CD, R), dtype=torch.float32, device=self.device)
        G = torch.empty((C, R * R), dtype=torch.float32, device=self.device)
        self._check(self.lib.skm_tv_prepare(self.ctx, _c_void_p(T.data_ptr()), C, D, R, _c_void_p(T32.data_ptr()),
                                            _c_void_p(G.data_ptr()), self._stream()))
        return T32, G

    TV_CHAIN = {None: 0, "default": 0, "f32": 1, "f64": 2, "precise": 3}

    def tv_ftilde_planes(self, ft, C, D):
        """The bf16 x 3 planes of f~ (skm_tv_ftilde_planes): built once per training run and handed to every
        tv_estep(..., ft_planes=...), which then runs aux = f~ T and C += e_h' f~ on the bf16 matrix cores too.
        12 bytes per element of f~; None when SKM_TV_FT_PLANES=0 or when they would not fit beside what is resident."""
        import os
        if os.environ.get("SKM_TV_FT_PLANES", "1") == "0" or os.environ.get("SKM_TV_SPLIT", "1") == "0":
            return None
        S = ft.shape[0]
        need = int(self.lib.skm_tv_ftilde_planes_bytes(S, C, D))
        free, _total = self.torch.cuda.mem_get_info(self.device)
        if need == 0 or need > 0.6 * free:
            return None
        planes = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
        self._check(self.lib.skm_tv_ftilde_planes(self.ctx, _c_void_p(ft.data_ptr()), S, C, D,
                                                  _c_void_p(planes.data_ptr()), self._stream()))
        return planes

    def tv_estep(self, naug, ft, T32, G, C, D, batch=1024, A_aug=None, Cacc=None, uncertainty=False, chain=None,
                 T64=None, ft_planes=None):
        """E-step over the sessions of (naug, ft).  Returns (e_h, iv_sigma or None); when A_aug/Cacc are
        given (training) the accumulators are updated in place.  chain: "f32" / "f64" = type of the per-session
        Cholesky / inverse chain (None: float32 unless SKM_TV_CHAIN says otherwise); "precise" (with T64, the float64
        loading matrix on the device) = everything e_h depends on in float64, for long sessions / large loadings.
        ft_planes: what tv_ftilde_planes(ft, C, D) returned, or None."""
        torch = self.torch
        S = naug.shape[0]
        R = T32.shape[1]
        batch = int(max(1, min(batch, S)))
        e_h = torch.empty((S, R), dtype=torch.float64, device=self.device)
        sig = torch.empty((S, R), dtype=torch.float64, device=self.device) if uncertainty else None
        assert (chain == "precise") == (T64 is not None), "chain='precise' goes with T64"
        if T64 is not None:
            assert T64.dtype == torch.float64 and T64.shape == (C * D, R) and T64.is_contiguous()
        need = self.lib.skm_tv_workspace_bytes(self.ctx, C, D, R, S, batch, self.TV_CHAIN[chain])
        ws = self.workspace(need)
        null = _c_void_p(None)
        if ft_planes is not None:
            full = self.lib.skm_tv_ftilde_planes_bytes(S, C, D)
            rows = self.lib.skm_tv_ftilde_rows_bytes(S, C, D)
            assert ft_planes.numel() == full or (ft_planes.numel() == rows and A_aug is None), "ft_planes of another shape"
        assert ft is not None or (ft_planes is not None and A_aug is None and T64 is None), \
            "ft may only be None for extraction from its planes"
        self._check(self.lib.skm_tv_estep_planes(self.ctx, _c_void_p(naug.data_ptr()),
                                          _c_void_p(ft.data_ptr()) if ft is not None else null,
                                          _c_void_p(ft_planes.data_ptr()) if ft_planes is not None else null, S, C, D, R,
                                          _c_void_p(T32.data_ptr()), _c_void_p(G.data_ptr()), batch,
                                          _c_void_p(A_aug.data_ptr()) if A_aug is not None else null,
                                          _c_void_p(Cacc.data_ptr()) if Cacc is not None else null,
                                          _c_void_p(e_h.data_ptr()),
                                          _c_void_p(sig.data_ptr()) if sig is not None else null,
                                          self.TV_CHAIN[chain],
                                          _c_void_p(T64.data_ptr()) if T64 is not None else null,
                                          _c_void_p(ws), need, self._stream()))
        return e_h, sig

    def tv_mstep(self, A_aug, Cacc, C, D, R, batch=1024):
        """New loading matrix (CD x R float64) from the accumulators; A_aug[0:C] is destroyed."""
        torch = self.torch
        batch = int(max(1, min(batch, C)))
        F = torch.empty((C * D, R), dtype=torch.float64, device=self.device)
        need = self.lib.skm_tv_workspace_bytes(self.ctx, C, D, R, 0, batch, 0)
        ws = self.workspace(need)
        self._check(self.lib.skm_tv_mstep(self.ctx, _c_void_p(A_aug.data_ptr()), _c_void_p(Cacc.data_ptr()), C, D, R,
                                          batch, _c_void_p(F.data_ptr()), _c_void_p(ws), need, self._stream()))
        return F

    # ------------------------------------------------------------------ per-kernel timing (HIP events)
    def profile(self, on):
        self._check(self.lib.skm_prof_enable(self.ctx, 1 if on else 0))

    def profile_read(self):
        """{kernel name: (launches, total_ms)} since the last read; waits for the events."""
        buf = ctypes.create_string_buffer(8192)
        n = self.lib.skm_prof_read(self.ctx, buf, 8192)
        if n < 0:
            self._check(int(n))
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    # ------------------------------------------------------------------ host conveniences
    def to_host(self, t):
        return t.detach().cpu().numpy()

    def upload_rows(self, a, dtype=None, chunk_bytes=64 << 20):
        """Device copy of a 2-D host array, converted to `dtype` ON THE DEVICE (same IEEE rounding as numpy's astype),
        through two page-locked staging buffers: the host copy of one chunk into its buffer overlaps the transfer of the
        previous one.  (`torch.from_numpy(a.astype(float32)).to(device)` converts on one host core and then moves
        pageable memory.)"""
        torch = self.torch
        a = np.asarray(a)
        assert a.ndim == 2
        src_dtype = torch.from_numpy(np.empty(0, dtype=a.dtype)).dtype
        dtype = dtype or src_dtype
        n, w = a.shape
        out = torch.empty((n, w), dtype=dtype, device=self.device)
        if n == 0 or w == 0:
            return out
        per = max(1, int(chunk_bytes // (w * a.dtype.itemsize)))
        key = ("up", src_dtype, per * w)
        if getattr(self, "_pin_up_key", None) != key:
            self._pin_up = [torch.empty(per * w, dtype=src_dtype, pin_memory=True) for _ in range(2)]
            self._pin_up_ev = [torch.cuda.Event() for _ in range(2)]
            self._pin_up_used = [False, False]
            self._pin_up_key = key
        st = torch.cuda.current_stream(self.device)
        for k, r0 in enumerate(range(0, n, per)):
            r1 = min(n, r0 + per)
            j = k & 1
            if self._pin_up_used[j]:
                self._pin_up_ev[j].synchronize()          # the transfer that last read this buffer is done
            buf = self._pin_up[j][:(r1 - r0) * w].view(r1 - r0, w)
            self._host_copy(buf.numpy(), a[r0:r1])
            out[r0:r1].copy_(buf.to(self.device, non_blocking=True) if dtype != src_dtype else buf, non_blocking=True)
            self._pin_up_ev[j].record(st)
            self._pin_up_used[j] = True
        return out

    def copy_stream(self):
        """A second stream of this device for transfers that must overlap the kernels (StatServer.accumulate_stat)."""
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = self.torch.cuda.Stream(device=self.device)
        return self._copy_stream

    def upload_sessions(self, sessions, chunk_bytes=64 << 20):
        """The frames of a list of host sessions (each n_i x D, any float dtype) as ONE float32 device tensor
        (sum n_i x D), in order.  Every session is copied (and converted) once, straight into one of two page-locked
        staging buffers; full buffers go to the device asynchronously while the next one is filled.  Replaces
        `upload_frames(np.concatenate(sessions))`: a pageable concatenation followed by a pageable transfer (0.8 s for
        the 1.9 GB of 4000 sessions x 2000 frames on this box)."""
        torch = self.torch
        sessions = [np.asarray(x) for x in sessions]
        sessions = [x[:, None] if x.ndim == 1 else x for x in sessions]
        D = sessions[0].shape[1] if sessions else 0
        T = int(sum(x.shape[0] for x in sessions))
        out = torch.empty((T, D), dtype=torch.float32, device=self.device)
        if T == 0 or D == 0:
            return out
        per = max(1, int(chunk_bytes // (D * 4)))              # rows per staging buffer
        key = ("up", torch.float32, per * D)
        if getattr(self, "_pin_up_key", None) != key:
            self._pin_up = [torch.empty(per * D, dtype=torch.float32, pin_memory=True) for _ in range(2)]
            self._pin_up_ev = [torch.cuda.Event() for _ in range(2)]
            self._pin_up_used = [False, False]
            self._pin_up_key = key
        st = torch.cuda.current_stream(self.device)
        state = {"j": 0, "fill": 0, "done": 0}

        def buffer():
            j = state["j"]
            if state["fill"] == 0 and self._pin_up_used[j]:
                self._pin_up_ev[j].synchronize()              # the transfer that last read this buffer is done
            return self._pin_up[j].numpy().reshape(per, D)

        def send():
            j, n = state["j"], state["fill"]
            if n:
                out[state["done"]:state["done"] + n].copy_(self._pin_up[j][:n * D].view(n, D), non_blocking=True)
                self._pin_up_ev[j].record(st)
                self._pin_up_used[j] = True
                state["done"] += n
                state["fill"] = 0
                state["j"] = j ^ 1

        for x in sessions:
            assert x.shape[1] == D, "sessions of different dimensions"
            r = 0
            while r < x.shape[0]:
                buf = buffer()
                n = min(x.shape[0] - r, per - state["fill"])
                buf[state["fill"]:state["fill"] + n] = x[r:r + n]      # numpy converts to float32 while copying
                state["fill"] += n
                r += n
                if state["fill"] == per:
                    send()
        send()
        return out

    def _host_copy(self, dst, src, threads=4):
        """dst[...] = src for two 2-D host arrays, the rows split over a few threads (numpy releases the GIL while it
        copies; one core moves ~11 GB/s on this box, the pinned staging buffers arrive at 25+)."""
        n = src.shape[0]
        if n < 2 * threads or src.nbytes < (8 << 20):
            dst[...] = src
            return
        if getattr(self, "_copy_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._copy_pool = ThreadPoolExecutor(max_workers=threads)
        step = -(-n // threads)

        def part(i):
            dst[i:i + step] = src[i:i + step]
        list(self._copy_pool.map(part, range(0, n, step)))

    def to_host_rows(self, t, out, rows=None, chunk_bytes=64 << 20):
        """out[rows] = t (a 2-D device tensor; rows None: out[:] = t) through two page-locked staging buffers: the copy
        of one chunk off the device overlaps the host copy of the previous one into `out`.  A plain `.cpu()` into
        pageable memory moves ~4 GB/s on this box; this path is bound by the host memcpy (the drop-in API hands
        statistics back as host float64 arrays: 49 GB at BASELINE configs[3])."""
        torch = self.torch
        t = t.detach()
        assert t.dim() == 2 and out.ndim == 2 and t.shape[1] == out.shape[1]
        n = t.shape[0]
        if rows is not None:
            rows = np.asarray(rows)
            assert rows.shape[0] == n
            if n and np.array_equal(rows, np.arange(rows[0], rows[0] + n)):
                out, rows = out[int(rows[0]):int(rows[0]) + n], None
        if n == 0:
            return
        per = max(1, int(chunk_bytes // max(1, t.shape[1] * t.element_size())))
        # one pair of page-locked buffers per dtype, sized by capacity: stat0 (width C) and stat1 (width C * D) of one
        # batch take turns through the same pair instead of re-allocating it on every call
        need = per * t.shape[1]
        pins = getattr(self, "_pins", None)
        if pins is None:
            pins = self._pins = {}
        ent = pins.get(t.dtype)
        if ent is None or ent[0][0].numel() < need:
            ent = pins[t.dtype] = ([torch.empty(need, dtype=t.dtype, pin_memory=True) for _ in range(2)],
                                   [torch.cuda.Event() for _ in range(2)])
        self._pin, self._pin_ev = ent
        st = torch.cuda.current_stream(self.device)
        spans = [(a, min(n, a + per)) for a in range(0, n, per)]

        def issue(k):
            a, b = spans[k]
            buf = self._pin[k & 1][:(b - a) * t.shape[1]].view(b - a, t.shape[1])
            buf.copy_(t[a:b], non_blocking=True)
            self._pin_ev[k & 1].record(st)
            return buf
        pending = issue(0)
        for k, (a, b) in enumerate(spans):
            nxt = issue(k + 1) if k + 1 < len(spans) else None      # goes to the other buffer while this one is drained
            self._pin_ev[k & 1].synchronize()
            src = pending.numpy()
            if rows is None:
                self._host_copy(out[a:b], src)
            else:
                out[rows[a:b]] = src
            pending = nxt

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)


class PackedGMM(object):
    """Device image of a diagonal GMM in MFMA fragment order (see csrc/skm_gmm.hip).  `shift` (a FrameShift or
    None): the image is the model of the frames minus that offset.  `ref` (a FrameRef or None): the image is
    relative to that reference Gaussian (skm_gmm_pack_ref)."""

    def __init__(self, packed, C, D, shift=None, ref=None):
        self.packed = packed
        self.C = C
        self.D = D
        self.shift = shift
        self.ref = ref


def _plain(gmm):
    """The same device image with no correction attached (what the kernels see)."""
    return PackedGMM(gmm.packed, gmm.C, gmm.D, None, None)


class FrameRef(object):
    """Reference Gaussian N(0, diag(var)) of an image packed by skm_gmm_pack_ref: `host` / `dev` hold 1 / var
    (0 where the variance is not a positive finite number: that dimension has no reference)."""

    def __init__(self, engine, var):
        var = np.asarray(var, dtype=np.float64).reshape(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.where(np.isfinite(var) & (var > 0), 1.0 / var, 0.0)
        self.host = np.ascontiguousarray(np.where(np.isfinite(inv), inv, 0.0))
        self.D = self.host.size
        self.dev = engine.torch.from_numpy(self.host.copy()).to(engine.device)


class FrameShift(object):
    """Per-dimension offset subtracted from un-normalised frames (include/skm.h, skm_frames_center).  `host` is the
    float32 rounding of the requested offset, as float64: the one number the frames, the model and the
    back-mapping of the statistics all use."""

    def __init__(self, engine, m):
        m32 = np.asarray(m, dtype=np.float64).reshape(-1).astype(np.float32)
        self.host = m32.astype(np.float64)
        self.D = self.host.size
        self.dev = engine.torch.from_numpy(self.host.copy()).to(engine.device)


def stats_conditioning(w, mu, invcov):
    """(offset or None, ref_var or None) that condition the float32 logits of a model for the statistics kernels
    (both leave posteriors and statistics unchanged, include/skm.h):
    offset   to centre the frames on when the model sits far from the origin (un-normalised features): the model's
             own mean sum_c w_c mu_c, judged against its total standard deviation per dimension (centering_offset);
    ref_var  per-dimension second moment of the whole model about that offset (float32-rounded, as the frames will
             be shifted) -- the variance of the reference Gaussian of skm_gmm_pack_ref.  SKM_REF=0 switches it off."""
    w = np.asarray(w, dtype=np.float64)
    mu = np.asarray(mu, dtype=np.float64)
    invcov = np.asarray(invcov, dtype=np.float64)
    if mu.ndim != 2 or mu.size == 0:
        return None, None
    w = w / np.sum(w)
    with np.errstate(divide="ignore", invalid="ignore"):
        m = w @ mu
        second = w @ (1.0 / invcov + np.square(mu))
    off = centering_offset(m, np.sqrt(np.maximum(second - np.square(m), 0.0)))
    s = 0.0 if off is None else np.asarray(off, dtype=np.float32).astype(np.float64)
    ref = None if os.environ.get("SKM_REF", "1") == "0" else second - 2.0 * s * m + np.square(s)
    return off, ref


def centering_offset(mean, std):
    """Offset to centre frames on, or None when the data already sit at the origin: some dimension's mean is more
    than half a standard deviation away from zero.  cmvn / cms features never qualify (the unchanged float32 path);
    raw cepstra do.  SKM_CENTER=0 / 1 forces the answer."""
    mode = os.environ.get("SKM_CENTER", "auto")
    mean = np.asarray(mean, dtype=np.float64).reshape(-1)
    if mode == "0" or not np.all(np.isfinite(mean)):
        return None
    if mode == "1":
        return mean
    std = np.asarray(std, dtype=np.float64).reshape(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        far = np.abs(mean) > 0.5 * std
    return mean if bool(np.any(far & np.isfinite(std))) else None


_default_engine = None


def default_engine():
    """The process-wide engine on the current CUDA/HIP device (LOCAL_RANK aware); inside a rank thread of a
    single-process team (team.py) the engine of that rank's GPU."""
    global _default_engine
    from . import team as _team
    mine = _team.engine_of_thread()
    if mine is not None:
        return mine
    if _default_engine is None:
        _default_engine = HipEngine()
    return _default_engine


def set_default_engine(engine):
    global _default_engine
    _default_engine = engine
