"""TEST DOUBLE (never shipped, never imported by the package): an engine with the interface of
speaker-recognition_amd.engine.HipEngine whose arithmetic is the CPU oracle.  It exists so that the
host-side logic of the drop-in classes -- sharding over ranks, all-reduce of packed statistics,
M-step, file products -- can be exercised on CPU with the gloo backend (world_size 2), where no
GPU and therefore no HIP kernel is available."""
import numpy as np
import torch

from oracle import sidekit_ref as orc


class _Packed(object):
    def __init__(self, w, mu, invcov, shift=None):
        self.gmm = orc.GMM(w, mu, invcov)
        self.C, self.D = self.gmm.C, self.gmm.D
        self.shift = shift
        self.ref = None         # the float64 oracle needs no reference Gaussian (a float32 conditioning device)


class _Shift(object):
    """HipEngine's FrameShift: the float32 rounding of the offset, as float64."""
    def __init__(self, m):
        self.host = np.asarray(m, dtype=np.float64).reshape(-1).astype(np.float32).astype(np.float64)
        self.D = self.host.size


class OracleEngine(object):
    def __init__(self):
        self.torch = torch
        self.device = torch.device("cpu")

    def upload_frames(self, x):
        if isinstance(x, torch.Tensor):
            return x.to(torch.float32).contiguous()
        a = np.asarray(x, dtype=np.float32)
        if a.ndim == 1:
            a = a[:, None]
        return torch.from_numpy(np.ascontiguousarray(a))

    def frame_shift(self, m):
        return m if m is None or isinstance(m, _Shift) else _Shift(m)

    def center_frames(self, X, shift, inplace=False):
        out = (X.numpy() - shift.host.astype(np.float32)[None, :]).astype(np.float32)
        if inplace:
            X.copy_(torch.from_numpy(out))
            return X
        return torch.from_numpy(out)

    def stats_unshift_(self, flat, C, D, shift):
        a = flat.numpy()
        N = a[:C].copy()
        F = a[C:C + C * D].reshape(C, D).copy()
        m = shift.host[None, :]
        a[C:C + C * D] = (F + N[:, None] * m).ravel()
        a[C + C * D:C + 2 * C * D] += (m * (2.0 * F + N[:, None] * m)).ravel()
        return flat

    def pack(self, w, mu, invcov, shift=None, ref_var=None):
        shift = self.frame_shift(shift)
        mu = np.asarray(mu, dtype=np.float64)
        return _Packed(w, mu if shift is None else mu - shift.host[None, :], invcov, shift)

    def mstep_pack(self, stats, cov_var_ctl, gmm, ceil_cov, floor_cov, params=None, llk_mean=None):
        """Oracle M-step (sidekit/mixture.py:590-611) on the packed statistics; `gmm` is updated in place."""
        C, D = gmm.C, gmm.D
        flat = stats.numpy()
        N = flat[:C].copy()
        F = flat[C:C + C * D].reshape(C, D)
        S = flat[C + C * D:C + 2 * C * D].reshape(C, D)
        g = gmm.gmm
        g.cov_var_ctl = cov_var_ctl.numpy().reshape(C, D)
        orc.maximization(g, N, F, S, ceil_cov=ceil_cov, floor_cov=floor_cov)
        if llk_mean is not None:
            llk_mean[0] = flat[-1] / N.sum()
        return torch.from_numpy(np.concatenate([g.w, g.mu.ravel(), g.invcov.ravel(), [N.sum()]]))

    def estep(self, X, gmm, out=None, accumulate=False, centered=False, unshift=True):
        if gmm.shift is not None and not centered:
            X = self.center_frames(X, gmm.shift)
        N, F, S, llk = orc.expectation(gmm.gmm, X.numpy().astype(np.float64))
        flat = torch.from_numpy(np.concatenate([N, F.ravel(), S.ravel(), [llk]]))
        if gmm.shift is not None and unshift:
            self.stats_unshift_(flat, gmm.C, gmm.D, gmm.shift)
        if out is not None and accumulate:
            out += flat
            return out
        return flat

    def feat_postproc(self, X, seg_off, label=None, cols=None, rasta=False, ndelta=0, delta_filter=None, norm=0,
                      global_mean=None, global_std=None, keep_all=True, out_dtype=None):
        """Oracle post-processing (sidekit/features_server.py:202-246) session by session."""
        x = X.numpy().astype(np.float64)
        off = np.asarray(seg_off)
        lab = np.ones(x.shape[0], dtype=bool) if label is None else np.asarray(label).astype(bool)
        feats, labs, rows = [], [], [0]
        for i in range(len(off) - 1):
            f, l = orc.post_processing(x[off[i]:off[i + 1]], lab[off[i]:off[i + 1]], mask=cols, rasta=rasta,
                                       delta=ndelta >= 1, double_delta=ndelta == 2,
                                       delta_filter=orc.DELTA_FILTER if delta_filter is None else delta_filter,
                                       feat_norm=[None, "cms", "cmvn"][norm], keep_all_features=keep_all,
                                       global_mean=global_mean, global_std=global_std)
            feats.append(f)
            labs.append(l)
            rows.append(rows[-1] + f.shape[0])
        out = torch.from_numpy(np.vstack(feats)).to(out_dtype or torch.float32)
        return out, torch.from_numpy(np.concatenate(labs).astype(np.uint8)), torch.from_numpy(np.array(rows))

    def feat_sliding_norm(self, Y, Yn, label, seg_host, win=301, reduce=False, compact=False, out_dtype=None):
        """Oracle cep_sliding_norm (sidekit/frontend/normfeat.py:194-229) session by session; same contract as
        HipEngine.feat_sliding_norm (the whole-session fallback is taken from Yn as the device kernel does)."""
        y = Y.numpy().copy()
        lab = (label.numpy() if isinstance(label, torch.Tensor) else np.asarray(label)).astype(bool)
        seg = np.asarray(seg_host, dtype=np.int64)
        n = np.array([lab[seg[i]:seg[i + 1]].sum() for i in range(seg.size - 1)], dtype=np.int64)
        if callable(Yn):
            Yn = Yn() if (n <= win).any() else None
        for i in range(seg.size - 1):
            a, b = seg[i], seg[i + 1]
            if n[i] <= win:
                y[a:b] = Yn.numpy()[a:b]
            else:
                blk = y[a:b]
                orc.cep_sliding_norm(blk, lab[a:b], win=win, reduce=reduce)
        idx = np.flatnonzero(lab[:seg[-1]]).astype(np.int64)
        if compact:
            y = np.ascontiguousarray(y[idx])
        out = torch.from_numpy(y)
        if compact:
            out = out.to(out_dtype or torch.float32)
        return (out, torch.from_numpy(idx),
                np.concatenate([[0], np.cumsum(n)]).astype(np.int64))

    def moments(self, X):
        x = X.numpy().astype(np.float64)
        return torch.from_numpy(np.concatenate([x.sum(0), (x ** 2).sum(0)]))

    def bw_stats(self, X, seg_off, gmm, stat0=None, stat1=None, centered=False, own_frames=False):
        if gmm.shift is not None and not centered:
            X = self.center_frames(X, gmm.shift)
        x = X.numpy().astype(np.float64)
        off = np.asarray(seg_off)
        s0, s1 = orc.accumulate_stat(gmm.gmm, [x[off[i]:off[i + 1]] for i in range(len(off) - 1)])
        if gmm.shift is not None:
            s1 += np.repeat(s0, gmm.D, axis=1) * np.tile(gmm.shift.host, gmm.C)[None, :]
        return torch.from_numpy(s0), torch.from_numpy(s1)

    def loglik(self, X, gmm, centered=False):
        if gmm.shift is not None and not centered:
            X = self.center_frames(X, gmm.shift)
        lp = orc.log_posterior(gmm.gmm, X.numpy().astype(np.float64))
        return torch.from_numpy(orc.logsumexp_rows(lp)[1].astype(np.float32))

    def segment_mean(self, v, seg_off):
        off = np.asarray(seg_off)
        x = v.numpy().astype(np.float64)
        return torch.from_numpy(np.array([x[off[i]:off[i + 1]].mean() for i in range(len(off) - 1)]))

    def to_host(self, t):
        return t.detach().cpu().numpy()

    # ---- TV (dense float64 restatement of csrc/skm_tv.hip's data flow)
    def tv_whiten(self, stat0, stat1, mu_sv, invcov_sv):
        s0 = stat0.numpy().copy()
        s1 = stat1.numpy().copy()
        C = s0.shape[1]
        D = s1.shape[1] // C
        idx = np.repeat(np.arange(C), D)
        s1 -= s0[:, idx] * np.asarray(mu_sv)
        s1 *= np.sqrt(np.asarray(invcov_sv))
        naug = np.concatenate([s0, np.ones((s0.shape[0], 1), dtype=s0.dtype)], axis=1).astype(np.float32)
        return torch.from_numpy(s1.astype(np.float32)), torch.from_numpy(naug)

    def tv_prepare(self, T, C, D):
        return T.clone(), None

    def tv_estep(self, naug, ft, T32, G, C, D, batch=256, A_aug=None, Cacc=None, uncertainty=False, chain=None, T64=None):
        Fm = T32.numpy()
        R = Fm.shape[1]
        n = naug.numpy().astype(np.float64)[:, :C + 1]
        f = ft.numpy().astype(np.float64)
        idx = np.repeat(np.arange(C), D)
        e_h = np.zeros((n.shape[0], R))
        sig = np.zeros((n.shape[0], R))
        for s in range(n.shape[0]):
            inv = np.linalg.inv(np.eye(R) + (Fm.T * n[s, idx]).dot(Fm))
            e_h[s] = Fm.T.dot(f[s]).dot(inv)
            e_hh = inv + np.outer(e_h[s], e_h[s])
            sig[s] = np.diag(e_hh)
            if A_aug is not None:
                A_aug += torch.from_numpy(np.outer(n[s], e_hh.ravel()))
        if Cacc is not None:
            Cacc += torch.from_numpy(e_h.T.dot(f))
        return torch.from_numpy(e_h), (torch.from_numpy(sig) if uncertainty else None)

    def tv_mstep(self, A_aug, Cacc, C, D, R, batch=128):
        A = A_aug.numpy().reshape(-1, R, R)
        Cc = Cacc.numpy()
        Fm = np.zeros((C * D, R))
        for c in range(C):
            Fm[c * D:(c + 1) * D] = np.linalg.solve(A[c], Cc[:, c * D:(c + 1) * D]).T
        return torch.from_numpy(Fm)

    def gemm(self, A, B, C=None, transA=False, transB=False, alpha=1.0, beta=0.0, **kw):
        a = A.numpy().T if transA else A.numpy()
        r = alpha * a.dot(B.numpy().T if transB else B.numpy())
        if C is not None and beta != 0.0:
            r = r + beta * C.numpy()
        if C is not None:                       # the HIP engine writes C in place
            C.copy_(torch.from_numpy(np.ascontiguousarray(r)))
            return C
        return torch.from_numpy(np.ascontiguousarray(r))

    # i-vector back-end reductions (numpy statements of the skm_class_sums / skm_rows_center / skm_row_dot /
    # skm_pair_combine contracts in include/skm.h)
    def _f64_dev(self, x):
        return x.double().contiguous() if isinstance(x, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))

    def class_sums(self, X, cls_off, src=None, mean=False, alpha=1.0):
        x = X.numpy()
        off = np.asarray(cls_off, dtype=np.int64)
        src = np.arange(x.shape[0]) if src is None else np.asarray(src, dtype=np.int64)
        out = np.zeros((off.size - 1, x.shape[1]))
        for k in range(off.size - 1):
            rows = x[src[off[k]:off[k + 1]]]
            out[k] = alpha * (rows.mean(axis=0) if mean else rows.sum(axis=0))
        return torch.from_numpy(out)

    def col_mean(self, X):
        return torch.from_numpy(X.numpy().mean(axis=0, keepdims=True))

    def rows_center(self, X, Mu=None, cls=None, src=None, scale=None):
        x = X.numpy()
        x = x if src is None else x[np.asarray(src, dtype=np.int64)]
        if Mu is not None:
            mu = self._f64_dev(Mu).numpy().reshape(-1, x.shape[1])
            x = x - (mu[0] if cls is None else mu[np.asarray(cls, dtype=np.int64)])
        if scale is not None:
            x = x * self._f64_dev(scale).numpy()[:, None]
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))

    def gram(self, X, alpha=1.0):
        return torch.from_numpy(alpha * X.numpy().T.dot(X.numpy()))

    def row_dot(self, A, B):
        return torch.from_numpy((A.numpy() * B.numpy()).sum(axis=1))

    def pair_combine_(self, Sc, a, b, alpha=1.0, ca=1.0, cb=1.0):
        s = Sc.numpy()
        s[...] = alpha * s + ca * a.numpy()[:, None] + cb * b.numpy()[None, :]
        return Sc

    def row_normalize_(self, X):
        x = X.numpy()
        x /= np.clip(np.linalg.norm(x, axis=1), 1e-8, np.inf)[:, None]
        return X
