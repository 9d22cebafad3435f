"""Input boundary of the hot path: a features server hands frames to Mixture.EM_split and
StatServer.accumulate_stat.  The reference's FeaturesServer (sidekit/features_server.py) reads
HDF5 files and post-processes them on the host; file reading and feature extraction are out of
scope.  This in-memory server keeps the calls the hot path makes:

    load(show, channel=0, input_feature_filename=None, label=None, start=None, stop=None)
        -> (T x D array, bool[T] VAD labels)                  (features_server.py:397-451)
    stack_features_parallel(show_list, ..., num_thread=1) -> N x D array   (features_server.py:603-666)
    post_processing(feat, label, global_mean=None, global_std=None)         (features_server.py:202-246)

plus the attributes accumulate_stat touches (keep_all_features, features_extractor,
feature_filename_structure; statserver.py:631-645).  Values may be NumPy arrays or torch
tensors already resident on the GPU.

Post-processing (the step in front of the path) runs on the GPU through
skm_feat_postproc: column mask, RASTA, delta / double delta, cms / cmvn, VAD selection -- the
configuration of the reference's experiments (model_interface.py:45-52) included.  `load_batch`
post-processes many sessions in one launch and leaves the frames in HBM for the statistics kernels.
Not covered: 'stg' (broken in the reference: float slice indices, frontend/normfeat.py:166),
dct_pca / sdc / context / traps.
"""
import re

import numpy as np

_NORMS = {None: 0, "cms": 1, "cmvn": 2}


def parse_mask(mask):
    """'[1-3,10,15-20]' -> [1, 2, 3, 10, 15, ..., 20]: comma-separated column indices or inclusive ranges inside
    one pair of brackets (same grammar and errors as sidekit/sv_utils.py:377-394)."""
    body = "".join(mask.split())[1:-1]
    indices = []
    for item in body.split(","):
        m = re.fullmatch(r"(\d+)(?:-(\d+))?", item)
        if m is None:
            raise Exception("Wrong mask format")
        first = int(m.group(1))
        last = first if m.group(2) is None else int(m.group(2))
        indices.extend(range(first, last + 1))
    return indices


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class FeaturesServer(object):
    def __init__(self, features=None, vad=None, features_extractor=None, feature_filename_structure=None,
                 dataset_list=None, mask=None, feat_norm=None, global_cmvn=None, delta=None, double_delta=None,
                 delta_filter=None, rasta=None, keep_all_features=True, is_sharded=False, engine=None,
                 dct_pca=False, sdc=False, context=None, traps_dct_nb=None, **_ignored):
        self.features = dict(features) if features is not None else {}
        self.vad = dict(vad) if vad is not None else {}
        self.features_extractor = features_extractor
        self.feature_filename_structure = feature_filename_structure
        self.dataset_list = dataset_list
        self.keep_all_features = keep_all_features
        # True when every rank of a multi-GPU job was given only ITS OWN frames already
        self.is_sharded = is_sharded
        # post-processing options, defaults of features_server.py:119-133
        self.mask = parse_mask(mask) if isinstance(mask, str) else (list(mask) if mask is not None else None)
        self.feat_norm = feat_norm
        self.global_cmvn = global_cmvn
        self.delta = bool(delta) if delta is not None else False
        self.double_delta = bool(double_delta) if double_delta is not None else False
        self.delta_filter = np.array([.25, .5, .25, 0, -.25, -.5, -.25]) if delta_filter is None \
            else np.asarray(delta_filter, dtype=np.float64)
        self.rasta = bool(rasta) if rasta is not None else False
        if dct_pca or sdc or (context is not None and tuple(context) != (0, 0)) or traps_dct_nb:
            raise NotImplementedError("dct_pca / sdc / context / traps are outside the device post-processing path")
        self._engine = engine

    @property
    def engine(self):
        if self._engine is None:
            from .engine import default_engine
            self._engine = default_engine()
        return self._engine

    def add(self, show, cep, vad=None):
        self.features[show] = cep
        if vad is not None:
            self.vad[show] = np.asarray(vad, dtype=bool)

    # ------------------------------------------------------------------ post-processing on the device
    def has_post_processing(self):
        return bool(self.mask is not None or self.rasta or self.delta or self.double_delta
                    or self.feat_norm is not None)

    def _options(self):
        if self.feat_norm not in _NORMS:
            if self.feat_norm == 'stg':
                # the reference raises here as well (float slice indices, frontend/normfeat.py:166)
                raise TypeError("slice indices must be integers or None or have an __index__ method")
            norm = 0                                          # features_server.py:288-289: warning, no normalisation
        else:
            norm = _NORMS[self.feat_norm]
        # features_server.py:298-305: double delta only on top of delta
        ndelta = (2 if self.double_delta else 1) if self.delta else 0
        return norm, ndelta

    def _post_process_device(self, X, seg_off, label, keep_all, global_mean=None, global_std=None, out_dtype=None):
        """Batch of sessions (device tensor X, seg_off host int64[S+1], label uint8/bool [rows] or None)
        -> (out, label_out, out_off) device tensors."""
        norm, ndelta = self._options()
        seg = np.asarray(seg_off, dtype=np.int64)
        if self.rasta and seg.size > 1 and (np.diff(seg) < 3).any():
            # features_server.py:322: cep[2, :] of a session with fewer than three frames
            raise IndexError("index 2 is out of bounds: RASTA needs at least 3 frames per session")
        eng = self.engine
        if self.feat_norm in ('cmvn_sliding', 'cms_sliding'):
            return self._sliding(X, seg, label, keep_all, ndelta, out_dtype)
        return eng.feat_postproc(X, seg, label=label, cols=self.mask, rasta=self.rasta, ndelta=ndelta,
                                 delta_filter=self.delta_filter, norm=norm, global_mean=global_mean,
                                 global_std=global_std, keep_all=keep_all, out_dtype=out_dtype)

    def _sliding(self, X, seg, label, keep_all, ndelta, out_dtype, win=301):
        """'cmvn_sliding' / 'cms_sliding' (features_server.py:282-287; global statistics are not used by them):
        mask / RASTA / deltas without normalisation for every frame, then cep_sliding_norm on the device; sessions
        with at most `win` labelled frames take the whole-session cmvn / cms pass instead (normfeat.py:206-210)."""
        eng = self.engine
        torch = eng.torch
        reduce = self.feat_norm == 'cmvn_sliding'
        # the un-normalised frames are float64 unless the result is float32 and compacted (the statistics kernels'
        # input): then float32 halves the bytes of the window walks, whose sums stay float64
        mid = torch.float32 if (not keep_all and (out_dtype or torch.float32) == torch.float32) else torch.float64
        common = dict(label=label, cols=self.mask, rasta=self.rasta, ndelta=ndelta, delta_filter=self.delta_filter,
                      keep_all=True, out_dtype=mid)
        Y, lab_out, _ = eng.feat_postproc(X, seg, norm=0, **common)
        # the whole-session pass is only run when some session has at most `win` labelled frames; without
        # keep_all the kernel writes the labelled frames only, compacted, in the output type
        out, idx_dev, coff = eng.feat_sliding_norm(
            Y, lambda: eng.feat_postproc(X, seg, norm=2 if reduce else 1, **common)[0], lab_out, seg, win=win,
            reduce=reduce, compact=not keep_all, out_dtype=out_dtype or torch.float32)
        if not keep_all:
            lab_out = torch.ones(out.shape[0], dtype=torch.uint8, device=eng.device)
            out_off = torch.from_numpy(coff).to(eng.device)
        else:
            out_off = torch.from_numpy(seg - seg[0]).to(eng.device)
        return out.to(out_dtype or torch.float32), lab_out, out_off

    def post_processing(self, feat, label, global_mean=None, global_std=None):
        """One session (features_server.py:202-246).  NumPy in -> NumPy float64 out; a device tensor in ->
        device tensors out (float32, the statistics kernels' input type)."""
        eng = self.engine
        torch = eng.torch
        on_device = _is_torch(feat)
        if on_device:
            X = feat.to(eng.device)
            if X.dtype not in (torch.float32, torch.float64):
                X = X.to(torch.float32)
        else:
            a = np.asarray(feat)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            X = torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)
        lab = None
        if label is not None:
            lab = label if _is_torch(label) else torch.from_numpy(np.ascontiguousarray(np.asarray(label, dtype=np.uint8)))
        out, lab_out, _ = self._post_process_device(X, [0, X.shape[0]], lab, self.keep_all_features, global_mean,
                                                    global_std, out_dtype=None if on_device else torch.float64)
        if on_device:
            return out, lab_out.to(torch.bool)
        return eng.to_host(out), lab_out.cpu().numpy().astype(bool)

    def load_batch(self, show_list, keep_all_features=None):
        """Frames of many sessions, post-processed in ONE device pass and left in HBM:
        (X float32 [rows x D] device tensor, seg_off int64 [S+1] NumPy, labels uint8 device tensor)."""
        eng = self.engine
        torch = eng.torch
        keep_all = self.keep_all_features if keep_all_features is None else keep_all_features
        blocks = [self.features[s] for s in show_list]
        lens = np.array([b.shape[0] for b in blocks], dtype=np.int64)
        seg = np.concatenate([[0], np.cumsum(lens)])
        dev_blocks = [b.to(eng.device) if _is_torch(b) else torch.from_numpy(np.ascontiguousarray(np.asarray(b))).to(eng.device)
                      for b in blocks]
        dt = torch.float64 if any(b.dtype == torch.float64 for b in dev_blocks) else torch.float32
        X = torch.cat([b.to(dt) for b in dev_blocks], dim=0) if len(dev_blocks) > 1 else dev_blocks[0].to(dt)
        lab = None
        if any(s in self.vad for s in show_list):
            lab = torch.from_numpy(np.concatenate(
                [np.asarray(self.vad[s], dtype=np.uint8) if s in self.vad else np.ones(n, dtype=np.uint8)
                 for s, n in zip(show_list, lens)])).to(eng.device)
        if not self.has_post_processing() and (keep_all or lab is None):
            lab_out = lab if lab is not None else torch.ones(X.shape[0], dtype=torch.uint8, device=eng.device)
            return X.to(torch.float32).contiguous(), seg, lab_out
        out, lab_out, out_off = self._post_process_device(X.contiguous(), seg, lab, keep_all)
        return out, out_off.cpu().numpy(), lab_out

    # ------------------------------------------------------------------ what a file holds, without loading it
    def _labels_after(self, show):
        """VAD labels of a show as post_processing leaves them (RASTA copies the third label onto the first two,
        features_server.py:322-323), or None when every frame is labelled."""
        lab = self.vad.get(show)
        if lab is None:
            return None
        lab = np.asarray(lab, dtype=bool)
        if self.rasta and lab.shape[0] >= 3:
            lab = lab.copy()
            lab[:2] = lab[2]
        return lab

    def frame_counts(self, show_list, keep_all_features=None):
        """Rows that load(show)[0] returns for each show under the given (default: current) keep_all_features."""
        keep_all = self.keep_all_features if keep_all_features is None else keep_all_features
        out = []
        for show in show_list:
            lab = None if keep_all else self._labels_after(show)
            out.append(int(self.features[show].shape[0]) if lab is None else int(lab.sum()))
        return out

    def selection_is_identity(self, show_list):
        """True when keep_all_features = False drops no frame of these shows."""
        return all(self.frame_counts([s], False) == self.frame_counts([s], True) for s in show_list)

    # ------------------------------------------------------------------ the calls of the hot path
    def load(self, show, channel=0, input_feature_filename=None, label=None, start=None, stop=None):
        cep = self.features[show]
        lab = self.vad.get(show) if label is None else label
        if lab is None:
            lab = np.ones(cep.shape[0], dtype=bool)
        if start is not None or stop is not None:
            cep = cep[start:stop]
            lab = lab[start:stop]
        if self.has_post_processing():
            # the reference applies the mask once more after post_processing (features_server.py:495-496);
            # here it is applied once
            cep, lab = self.post_processing(cep, lab)
        elif not self.keep_all_features and show in self.vad:
            sel = np.asarray(lab, dtype=bool)                 # features_server.py:241-244, selection only
            cep = cep[self.engine.torch.as_tensor(sel).to(cep.device)] if _is_torch(cep) else np.asarray(cep)[sel]
            lab = lab[sel]
        return cep, lab

    def stack_features_parallel(self, show_list, channel_list=None, feature_filename_list=None, label_list=None,
                                start_list=None, stop_list=None, num_thread=1):
        if self.has_post_processing():
            return self.load_batch(show_list)[0]
        blocks = [self.load(show)[0] for show in show_list]
        if getattr(self, "_blocks_wanted", False):          # Mixture._resident_frames uploads the pieces itself
            return blocks
        try:
            import torch
            if any(isinstance(b, torch.Tensor) for b in blocks):
                return torch.cat([b if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).to(blocks[0].device)
                                  for b in blocks], dim=0)
        except ImportError:
            pass
        return np.vstack(blocks)
