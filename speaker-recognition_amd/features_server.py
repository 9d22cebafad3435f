"""Input boundary of the hot path: a features server hands frames to Mixture.EM_split and
StatServer.accumulate_stat.  The reference's FeaturesServer (sidekit/features_server.py) reads
HDF5 files and post-processes them on the host; that data plane is out of scope (SURVEY.md
section 2, row 8).  This in-memory server keeps the two calls the hot path makes:

    load(show, channel=0, input_feature_filename=None, label=None, start=None, stop=None)
        -> (T x D array, bool[T] VAD labels)                  (features_server.py:397-451)
    stack_features_parallel(show_list, ..., num_thread=1) -> N x D array   (features_server.py:603-666)

plus the attributes accumulate_stat touches (keep_all_features, features_extractor,
feature_filename_structure; statserver.py:631-645).  Values may be NumPy arrays or torch
tensors already resident on the GPU.
"""
import numpy as np


class FeaturesServer(object):
    def __init__(self, features=None, vad=None, features_extractor=None, feature_filename_structure=None,
                 dataset_list=None, keep_all_features=True, is_sharded=False, **_ignored):
        self.features = dict(features) if features is not None else {}
        self.vad = dict(vad) if vad is not None else {}
        self.features_extractor = features_extractor
        self.feature_filename_structure = feature_filename_structure
        self.dataset_list = dataset_list
        self.keep_all_features = keep_all_features
        # True when every rank of a multi-GPU job was given only ITS OWN frames already
        self.is_sharded = is_sharded

    def add(self, show, cep, vad=None):
        self.features[show] = cep
        if vad is not None:
            self.vad[show] = np.asarray(vad, dtype=bool)

    def load(self, show, channel=0, input_feature_filename=None, label=None, start=None, stop=None):
        cep = self.features[show]
        lab = self.vad.get(show)
        if lab is None:
            lab = np.ones(cep.shape[0], dtype=bool)
        if start is not None or stop is not None:
            cep = cep[start:stop]
            lab = lab[start:stop]
        return cep, lab

    def stack_features_parallel(self, show_list, channel_list=None, feature_filename_list=None, label_list=None,
                                start_list=None, stop_list=None, num_thread=1):
        blocks = [self.load(show)[0] for show in show_list]
        try:
            import torch
            if any(isinstance(b, torch.Tensor) for b in blocks):
                return torch.cat([b if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).to(blocks[0].device)
                                  for b in blocks], dim=0)
        except ImportError:
            pass
        return np.vstack(blocks)
