"""Session list container: the four-array contract of sidekit/bosaris/idmap.py:54-73 that
StatServer is built from (sidekit/statserver.py:230-246)."""
import numpy as np


class IdMap(object):
    def __init__(self, idmap_filename=''):
        self.leftids = np.empty(0, dtype="|O")
        self.rightids = np.empty(0, dtype="|O")
        self.start = np.empty(0, dtype="|O")
        self.stop = np.empty(0, dtype="|O")
        if idmap_filename != '':
            tmp = IdMap.read(idmap_filename)
            self.leftids, self.rightids, self.start, self.stop = tmp.leftids, tmp.rightids, tmp.start, tmp.stop

    def set(self, left, right, start=None, stop=None):
        self.leftids = np.array(left)
        self.rightids = np.array(right)
        self.start = np.empty(self.rightids.shape, '|O') if start is None else np.array(start, dtype='|O')
        self.stop = np.empty(self.rightids.shape, '|O') if stop is None else np.array(stop, dtype='|O')
        return self

    def validate(self, warn=False):
        ok = (self.leftids.shape == self.rightids.shape == self.start.shape == self.stop.shape) \
            and self.leftids.ndim == 1
        return bool(ok)

    def write(self, output_file_name):
        start = np.array([-1 if v is None else int(v) for v in self.start], dtype=np.int64)
        stop = np.array([-1 if v is None else int(v) for v in self.stop], dtype=np.int64)
        np.savez(output_file_name, leftids=self.leftids.astype('U'), rightids=self.rightids.astype('U'),
                 start=start, stop=stop)

    @staticmethod
    def read(input_file_name):
        z = np.load(input_file_name, allow_pickle=False)
        m = IdMap()
        m.leftids = z["leftids"]
        m.rightids = z["rightids"]
        m.start = np.empty(m.rightids.shape, '|O')
        m.stop = np.empty(m.rightids.shape, '|O')
        m.start[z["start"] != -1] = z["start"][z["start"] != -1]
        m.stop[z["stop"] != -1] = z["stop"][z["stop"] != -1]
        return m


class Ndx(object):
    """Trial index: modelset [M], segset [N], trialmask [M x N] bool -- the three-array contract of
    sidekit/bosaris/ndx.py:63-76 that gmm_scoring reads (sidekit/gmm_scoring.py:87-93)."""

    def __init__(self, ndx_file_name='', models=np.array([]), testsegs=np.array([])):
        self.modelset = np.empty(0, dtype="|O")
        self.segset = np.empty(0, dtype="|O")
        self.trialmask = np.array([], dtype="bool")
        if ndx_file_name == '':
            # one trial per (model, segment) pair of the two lists (ndx.py:78-96)
            models, testsegs = np.asarray(models), np.asarray(testsegs)
            if models.shape[0] and models.shape == testsegs.shape:
                self.modelset = np.unique(models)
                self.segset = np.unique(testsegs)
                self.trialmask = np.zeros((self.modelset.shape[0], self.segset.shape[0]), dtype="bool")
                mi = {m: i for i, m in enumerate(self.modelset)}
                si = {s: i for i, s in enumerate(self.segset)}
                for m, s in zip(models, testsegs):
                    self.trialmask[mi[m], si[s]] = True
        else:
            z = np.load(ndx_file_name, allow_pickle=False)
            self.modelset, self.segset, self.trialmask = z["modelset"], z["segset"], z["trialmask"].astype(bool)

    def write(self, output_file_name):
        np.savez(output_file_name, modelset=np.asarray(self.modelset).astype('U'),
                 segset=np.asarray(self.segset).astype('U'), trialmask=self.trialmask.astype(np.int8))

    def filter(self, modlist, seglist, keep):
        """ndx.py:128-165."""
        modlist, seglist = np.asarray(modlist), np.asarray(seglist)
        km = np.isin(self.modelset, modlist)
        ks = np.isin(self.segset, seglist)
        if not keep:
            km, ks = ~km, ~ks
        out = Ndx()
        out.modelset = self.modelset[km]
        out.segset = self.segset[ks]
        out.trialmask = self.trialmask[km, :][:, ks]
        return out

    def validate(self):
        """ndx.py:167-184."""
        return bool(isinstance(self.trialmask, np.ndarray) and self.modelset.ndim == 1 and self.segset.ndim == 1
                    and self.trialmask.ndim == 2
                    and self.trialmask.shape == (self.modelset.shape[0], self.segset.shape[0]))


class Scores(object):
    """Score matrix container (sidekit/bosaris/scores.py:60-77): modelset, segset, scoremask, scoremat."""

    def __init__(self):
        self.modelset = np.empty(0, dtype="|O")
        self.segset = np.empty(0, dtype="|O")
        self.scoremask = np.array([], dtype="bool")
        self.scoremat = np.array([])

    def validate(self):
        return bool(self.scoremat.shape == self.scoremask.shape
                    and self.scoremat.shape == (self.modelset.shape[0], self.segset.shape[0]))
