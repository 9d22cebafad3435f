"""Session list container: the four-array contract of sidekit/bosaris/idmap.py:54-73 that
StatServer is built from (sidekit/statserver.py:230-246)."""
import numpy as np


def _is_h5(name):
    return str(name).endswith((".h5", ".hdf5"))


def _h5():
    from .mixture import _h5 as h
    return h()


def _h5_ids(f, name, width):
    """A fixed-width byte-string dataset as unicode (idmap.py:284-290, ndx.py:192-198)."""
    return np.asarray(f[name][()]).astype("U%d" % width)


def _pairs_to_mask(models, testsegs):
    """(modelset, segset, mask) of a list of (model, segment) pairs (ndx.py:217-229)."""
    modelset, mi = np.unique(models, return_inverse=True)
    segset, si = np.unique(testsegs, return_inverse=True)
    mask = np.zeros((modelset.shape[0], segset.shape[0]), dtype="bool")
    mask[mi, si] = True
    return modelset, segset, mask, mi, si


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
        """`.h5`: the four datasets of idmap.py:84-115 (byte strings, int32 start/stop with -1 for "whole
        file", gzip + fletcher32, growable).  Any other name: the same arrays as .npz."""
        assert self.validate(), "Error: wrong IdMap format"
        start = np.array([-1 if v is None else int(v) for v in self.start], dtype=np.int32)
        stop = np.array([-1 if v is None else int(v) for v in self.stop], dtype=np.int32)
        if _is_h5(output_file_name):
            with _h5().File(output_file_name, "w") as f:
                for name, a in (("leftids", self.leftids.astype('S')), ("rightids", self.rightids.astype('S')),
                                ("start", start), ("stop", stop)):
                    f.create_dataset(name, data=a, maxshape=(None,), compression="gzip", fletcher32=True)
            return
        np.savez(output_file_name, leftids=self.leftids.astype('U'), rightids=self.rightids.astype('U'),
                 start=start, stop=stop)

    def write_txt(self, output_file_name):
        """idmap.py:118-126: `left right [start stop]` per line."""
        with open(output_file_name, 'w') as fh:
            for left, right, start, stop in zip(self.leftids, self.rightids, self.start, self.stop):
                fh.write(' '.join(filter(None, (left, right, str(start), str(stop)))) + '\n')

    @staticmethod
    def read(input_file_name):
        """idmap.py:276-300."""
        m = IdMap()
        if _is_h5(input_file_name):
            with _h5().File(input_file_name, "r") as f:
                m.leftids, m.rightids = _h5_ids(f, "leftids", 255), _h5_ids(f, "rightids", 255)
                start, stop = np.asarray(f["start"][()]), np.asarray(f["stop"][()])
        else:
            z = np.load(input_file_name, allow_pickle=False)
            m.leftids, m.rightids, start, stop = z["leftids"], z["rightids"], z["start"], z["stop"]
        m.start = np.empty(m.rightids.shape, '|O')
        m.stop = np.empty(m.rightids.shape, '|O')
        m.start[start != -1] = start[start != -1]
        m.stop[stop != -1] = stop[stop != -1]
        assert m.validate(), "Error: wrong IdMap format"
        return m

    @classmethod
    def read_txt(cls, input_file_name):
        """idmap.py:302-331: two columns (ids) or four (ids, start, stop)."""
        with open(input_file_name, 'r') as fh:
            rows = [l.split() for l in fh if l.strip()]
        m = cls()
        m.leftids = np.array([r[0] for r in rows], dtype='|O')
        m.rightids = np.array([r[1] for r in rows], dtype='|O')
        m.start = np.empty(m.rightids.shape, '|O')
        m.stop = np.empty(m.rightids.shape, '|O')
        if rows and len(rows[0]) == 4:
            for i, r in enumerate(rows):
                m.start[i] = None if r[2] == 'None' else int(r[2])
                m.stop[i] = None if r[3] == 'None' else int(r[3])
        if not m.validate():
            raise Exception('Wrong format of IdMap')
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
            tmp = Ndx.read(ndx_file_name)
            self.modelset, self.segset, self.trialmask = tmp.modelset, tmp.segset, tmp.trialmask

    def write(self, output_file_name):
        """`.h5`: modelset, segset, trial_mask (int8) as ndx.py:93-112; any other name: .npz."""
        assert self.validate(), "Error: wrong Ndx format"
        if _is_h5(output_file_name):
            with _h5().File(output_file_name, "w") as f:
                f.create_dataset("modelset", data=np.asarray(self.modelset).astype('S'), maxshape=(None,),
                                 compression="gzip", fletcher32=True)
                f.create_dataset("segset", data=np.asarray(self.segset).astype('S'), maxshape=(None,),
                                 compression="gzip", fletcher32=True)
                f.create_dataset("trial_mask", data=self.trialmask.astype('int8'), maxshape=(None, None),
                                 compression="gzip", fletcher32=True)
            return
        np.savez(output_file_name, modelset=np.asarray(self.modelset).astype('U'),
                 segset=np.asarray(self.segset).astype('U'), trialmask=self.trialmask.astype(np.int8))

    def save_txt(self, output_file_name):
        """ndx.py:115-126: one `model segment` line per trial."""
        with open(output_file_name, 'w') as fh:
            for m in range(self.modelset.shape[0]):
                for s in self.segset[self.trialmask[m, ]]:
                    fh.write('{} {}\n'.format(self.modelset[m], s))

    @staticmethod
    def read(input_file_name):
        """ndx.py:185-203."""
        ndx = Ndx()
        if _is_h5(input_file_name):
            with _h5().File(input_file_name, "r") as f:
                ndx.modelset, ndx.segset = _h5_ids(f, "modelset", 100), _h5_ids(f, "segset", 100)
                ndx.trialmask = np.asarray(f["trial_mask"][()]).astype('bool')
        else:
            z = np.load(input_file_name, allow_pickle=False)
            ndx.modelset, ndx.segset, ndx.trialmask = z["modelset"], z["segset"], z["trialmask"].astype(bool)
        assert ndx.validate(), "Error: wrong Ndx format"
        return ndx

    @classmethod
    def read_txt(cls, input_filename):
        """ndx.py:207-236."""
        with open(input_filename, 'r') as fh:
            rows = [l.split() for l in fh if l.strip()]
        ndx = cls()
        ndx.modelset, ndx.segset, ndx.trialmask, _, _ = _pairs_to_mask(
            np.array([r[0] for r in rows], dtype='|O'), np.array([r[1] for r in rows], dtype='|O'))
        assert ndx.validate(), "Wrong Ndx format"
        return ndx

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

    def _write_h5(self, output_file_name, row_name, col_name):
        with _h5().File(output_file_name, "w") as f:
            f.create_dataset(row_name, data=np.asarray(self.modelset).astype('S'), maxshape=(None,),
                             compression="gzip", fletcher32=True)
            f.create_dataset(col_name, data=np.asarray(self.segset).astype('S'), maxshape=(None,),
                             compression="gzip", fletcher32=True)
            f.create_dataset("score_mask", data=self.scoremask.astype('int8'), maxshape=(None, None),
                             compression="gzip", fletcher32=True)
            f.create_dataset("scores", data=np.asarray(self.scoremat), maxshape=(None, None),
                             compression="gzip", fletcher32=True)

    @staticmethod
    def _read_h5(input_file_name, row_name, col_name):
        s = Scores()
        with _h5().File(input_file_name, "r") as f:
            s.modelset, s.segset = _h5_ids(f, row_name, 100), _h5_ids(f, col_name, 100)
            s.scoremask = np.asarray(f["score_mask"][()]).astype('bool')
            s.scoremat = np.asarray(f["scores"][()])
        assert s.validate(), "Error: wrong Scores format"
        return s

    def write(self, output_file_name):
        """`.h5`: modelset, segset, score_mask (int8), scores as scores.py:94-115; any other name: .npz."""
        if _is_h5(output_file_name):
            return self._write_h5(output_file_name, "modelset", "segset")
        np.savez(output_file_name, modelset=np.asarray(self.modelset).astype('U'),
                 segset=np.asarray(self.segset).astype('U'), scoremask=self.scoremask.astype(np.int8),
                 scoremat=np.asarray(self.scoremat))

    def write_matlab(self, output_file_name):
        """The BOSARIS toolkit's HDF5 layout (scores.py:134-155): ids under /ID/row_ids and /ID/column_ids."""
        self._write_h5(output_file_name, "/ID/row_ids", "/ID/column_ids")

    def write_txt(self, output_file_name):
        """scores.py:118-131: `model segment score` per scored trial."""
        with open(output_file_name, 'w') as fh:
            for m in range(self.modelset.shape[0]):
                segs = self.segset[self.scoremask[m, ]]
                scores = self.scoremat[m, self.scoremask[m, ]]
                for s in range(segs.shape[0]):
                    fh.write('{} {} {}\n'.format(self.modelset[m], segs[s], scores[s]))

    @staticmethod
    def read(input_file_name):
        """scores.py:310-334."""
        if _is_h5(input_file_name):
            return Scores._read_h5(input_file_name, "modelset", "segset")
        z = np.load(input_file_name, allow_pickle=False)
        s = Scores()
        s.modelset, s.segset = z["modelset"], z["segset"]
        s.scoremask, s.scoremat = z["scoremask"].astype(bool), z["scoremat"]
        assert s.validate(), "Error: wrong Scores format"
        return s

    @staticmethod
    def read_matlab(input_file_name):
        """scores.py:337-361."""
        return Scores._read_h5(input_file_name, "ID/row_ids", "ID/column_ids")

    @classmethod
    def read_txt(cls, input_file_name):
        """scores.py:365-406 (unique() leaves both id lists sorted, which is what its final sort() does)."""
        with open(input_file_name, 'r') as fh:
            rows = [l.split() for l in fh if l.strip()]
        s = cls()
        s.modelset, s.segset, s.scoremask, mi, si = _pairs_to_mask(
            np.array([r[0] for r in rows], dtype='|O'), np.array([r[1] for r in rows], dtype='|O'))
        s.scoremat = np.zeros(s.scoremask.shape)
        s.scoremat[mi, si] = [float(r[2]) for r in rows]
        assert s.validate(), "Wrong Scores format"
        return s

    def get_tar_non(self, key):
        """Target and non-target scores of the trials a Key marks (scores.py:157-181); `key` has modelset,
        segset, tar and non."""
        mi = {m: i for i, m in enumerate(self.modelset)}
        si = {t: i for i, t in enumerate(self.segset)}
        rows = np.array([mi.get(m, -1) for m in key.modelset])
        cols = np.array([si.get(t, -1) for t in key.segset])
        ok = (rows >= 0)[:, None] & (cols >= 0)[None, :]
        sub_mask = np.zeros(ok.shape, dtype=bool)
        sub = np.zeros(ok.shape)
        rr, cc = np.nonzero(ok)
        sub_mask[rr, cc] = self.scoremask[rows[rr], cols[cc]]
        sub[rr, cc] = self.scoremat[rows[rr], cols[cc]]
        return sub[np.asarray(key.tar, bool) & sub_mask], sub[np.asarray(key.non, bool) & sub_mask]


class Key(object):
    """Trial key (sidekit/bosaris/key.py:49-107): modelset [M], segset [N], tar / non [M x N] bool -- which trials are
    target and which are non-target trials; what Scores.get_tar_non splits a score matrix with."""

    def __init__(self, key_file_name='', models=np.array([]), testsegs=np.array([]), trials=np.array([])):
        self.modelset = np.empty(0, dtype="|O")
        self.segset = np.empty(0, dtype="|O")
        self.tar = np.array([], dtype="bool")
        self.non = np.array([], dtype="bool")
        if key_file_name != '':
            tmp = Key.read(key_file_name)
            self.modelset, self.segset, self.tar, self.non = tmp.modelset, tmp.segset, tmp.tar, tmp.non
            return
        models, testsegs, trials = np.asarray(models), np.asarray(testsegs), np.asarray(trials)
        self._from_lists(models, testsegs, trials)

    def _from_lists(self, models, testsegs, trials):
        """One (model, segment, 'target' | 'nontarget') triple per trial (key.py:78-98); the last triple of a pair wins."""
        self.modelset, mi = np.unique(models, return_inverse=True)
        self.segset, si = np.unique(testsegs, return_inverse=True)
        self.tar = np.zeros((self.modelset.shape[0], self.segset.shape[0]), dtype="bool")
        self.non = np.zeros((self.modelset.shape[0], self.segset.shape[0]), dtype="bool")
        if models.shape[0]:
            self.tar[mi, si] = (trials == 'target')          # later assignments overwrite earlier ones, as the dict does
            self.non[mi, si] = (trials == 'nontarget')
        assert self.validate(), "Wrong Key format"

    def validate(self):
        """key.py:199-216."""
        return bool(isinstance(self.modelset, np.ndarray) and isinstance(self.segset, np.ndarray)
                    and isinstance(self.tar, np.ndarray) and isinstance(self.non, np.ndarray)
                    and self.modelset.ndim == 1 and self.segset.ndim == 1 and self.tar.ndim == 2 and self.non.ndim == 2
                    and self.tar.shape == self.non.shape
                    and self.tar.shape == (self.modelset.shape[0], self.segset.shape[0]))

    def to_ndx(self):
        """key.py:188-197."""
        ndx = Ndx()
        ndx.modelset, ndx.segset, ndx.trialmask = self.modelset, self.segset, self.tar | self.non
        return ndx

    def filter(self, modlist, seglist, keep):
        """key.py:147-186."""
        km, ks = np.isin(self.modelset, np.asarray(modlist)), np.isin(self.segset, np.asarray(seglist))
        if not keep:
            km, ks = ~km, ~ks
        out = Key()
        out.modelset, out.segset = self.modelset[km], self.segset[ks]
        out.tar, out.non = self.tar[km, :][:, ks], self.non[km, :][:, ks]
        assert out.validate(), "Wrong Key format"
        return out

    def write(self, output_file_name):
        """`.h5`: modelset, segset, trial_mask = tar - non as int8 (key.py:109-129); any other name: .npz."""
        assert self.validate(), "Error: wrong Key format"
        mask = self.tar.astype('int8') - self.non.astype('int8')
        if _is_h5(output_file_name):
            with _h5().File(output_file_name, "w") as f:
                f.create_dataset("modelset", data=np.asarray(self.modelset).astype('S'), maxshape=(None,),
                                 compression="gzip", fletcher32=True)
                f.create_dataset("segset", data=np.asarray(self.segset).astype('S'), maxshape=(None,),
                                 compression="gzip", fletcher32=True)
                f.create_dataset("trial_mask", data=mask, maxshape=(None, None), compression="gzip", fletcher32=True)
            return
        np.savez(output_file_name, modelset=np.asarray(self.modelset).astype('U'),
                 segset=np.asarray(self.segset).astype('U'), trial_mask=mask)

    def write_txt(self, output_file_name):
        """key.py:132-145: `model segment target|nontarget` per trial."""
        with open(output_file_name, 'w') as fh:
            for m in range(self.modelset.shape[0]):
                for s in self.segset[self.tar[m, ]]:
                    fh.write('{} {} {}\n'.format(self.modelset[m], s, 'target'))
                for s in self.segset[self.non[m, ]]:
                    fh.write('{} {} {}\n'.format(self.modelset[m], s, 'nontarget'))

    @staticmethod
    def read(input_file_name):
        """key.py:219-240."""
        key = Key()
        if _is_h5(input_file_name):
            with _h5().File(input_file_name, "r") as f:
                key.modelset, key.segset = _h5_ids(f, "modelset", 100), _h5_ids(f, "segset", 100)
                mask = np.asarray(f["trial_mask"][()])
        else:
            z = np.load(input_file_name, allow_pickle=False)
            key.modelset, key.segset, mask = z["modelset"], z["segset"], z["trial_mask"]
        key.tar, key.non = (mask == 1), (mask == -1)
        assert key.validate(), "Error: wrong Key format"
        return key

    @staticmethod
    def read_txt(input_file_name):
        """key.py:243-284."""
        with open(input_file_name, 'r') as fh:
            rows = [l.split() for l in fh if l.strip()]
        key = Key()
        key._from_lists(np.array([r[0] for r in rows]), np.array([r[1] for r in rows]), np.array([r[2] for r in rows]))
        return key
