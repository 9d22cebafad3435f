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
