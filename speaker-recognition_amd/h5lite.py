"""HDF5 files without h5py: the small part of h5py's API that the drop-in classes use, on top of the HDF5 C library
itself (libhdf5, loaded with ctypes).  The reference stores its products as HDF5 through h5py
(sidekit/mixture.py:280-359, statserver.py:377-476, factor_analyser.py:264-324); h5py is not installed in this image,
libhdf5 is, and h5py is a wrapper around exactly this library -- the files written here are what h5py would write for
the same calls (chunked datasets with the deflate and fletcher32 filters, groups created along a path), and files written
by a real SIDEKIT are read through the same calls h5py would make.

Covered: File(name, "r" | "w" | "a" | "r+"), context manager, close(); create_dataset(name, shape=None, dtype=None,
data=None, maxshape=None, compression=None | "gzip", fletcher32=False); f[name] -> Dataset with .shape, .dtype, [()],
[index], .value, .resize(shape), dset[n:] = rows; f.get(name); name in f; keys().  Types: float32 / float64, (u)int8..64, fixed-length byte strings.
"""
import ctypes
import ctypes.util
import os

import numpy as np

_hid = ctypes.c_int64
_hsz = ctypes.c_uint64
_lib = None

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_STR_NULLPAD = 1
H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF
H5_INDEX_NAME, H5_ITER_INC = 0, 2


def _candidates():
    env = os.environ.get("SKM_LIBHDF5")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/lib64",
              "/usr/local/lib"):
        for n in ("libhdf5.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5_serial.so", "libhdf5_serial.so.103"):
            yield os.path.join(d, n)


def _load():
    global _lib
    if _lib is not None:
        return _lib
    err = None
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:
            err = e
            continue
        if lib.H5open() < 0:
            continue
        sig = {
            "H5Fcreate": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
            "H5Fopen": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
            "H5Fclose": (ctypes.c_int, [_hid]),
            "H5Screate_simple": (_hid, [ctypes.c_int, ctypes.POINTER(_hsz), ctypes.POINTER(_hsz)]),
            "H5Screate": (_hid, [ctypes.c_int]),
            "H5Sclose": (ctypes.c_int, [_hid]),
            "H5Sget_simple_extent_ndims": (ctypes.c_int, [_hid]),
            "H5Sget_simple_extent_dims": (ctypes.c_int, [_hid, ctypes.POINTER(_hsz), ctypes.POINTER(_hsz)]),
            "H5Pcreate": (_hid, [_hid]),
            "H5Pclose": (ctypes.c_int, [_hid]),
            "H5Pset_chunk": (ctypes.c_int, [_hid, ctypes.c_int, ctypes.POINTER(_hsz)]),
            "H5Pset_deflate": (ctypes.c_int, [_hid, ctypes.c_uint]),
            "H5Pset_fletcher32": (ctypes.c_int, [_hid]),
            "H5Pset_create_intermediate_group": (ctypes.c_int, [_hid, ctypes.c_uint]),
            "H5Dcreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
            "H5Dopen2": (_hid, [_hid, ctypes.c_char_p, _hid]),
            "H5Dclose": (ctypes.c_int, [_hid]),
            "H5Dwrite": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            "H5Dread": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            "H5Dget_type": (_hid, [_hid]),
            "H5Dget_space": (_hid, [_hid]),
            "H5Tcopy": (_hid, [_hid]),
            "H5Tclose": (ctypes.c_int, [_hid]),
            "H5Tset_size": (ctypes.c_int, [_hid, ctypes.c_size_t]),
            "H5Tset_strpad": (ctypes.c_int, [_hid, ctypes.c_int]),
            "H5Tget_class": (ctypes.c_int, [_hid]),
            "H5Tget_size": (ctypes.c_size_t, [_hid]),
            "H5Tget_sign": (ctypes.c_int, [_hid]),
            "H5Tis_variable_str": (ctypes.c_int, [_hid]),
            "H5Lexists": (ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
            "H5Dset_extent": (ctypes.c_int, [_hid, ctypes.POINTER(_hsz)]),
            "H5Sselect_hyperslab": (ctypes.c_int, [_hid, ctypes.c_int, ctypes.POINTER(_hsz), ctypes.POINTER(_hsz),
                                                   ctypes.POINTER(_hsz), ctypes.POINTER(_hsz)]),
            "H5Eset_auto2": (ctypes.c_int, [_hid, ctypes.c_void_p, ctypes.c_void_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib.H5Eset_auto2(0, None, None)        # errors come back as return codes, not as a stack dump on stderr
        _lib = lib
        return lib
    raise ImportError("libhdf5 not found (set SKM_LIBHDF5 to its path): %s" % err)


def available():
    try:
        _load()
        return True
    except ImportError:
        return False


def _g(name):
    return _hid.in_dll(_load(), name).value


def _file_type(dt):
    """(type id, owned) for a numpy dtype."""
    lib = _load()
    dt = np.dtype(dt)
    if dt.kind == "S":
        t = lib.H5Tcopy(_g("H5T_C_S1_g"))
        lib.H5Tset_size(t, max(dt.itemsize, 1))
        lib.H5Tset_strpad(t, H5T_STR_NULLPAD)
        return t, True
    names = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g", "i1": "H5T_NATIVE_INT8_g", "i2": "H5T_NATIVE_INT16_g",
             "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g", "u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g",
             "u4": "H5T_NATIVE_UINT32_g", "u8": "H5T_NATIVE_UINT64_g"}
    key = dt.kind + str(dt.itemsize)
    if key not in names:
        raise TypeError("h5lite: dtype %s is not supported" % dt)
    return _g(names[key]), False


def _numpy_type(tid):
    lib = _load()
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == H5T_FLOAT and size in (4, 8):
        return np.dtype("f%d" % size)
    if cls == H5T_INTEGER and size in (1, 2, 4, 8):
        return np.dtype(("i%d" if lib.H5Tget_sign(tid) else "u%d") % size)
    if cls == H5T_STRING and lib.H5Tis_variable_str(tid) <= 0:
        return np.dtype("S%d" % size)
    raise TypeError("h5lite: HDF5 type class %d of size %d is not supported" % (cls, size))


class Dataset(object):
    def __init__(self, did, name):
        self._id, self.name = did, name
        lib = _load()
        sp = lib.H5Dget_space(did)
        nd = lib.H5Sget_simple_extent_ndims(sp)
        dims = (_hsz * max(nd, 1))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(sp, dims, None)
        lib.H5Sclose(sp)
        self.shape = tuple(int(dims[i]) for i in range(nd))
        tid = lib.H5Dget_type(did)
        try:
            self.dtype = _numpy_type(tid)
        finally:
            lib.H5Tclose(tid)

    def _read(self):
        lib = _load()
        out = np.empty(self.shape, dtype=self.dtype)
        if out.size:
            mt, owned = _file_type(self.dtype)
            rc = lib.H5Dread(self._id, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p))
            if owned:
                lib.H5Tclose(mt)
            if rc < 0:
                raise IOError("h5lite: cannot read dataset %s (checksum or filter failure?)" % self.name)
        return out

    def __getitem__(self, key):
        a = self._read()
        if isinstance(key, tuple) and key == ():
            return a if a.ndim else a[()]
        return a[key]

    @property
    def value(self):
        return self._read()

    def __len__(self):
        return self.shape[0]

    def resize(self, shape):
        """New extent of a dataset created with maxshape (h5py's Dataset.resize)."""
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        assert len(shape) == len(self.shape)
        if _load().H5Dset_extent(self._id, (_hsz * len(shape))(*shape)) < 0:
            raise IOError("h5lite: cannot resize dataset %s to %s (created without maxshape?)" % (self.name, shape))
        self.shape = shape

    def __setitem__(self, key, value):
        """dset[n:] = rows / dset[n:, :] = rows: the rows from n on (what StatServer.write(mode='a') needs)."""
        lib = _load()
        k0 = key[0] if isinstance(key, tuple) else key
        rest = key[1:] if isinstance(key, tuple) else ()
        if not (isinstance(k0, slice) and k0.step in (None, 1) and all(r == slice(None) for r in rest)):
            raise NotImplementedError("h5lite: only dset[n:m] = rows")
        n0, n1, _ = k0.indices(self.shape[0])
        a = np.asarray(value)
        if a.dtype.kind == "U":
            a = a.astype("S")
        if a.dtype.kind != "S":
            a = a.astype(self.dtype)
        a = np.ascontiguousarray(a)
        assert a.shape == (n1 - n0,) + self.shape[1:], (a.shape, (n1 - n0,) + self.shape[1:])
        if a.size == 0:
            return
        nd = len(self.shape)
        fs = lib.H5Dget_space(self._id)
        start = (_hsz * nd)(n0, *([0] * (nd - 1)))
        count = (_hsz * nd)(*a.shape)
        lib.H5Sselect_hyperslab(fs, 0, start, None, count, None)
        ms = lib.H5Screate_simple(nd, count, None)
        mt, owned = _file_type(a.dtype)
        rc = lib.H5Dwrite(self._id, mt, ms, fs, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p))
        if owned:
            lib.H5Tclose(mt)
        lib.H5Sclose(ms)
        lib.H5Sclose(fs)
        if rc < 0:
            raise IOError("h5lite: cannot write rows %d..%d of dataset %s" % (n0, n1, self.name))

    def _close(self):
        if self._id:
            _load().H5Dclose(self._id)
            self._id = 0


class File(object):
    def __init__(self, name, mode="r"):
        lib = _load()
        self.filename = name
        self._open = []
        b = os.fsencode(name)
        if mode == "w" or (mode == "a" and not os.path.exists(name)):
            self._id = lib.H5Fcreate(b, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode in ("a", "r+"):
            self._id = lib.H5Fopen(b, H5F_ACC_RDWR, H5P_DEFAULT)
        elif mode == "r":
            self._id = lib.H5Fopen(b, H5F_ACC_RDONLY, H5P_DEFAULT)
        else:
            raise ValueError("h5lite: mode %r" % (mode,))
        if self._id < 0:
            raise IOError("h5lite: cannot open %s (mode %s)" % (name, mode))

    # ---------------------------------------------------------------- context manager
    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def close(self):
        for d in self._open:
            d._close()
        self._open = []
        if self._id:
            _load().H5Fclose(self._id)
            self._id = 0

    # ---------------------------------------------------------------- lookup
    def __contains__(self, name):
        lib = _load()
        path = ""
        for part in [p for p in name.split("/") if p]:
            path = path + "/" + part
            if lib.H5Lexists(self._id, path.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def __getitem__(self, name):
        if name not in self:
            raise KeyError(name)
        did = _load().H5Dopen2(self._id, name.encode(), H5P_DEFAULT)
        if did < 0:
            raise KeyError("%s is not a dataset" % name)
        d = Dataset(did, name)
        self._open.append(d)
        return d

    def get(self, name, default=None):
        return self[name] if name in self else default

    def keys(self):
        """Names of the links of the root group."""
        lib = _load()
        names = []
        cb_t = ctypes.CFUNCTYPE(ctypes.c_int, _hid, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)

        def cb(g, name, info, data):
            names.append(name.decode())
            return 0
        # HDF5 1.12+ exports H5Literate1 / H5Literate2 (H5Literate is a macro there); the callback ignores the info
        # argument, so either signature serves
        fn = next((getattr(lib, n) for n in ("H5Literate1", "H5Literate", "H5Literate2") if hasattr(lib, n)), None)
        if fn is None:
            raise IOError("h5lite: this libhdf5 exports no H5Literate")
        fn.restype = ctypes.c_int
        fn.argtypes = [_hid, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_hsz), cb_t, ctypes.c_void_p]
        idx = _hsz(0)
        fn(self._id, H5_INDEX_NAME, H5_ITER_INC, ctypes.byref(idx), cb_t(cb), None)
        return names

    # ---------------------------------------------------------------- writing
    def create_dataset(self, name, shape=None, dtype=None, data=None, maxshape=None, compression=None, fletcher32=False,
                       **unused):
        lib = _load()
        if data is not None:
            a = np.asarray(data) if dtype is None else np.asarray(data, dtype=dtype)
            if a.ndim and not a.flags.c_contiguous:     # (ascontiguousarray would turn a scalar into a vector)
                a = np.ascontiguousarray(a)
            if shape is not None and tuple(np.atleast_1d(shape)) != a.shape and a.size == int(np.prod(shape)):
                a = a.reshape(shape)
        else:
            a = np.zeros(shape if shape is not None else (), dtype=dtype or "f8")
        if a.dtype.kind == "U":
            a = a.astype("S")
        if a.dtype.kind == "b":
            a = a.astype("u1")
        ft, owned = _file_type(a.dtype)
        dcpl = lib.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
        lcpl = lib.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g"))
        lib.H5Pset_create_intermediate_group(lcpl, 1)
        if a.ndim == 0:
            sp = lib.H5Screate(H5S_SCALAR)
        else:
            dims = (_hsz * a.ndim)(*a.shape)
            mx = None
            # a dataset that may grow is chunked even when it starts empty (a later resize needs the chunked layout)
            filtered = maxshape is not None or (a.size > 0 and (compression is not None or fletcher32))
            if filtered:
                if maxshape is not None:
                    mx = (_hsz * a.ndim)(*[H5S_UNLIMITED if m is None else int(m) for m in maxshape])
                # chunks of about 1 MiB along the first dimension (whole rows)
                row = max(1, int(np.prod(a.shape[1:], dtype=np.int64)) * a.dtype.itemsize)
                c0 = int(max(1, min(max(a.shape[0], 1), (1 << 20) // row)))
                chunk = (_hsz * a.ndim)(c0, *[max(1, int(s)) for s in a.shape[1:]])
                lib.H5Pset_chunk(dcpl, a.ndim, chunk)
                # filter order as h5py's: checksum first, then deflate
                if fletcher32:
                    lib.H5Pset_fletcher32(dcpl)
                if compression is not None:
                    if compression != "gzip":
                        raise ValueError("h5lite: only gzip compression")
                    lib.H5Pset_deflate(dcpl, 4)
            sp = lib.H5Screate_simple(a.ndim, dims, mx)
        did = lib.H5Dcreate2(self._id, name.encode(), ft, sp, lcpl, dcpl, H5P_DEFAULT)
        rc = 0
        if did >= 0 and a.size:
            rc = lib.H5Dwrite(did, ft, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p))
        lib.H5Sclose(sp)
        lib.H5Pclose(dcpl)
        lib.H5Pclose(lcpl)
        if owned:
            lib.H5Tclose(ft)
        if did < 0 or rc < 0:
            if did >= 0:
                lib.H5Dclose(did)
            raise IOError("h5lite: cannot create / write dataset %s in %s" % (name, self.filename))
        d = Dataset(did, name)
        self._open.append(d)
        return d
