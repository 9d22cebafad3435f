"""HDF5 file products without h5py (speaker-recognition_amd/h5lite.py: libhdf5 through ctypes): the reference's on-disk
layouts -- Mixture (mixture.py:325-359), StatServer (statserver.py:418-476, float32 statistics, byte-string ids, -1 for
missing start / stop), FactorAnalyser (factor_analyser.py:264-324, fa/* group with the kind vector) -- written and read
back through the drop-in classes, and the raw API on the types they use.  h5py is not installed here, so these files are
checked against what the HDF5 library itself reports, not against h5py ("parity unpinned" for h5py-written files)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sidekit_amd as sk  # noqa: E402
h5lite = sk.h5lite

pytestmark = pytest.mark.skipif(not h5lite.available(), reason="libhdf5 not found")


def test_raw_api_round_trip(tmp_path):
    fn = str(tmp_path / "raw.h5")
    rs = np.random.RandomState(0)
    data = {"f8": rs.randn(7, 5), "f4": rs.randn(300, 40).astype(np.float32), "i8": np.arange(-3, 9, dtype=np.int64),
            "i2": np.array([0, 1, 1, 0, 1], dtype=np.int16), "names": np.array(["spk1", "a", "longer-name"]).astype("S"),
            "grp/sub/x": rs.randn(4), "empty": np.zeros((0,)), "scalar": np.float64(3.5)}
    with h5lite.File(fn, "w") as f:
        for k, v in data.items():
            kw = dict(compression="gzip", fletcher32=True) if np.ndim(v) else {}
            f.create_dataset(k, data=v, **kw)
        f.create_dataset("growable", data=data["f4"], maxshape=(None, 40), compression="gzip", fletcher32=True)
        f.create_dataset("positional", (7, 5), "d", data["f8"], compression="gzip", fletcher32=True)   # mixture.py's call form
    assert open(fn, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    with h5lite.File(fn, "r") as f:
        assert sorted(f.keys()) == sorted(["f8", "f4", "i8", "i2", "names", "grp", "empty", "scalar", "growable", "positional"])
        assert "grp/sub/x" in f and "grp/sub" in f and "grp/nope" not in f and f.get("missing") is None
        for k, v in data.items():
            got = f[k][()]
            assert got.dtype == np.asarray(v).dtype and got.shape == np.shape(v)
            np.testing.assert_array_equal(got, v)
        assert f["f4"].shape == (300, 40) and f["f4"].dtype == np.float32 and len(f["f4"]) == 300
        np.testing.assert_array_equal(f["f8"][2:5, 1], data["f8"][2:5, 1])
        np.testing.assert_array_equal(f.get("growable").value, data["f4"])
        np.testing.assert_array_equal(f["positional"][()], data["f8"])
        with pytest.raises(KeyError):
            f["missing"]
    with h5lite.File(fn, "a") as f:                       # append mode keeps what is there
        f.create_dataset("later", data=np.ones(3))
    with h5lite.File(fn, "r") as f:
        assert "later" in f and "f8" in f


def test_checksum_detects_corruption(tmp_path):
    fn = str(tmp_path / "c.h5")
    a = np.arange(50000, dtype=np.float64)
    with h5lite.File(fn, "w") as f:
        f.create_dataset("x", data=a, fletcher32=True)     # chunked, checksum only: the bytes are findable
    raw = bytearray(open(fn, "rb").read())
    pos = bytes(raw).find(a[1000:1004].tobytes())
    assert pos > 0
    raw[pos + 3] ^= 0x40
    open(fn, "wb").write(bytes(raw))
    with h5lite.File(fn, "r") as f:
        with pytest.raises(IOError):
            f["x"][()]


def test_mixture_statserver_factor_analyser_h5_products(tmp_path):
    rs = np.random.RandomState(3)
    C, D = 6, 5
    m = sk.Mixture()
    m.w = rs.dirichlet(np.ones(C))
    m.mu = rs.randn(C, D)
    m.invcov = rs.uniform(0.5, 2.0, size=(C, D))
    m.cov_var_ctl = 1.0 / m.invcov
    m._compute_all()
    fn = str(tmp_path / "ubm.h5")
    m.write(fn)
    back = sk.Mixture()
    back.read(fn)
    for k in ("w", "mu", "invcov", "cov_var_ctl", "cst", "det", "A"):
        np.testing.assert_array_equal(getattr(back, k), getattr(m, k))
    assert back.validate()
    with h5lite.File(fn, "r") as f:                        # dataset names and types of mixture.py:334-358
        assert sorted(f.keys()) == sorted(["w", "mu", "invcov", "invchol", "cov_var_ctl", "cst", "det", "a"])
        assert all(f[k].dtype == np.float64 for k in f.keys())

    S = 9
    ss = sk.StatServer()
    ss.modelset = np.array(["m%d" % (i % 4) for i in range(S)])
    ss.segset = np.array(["seg-%02d" % i for i in range(S)])
    ss.start = np.empty(S, "|O")
    ss.stop = np.empty(S, "|O")
    ss.start[2], ss.stop[2] = 10, 250
    ss.stat0 = rs.uniform(0, 30, size=(S, C))
    ss.stat1 = rs.randn(S, C * D) * 7
    fs = str(tmp_path / "stats.h5")
    ss.write(fs)
    rd = sk.StatServer.read(fs)
    assert list(rd.modelset) == list(ss.modelset) and list(rd.segset) == list(ss.segset)
    assert rd.start[2] == 10 and rd.stop[2] == 250 and rd.start[0] is None and rd.stop[5] is None
    np.testing.assert_array_equal(rd.stat0, ss.stat0.astype(np.float32).astype(np.float64))     # the float32 storage contract
    np.testing.assert_array_equal(rd.stat1, ss.stat1.astype(np.float32).astype(np.float64))
    sub = sk.StatServer.read_subset(fs, np.array([7, 1, 4]))
    assert list(sub.segset) == ["seg-07", "seg-01", "seg-04"]
    np.testing.assert_array_equal(sub.stat1, rd.stat1[[7, 1, 4]])
    with h5lite.File(fs, "r") as f:
        assert f["stat0"].dtype == np.float32 and f["stat1"].dtype == np.float32 and f["modelset"].dtype.kind == "S"
    # mode 'a' on an existing file: the datasets grow, the new rows follow the old ones (statserver.py:450-476)
    more = sk.StatServer()
    more.modelset = np.array(["m9", "m-with-a-longer-name"])
    more.segset = np.array(["seg-90", "seg-91"])
    more.start = np.empty(2, "|O")
    more.stop = np.empty(2, "|O")
    more.stat0 = rs.uniform(0, 30, size=(2, C))
    more.stat1 = rs.randn(2, C * D)
    for ext in (".h5", ".npz"):
        fg = str(tmp_path / ("grow" + ext))
        ss.write(fg)
        more.write(fg, mode="a")
        both = sk.StatServer.read(fg)
        assert both.stat0.shape == (S + 2, C) and list(both.segset[-2:]) == ["seg-90", "seg-91"]
        assert both.modelset[S] == "m9" and both.modelset[S + 1].startswith("m-")      # fixed-length ids of the first write (HDF5)
        np.testing.assert_array_equal(both.stat1[:S], rd.stat1)
        np.testing.assert_array_equal(both.stat1[S:], more.stat1.astype(np.float32).astype(np.float64))
        assert both.start[2] == 10 and both.start[S] is None
    ss.write(fs, prefix="second/")                         # a second StatServer under a prefix (statserver.py:418-449)
    with h5lite.File(fs, "r") as f:
        assert "second/stat1" in f and "stat1" not in f    # mode 'w' starts a new file, as h5py does
    rd2 = sk.StatServer.read(fs, prefix="second/")
    np.testing.assert_array_equal(rd2.stat0, rd.stat0)

    fa = sk.FactorAnalyser()
    fa.mean = rs.randn(C * D)
    fa.F = rs.randn(C * D, 4)
    fa.Sigma = rs.uniform(0.5, 2.0, size=C * D)
    ff = str(tmp_path / "tv.h5")
    fa.write(ff)
    fb = sk.FactorAnalyser.read(ff)
    np.testing.assert_array_equal(fb.F, fa.F)
    np.testing.assert_array_equal(fb.mean, fa.mean)
    np.testing.assert_array_equal(fb.Sigma, fa.Sigma)
    assert fb.G is None and fb.H is None
    with h5lite.File(ff, "r") as f:                        # fa/* layout and the kind vector of factor_analyser.py:264-290
        assert f.keys() == ["fa"] and list(f["fa/kind"][()]) == [1, 1, 0, 0, 1]


def _idmap():
    m = sk.IdMap()
    m.set(np.array(["spk1", "spk1", "spk22"], dtype="|O"), np.array(["a/b", "c", "dd"], dtype="|O"),
          start=[None, 10, 0], stop=[None, 250, 99])
    return m


@pytest.mark.parametrize("ext", [".h5", ".npz"])
def test_idmap_ndx_scores_files(tmp_path, ext):
    """The bosaris containers keep the reference's HDF5 datasets (idmap.py:84-115, ndx.py:93-112,
    scores.py:94-115) and read back what they wrote."""
    m = _idmap()
    p = str(tmp_path / ("idmap" + ext))
    m.write(p)
    r = sk.IdMap.read(p)
    assert list(r.leftids) == list(m.leftids) and list(r.rightids) == list(m.rightids)
    assert list(r.start) == [None, 10, 0] and list(r.stop) == [None, 250, 99]
    r2 = sk.IdMap(p)
    assert list(r2.rightids) == list(m.rightids)

    ndx = sk.Ndx(models=np.array(["m1", "m2", "m1"]), testsegs=np.array(["s1", "s1", "s3"]))
    p = str(tmp_path / ("ndx" + ext))
    ndx.write(p)
    for back in (sk.Ndx.read(p), sk.Ndx(p)):
        assert list(back.modelset) == ["m1", "m2"] and list(back.segset) == ["s1", "s3"]
        assert np.array_equal(back.trialmask, [[True, True], [True, False]])

    sc = sk.Scores()
    sc.modelset, sc.segset = ndx.modelset, ndx.segset
    sc.scoremask = ndx.trialmask.copy()
    sc.scoremat = np.array([[1.5, -2.25], [0.125, 0.0]])
    p = str(tmp_path / ("scores" + ext))
    sc.write(p)
    back = sk.Scores.read(p)
    assert np.array_equal(back.scoremat, sc.scoremat) and np.array_equal(back.scoremask, sc.scoremask)
    assert list(back.modelset) == ["m1", "m2"]
    if ext == ".h5":
        with sk.h5lite.File(str(tmp_path / "idmap.h5"), "r") as f:
            assert f["start"][()].dtype == np.int32 and f["leftids"][()].dtype.kind == "S"
        with sk.h5lite.File(str(tmp_path / "ndx.h5"), "r") as f:
            assert f["trial_mask"][()].dtype == np.int8
        pm = str(tmp_path / "scores_matlab.h5")
        sc.write_matlab(pm)
        back = sk.Scores.read_matlab(pm)
        assert np.array_equal(back.scoremat, sc.scoremat) and list(back.segset) == ["s1", "s3"]


def test_bosaris_text_files(tmp_path):
    m = _idmap()
    p = str(tmp_path / "idmap.txt")
    m.write_txt(p)
    r = sk.IdMap.read_txt(p)
    assert list(r.leftids) == list(m.leftids) and list(r.start) == [None, 10, 0] and list(r.stop) == [None, 250, 99]
    with open(str(tmp_path / "two.txt"), "w") as fh:
        fh.write("spk1 a\nspk2 b\n")
    r = sk.IdMap.read_txt(str(tmp_path / "two.txt"))
    assert list(r.rightids) == ["a", "b"] and list(r.start) == [None, None]

    ndx = sk.Ndx(models=np.array(["m2", "m1", "m1"]), testsegs=np.array(["s1", "s1", "s3"]))
    p = str(tmp_path / "ndx.txt")
    ndx.save_txt(p)
    back = sk.Ndx.read_txt(p)
    assert list(back.modelset) == ["m1", "m2"] and np.array_equal(back.trialmask, ndx.trialmask)

    sc = sk.Scores()
    sc.modelset, sc.segset, sc.scoremask = ndx.modelset, ndx.segset, ndx.trialmask.copy()
    sc.scoremat = np.array([[1.5, -2.25], [0.125, 7.0]])
    p = str(tmp_path / "scores.txt")
    sc.write_txt(p)
    back = sk.Scores.read_txt(p)
    assert np.array_equal(back.scoremask, sc.scoremask)
    assert np.array_equal(back.scoremat, np.where(sc.scoremask, sc.scoremat, 0.0))

    class Key:
        modelset, segset = np.array(["m2", "m1"]), np.array(["s3", "s1"])
        tar = np.array([[False, True], [True, False]])
        non = np.array([[True, False], [False, True]])
    tar, non = back.get_tar_non(Key)
    assert sorted(tar) == [-2.25, 0.125] and list(non) == [1.5]     # (m2, s3) was never scored


@pytest.mark.parametrize("ext", [".h5", ".npz", ".txt"])
def test_key_files_and_score_split(tmp_path, ext):
    """The BOSARIS Key (key.py:49-284): built from trial lists, written / read in the reference's datasets, turned
    into an Ndx, and used to split a Scores matrix into target and non-target scores."""
    models = np.array(["m2", "m1", "m1", "m2", "m1"])
    segs = np.array(["s1", "s1", "s3", "s3", "s2"])
    trials = np.array(["nontarget", "target", "nontarget", "target", "target"])
    key = sk.Key(models=models, testsegs=segs, trials=trials)
    assert list(key.modelset) == ["m1", "m2"] and list(key.segset) == ["s1", "s2", "s3"]
    assert np.array_equal(key.tar, [[True, True, False], [False, False, True]])
    assert np.array_equal(key.non, [[False, False, True], [True, False, False]])
    p = str(tmp_path / ("key" + ext))
    if ext == ".txt":
        key.write_txt(p)
        back = sk.Key.read_txt(p)
    else:
        key.write(p)
        back = sk.Key.read(p)
        assert np.array_equal(sk.Key(p).tar, key.tar)
    assert list(back.modelset) == ["m1", "m2"] and np.array_equal(back.tar, key.tar) and np.array_equal(back.non, key.non)
    if ext == ".h5":
        with sk.h5lite.File(p, "r") as f:
            assert np.array_equal(f["trial_mask"][()], [[1, 1, -1], [-1, 0, 1]]) and f["trial_mask"][()].dtype == np.int8
    ndx = key.to_ndx()
    assert np.array_equal(ndx.trialmask, [[True, True, True], [True, False, True]])
    sub = key.filter(["m1"], ["s1", "s3"], keep=True)
    assert list(sub.modelset) == ["m1"] and np.array_equal(sub.tar, [[True, False]]) and np.array_equal(sub.non, [[False, True]])
    sc = sk.Scores()
    sc.modelset, sc.segset = key.modelset, key.segset
    sc.scoremask = ndx.trialmask.copy()
    sc.scoremat = np.array([[3.0, 2.0, -1.0], [-2.0, 9.0, 4.0]])
    tar, non = sc.get_tar_non(key)
    assert sorted(tar) == [2.0, 3.0, 4.0] and sorted(non) == [-2.0, -1.0]


# ---- layout of the file products, checked by an INDEPENDENT reader: the HDF5 command-line tool h5dump ----
def _h5dump():
    import shutil
    for c in (os.environ.get("SKM_H5DUMP"), shutil.which("h5dump"), "/opt/conda/bin/h5dump", "/usr/bin/h5dump"):
        if c and os.path.exists(c):
            return c
    return None


def _header(fn):
    """{dataset path: text of its h5dump -H -p block}"""
    import re
    import subprocess
    tool = _h5dump()
    if tool is None:
        pytest.skip("h5dump not found")
    txt = subprocess.run([tool, "-H", "-p", fn], check=True, capture_output=True, text=True).stdout
    out, stack, cur = {}, [], None
    for line in txt.splitlines():
        m = re.match(r'\s*(GROUP|DATASET) "([^"]+)" \{', line)
        if m:
            stack.append((m.group(1), m.group(2), line.count("{") - line.count("}")))
            if m.group(1) == "DATASET":
                cur = "/".join(n for k, n, _ in stack if n != "/")
                out[cur] = ""
            continue
        if cur is not None:
            out[cur] += line + "\n"
        if stack:
            k, n, depth = stack[-1]
            depth += line.count("{") - line.count("}")
            stack[-1] = (k, n, depth)
            if depth <= 0:
                if k == "DATASET":
                    cur = None
                stack.pop()
    return out


def _expect(block, dtype, dims, maxdims, filtered=True):
    import re
    assert dtype in block, (dtype, block)
    m = re.search(r"DATASPACE\s+SIMPLE \{ \(([^)]*)\) / \(([^)]*)\) \}", block)
    assert m, block
    assert [s.strip() for s in m.group(1).split(",")] == [str(d) for d in dims], block
    assert [s.strip() for s in m.group(2).split(",")] == [str(d) for d in maxdims], block
    if filtered:
        assert "CHUNKED" in block and "CHECKSUM FLETCHER32" in block and "COMPRESSION DEFLATE { LEVEL 4 }" in block, block


def test_layout_of_file_products_as_h5dump_sees_them(tmp_path):
    """What the reference's own h5py calls produce (statserver.py:426-455, mixture.py:332-358,
    factor_analyser.py:270-296, bosaris/idmap.py:90-115): dataset names, storage types, extents and maximum extents,
    gzip level 4 (h5py's default) + fletcher32 on chunked datasets -- read back by the HDF5 tools, not by h5lite."""
    rs = np.random.RandomState(5)
    C, D, S, R = 4, 3, 6, 2
    # StatServer
    ss = sk.StatServer()
    ss.modelset = np.array(["spk%d" % (i % 2) for i in range(S)])
    ss.segset = np.array(["seg-%03d" % i for i in range(S)])
    ss.start = np.empty(S, "|O")
    ss.stop = np.empty(S, "|O")
    ss.stat0 = rs.rand(S, C)
    ss.stat1 = rs.randn(S, C * D)
    fn = str(tmp_path / "stat.h5")
    ss.write(fn)
    h = _header(fn)
    assert sorted(h) == ["modelset", "segset", "start", "stat0", "stat1", "stop"]
    _expect(h["stat0"], "H5T_IEEE_F32LE", (S, C), ("H5S_UNLIMITED", C))
    _expect(h["stat1"], "H5T_IEEE_F32LE", (S, C * D), ("H5S_UNLIMITED", C * D))
    for k in ("modelset", "segset"):
        _expect(h[k], "H5T_STRING", (S,), ("H5S_UNLIMITED",))
        assert "STRPAD H5T_STR_NULLPAD" in h[k] and "CSET H5T_CSET_ASCII" in h[k]
    assert "STRSIZE 4;" in h["modelset"] and "STRSIZE 7;" in h["segset"]
    for k in ("start", "stop"):                      # all None -> -1 in the reference's int8 (statserver.py:418-424)
        _expect(h[k], "H5T_STD_I8LE", (S,), ("H5S_UNLIMITED",))
    ss.start[:] = 300                                # beyond int8: widened instead of wrapped
    ss.stop[:] = 900
    ss.write(fn)
    _expect(_header(fn)["start"], "H5T_STD_I32LE", (S,), ("H5S_UNLIMITED",))
    # Mixture
    m = sk.Mixture()
    m.w = np.full(C, 1.0 / C)
    m.mu = rs.randn(C, D)
    m.invcov = rs.uniform(0.5, 2.0, (C, D))
    m.cov_var_ctl = 1.0 / m.invcov
    m._compute_all()
    fn = str(tmp_path / "ubm.h5")
    m.write(fn)
    h = _header(fn)
    assert sorted(h) == sorted(["w", "mu", "invcov", "invchol", "cov_var_ctl", "cst", "det", "a"])
    _expect(h["w"], "H5T_IEEE_F64LE", (C,), (C,))
    _expect(h["mu"], "H5T_IEEE_F64LE", (C, D), (C, D))
    _expect(h["invcov"], "H5T_IEEE_F64LE", (C, D), (C, D))
    _expect(h["a"], "H5T_IEEE_F64LE", (C,), (C,))
    # FactorAnalyser
    fa = sk.FactorAnalyser()
    fa.mean = rs.randn(C * D)
    fa.F = rs.randn(C * D, R)
    fa.Sigma = rs.rand(C * D)
    fn = str(tmp_path / "tv.h5")
    fa.write(fn)
    h = _header(fn)
    assert sorted(h) == ["fa/f", "fa/kind", "fa/mean", "fa/sigma"]
    _expect(h["fa/f"], "H5T_IEEE_F64LE", (C * D, R), (C * D, R))
    _expect(h["fa/kind"], "H5T_STD_I16LE", (5,), (5,))
    # IdMap
    im = sk.IdMap()
    im.leftids = ss.modelset
    im.rightids = ss.segset
    im.start = np.empty(S, "|O")
    im.stop = np.empty(S, "|O")
    fn = str(tmp_path / "idmap.h5")
    im.write(fn)
    h = _header(fn)
    assert sorted(h) == ["leftids", "rightids", "start", "stop"]
    _expect(h["leftids"], "H5T_STRING", (S,), ("H5S_UNLIMITED",))
    _expect(h["start"], "H5T_STD_I32LE", (S,), ("H5S_UNLIMITED",))
    _expect(h["stop"], "H5T_STD_I32LE", (S,), ("H5S_UNLIMITED",))


def test_growing_a_file_that_started_empty(tmp_path):
    """create_dataset(maxshape=...) on zero rows must still be chunked, or StatServer.write(mode='a') cannot resize."""
    fn = str(tmp_path / "grow.h5")
    with h5lite.File(fn, "w") as f:
        f.create_dataset("x", data=np.zeros((0, 3), dtype=np.float32), maxshape=(None, 3), compression="gzip",
                         fletcher32=True)
    with h5lite.File(fn, "a") as f:
        f["x"].resize((2, 3))
        f["x"][0:] = np.arange(6, dtype=np.float32).reshape(2, 3)
    with h5lite.File(fn, "r") as f:
        np.testing.assert_array_equal(f["x"][()], np.arange(6, dtype=np.float32).reshape(2, 3))
