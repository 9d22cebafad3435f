"""Opt-in split-precision E-step (SKM_SPLIT=bf16x3, csrc/skm_gmm_b3.hip): the fat kernel's two GEMMs as bf16 x 3
products on the bf16 matrix cores.  Same bars as the float32-MFMA kernel it stands in for (tests/test_gpu_gmm.py), against
the oracle, and against that kernel on the same frames; the drop-in EM loop with the switch on."""
import os

import numpy as np
import pytest

from oracle import sidekit_ref as orc
from conftest import golden, rel, rel_rows, rel_rownorm, check_first_order
from test_gpu_gmm import unpack

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def sk():
    import sidekit_amd
    return sidekit_amd


@pytest.fixture(scope="module")
def eng(sk):
    return sk.default_engine()


@pytest.fixture()
def split():
    os.environ["SKM_SPLIT"] = "bf16x3"
    try:
        yield
    finally:
        os.environ.pop("SKM_SPLIT", None)


def _which(eng, X, packed):
    eng.profile(True)
    eng.estep(X, packed)
    eng.torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    return prof


@pytest.mark.parametrize("C,D,T,flush", [(512, 60, 20000, None), (512, 60, 95, None), (512, 60, 1, None), (512, 32, 7001, None),
                                         (512, 44, 4097, None), (500, 60, 9999, None),      # model ends inside the last tile
                                         (481, 36, 3000, None),
                                         (512, 60, 300000, 3)])                              # several folds per wave
def test_split_estep_vs_oracle(eng, split, C, D, T, flush):
    gmm = orc.random_gmm(C, D, seed=C + D)
    cep = orc.synth_frames(T, dim=D, clusters=16, seed=T)
    X = eng.upload_frames(cep)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    if flush:
        os.environ["SKM_FLUSH"] = str(flush)
    try:
        out = eng.to_host(eng.estep(X, packed))
        assert "k_estep_b3" in _which(eng, X, packed)
        # the same launch twice is bit-identical (no float atomics)
        assert np.array_equal(out, eng.to_host(eng.estep(X, packed)))
    finally:
        os.environ.pop("SKM_FLUSH", None)
    N, F, S, llk = unpack(out, C, D)
    rN, rF, rS, rl = orc.expectation(gmm, cep.astype(np.float64))
    assert rel(N, rN) < TOL and rel(F, rF) < TOL and rel(S, rS) < TOL
    rows = (rel_rows(N, rN), rel_rows(S, rS), rel_rownorm(S, rS))
    assert max(rows) < TOL, rows
    check_first_order(F, rF)
    assert abs(llk - rl) < 1e-5 * abs(rl)
    assert abs(N.sum() - T) < 1e-4 * T
    # against the float32-MFMA kernel on the same frames: two float32-grade paths
    os.environ.pop("SKM_SPLIT")
    try:
        ref = eng.to_host(eng.estep(X, packed))
        assert "k_estep_fat" in _which(eng, X, packed)
    finally:
        os.environ["SKM_SPLIT"] = "bf16x3"
    assert rel(out[:C], ref[:C]) < 5e-5 and rel(out[C:-1], ref[C:-1]) < 5e-5


@pytest.mark.parametrize("C,D,T", [(1024, 60, 9999), (2048, 60, 6000), (2048, 36, 95), (2000, 60, 7000), (4096, 60, 4100),
                                   (1024, 44, 1), (2048, 60, 300000)])
def test_split_two_sweeps_vs_oracle(eng, split, C, D, T):
    """Above 512 components the split route takes two sweeps over blocks of 512 (the second repeats the first one's logits
    bit for bit, so the posteriors of a frame sum to one to float32 rounding of one log-sum-exp)."""
    gmm = orc.random_gmm(C, D, seed=C + D)
    cep = orc.synth_frames(T, dim=D, clusters=16, seed=T)
    X = eng.upload_frames(cep)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    out = eng.to_host(eng.estep(X, packed))
    prof = _which(eng, X, packed)
    assert "k_b3_sweep1" in prof and "k_b3_sweep2" in prof, prof
    assert np.array_equal(out, eng.to_host(eng.estep(X, packed)))
    N, F, S, llk = unpack(out, C, D)
    rN, rF, rS, rl = orc.expectation(gmm, cep.astype(np.float64))
    assert rel(N, rN) < TOL and rel(F, rF) < TOL and rel(S, rS) < TOL
    rows = (rel_rows(N, rN), rel_rows(S, rS), rel_rownorm(S, rS))
    assert max(rows) < TOL, rows
    check_first_order(F, rF)
    assert abs(llk - rl) < 1e-5 * abs(rl)
    assert abs(N.sum() - T) < 2e-6 * T + 1e-3          # posteriors sum to one per frame, well below the logit's own error
    acc = eng.estep(X, packed, out=eng.estep(X, packed), accumulate=True)
    assert rel(eng.to_host(acc), 2.0 * out) < 1e-12


def test_split_estep_vs_reference_golden(eng, split):
    """The reference's own numbers (tests/golden/estep_c64_d60.npz holds 64 components: padded here to the 512 the split
    kernel is built for by repeating the model 8 times with an eighth of the weight -- N, F, S of a copy are an eighth)."""
    g = golden("estep_c64_d60.npz")
    rep = 8
    w = np.tile(g["w"] / rep, rep)
    mu = np.tile(g["mu"], (rep, 1))
    ic = np.tile(g["invcov"], (rep, 1))
    X = eng.upload_frames(g["cep"])
    packed = eng.pack_stats(w, mu, ic)
    out = eng.to_host(eng.estep(X, packed))
    assert "k_estep_b3" in _which(eng, X, packed)
    C, D = 512, 60
    N, F, S, llk = unpack(out, C, D)
    N = N.reshape(rep, 64).sum(0)
    F = F.reshape(rep, 64, D).sum(0)
    S = S.reshape(rep, 64, D).sum(0)
    assert rel_rows(N, g["N"]) < TOL and rel_rows(S, g["S"]) < TOL and rel_rownorm(S, g["S"]) < TOL
    check_first_order(F, g["F"])
    assert abs(llk - float(g["llk"])) < 1e-5 * abs(float(g["llk"]))


def test_split_raw_scale_features(eng, split):
    """Un-normalised cepstra (offsets of 10..50, variances of 10..100 per column) through the centred, reference-Gaussian
    packing on the split kernel."""
    C, D, T = 512, 60, 30000
    gmm = orc.raw_gmm(C, D, seed=5)
    cep = orc.synth_raw_frames(T, D, seed=6)
    X = eng.upload_frames(cep)
    packed = eng.pack_stats(gmm.w, gmm.mu, gmm.invcov)
    assert packed.shift is not None and packed.ref is not None
    out = eng.to_host(eng.estep(X, packed))
    assert "k_estep_b3" in _which(eng, X, packed)
    N, F, S, llk = unpack(out, C, D)
    rN, rF, rS, rl = orc.expectation(gmm, cep.astype(np.float64))
    assert rel_rows(N, rN) < TOL and rel_rows(S, rS) < TOL and rel_rownorm(F, rF) < TOL
    assert abs(llk - rl) < 1e-5 * abs(rl)


def test_split_em_split_dropin(sk, split):
    """Mixture.EM_split to 512 components with the switch on: the last stage runs on the split kernel; against the same
    call on the float32-MFMA kernels (EM is chaotic over 37 iterations: compared where it is not, the log-likelihoods)."""
    rs = np.random.RandomState(3)
    frames = orc.synth_frames(60000, dim=40, clusters=24, seed=9).astype(np.float32)
    fs = sk.FeaturesServer(features={"a": frames[:30000], "b": frames[30000:]})
    u1 = sk.Mixture()
    l1 = u1.EM_split(features_server=fs, feature_list=["a", "b"], distrib_nb=512, iterations=(1, 2, 2, 2, 2, 2, 2, 2, 3))
    os.environ.pop("SKM_SPLIT")
    try:
        u0 = sk.Mixture()
        l0 = u0.EM_split(features_server=fs, feature_list=["a", "b"], distrib_nb=512, iterations=(1, 2, 2, 2, 2, 2, 2, 2, 3))
    finally:
        os.environ["SKM_SPLIT"] = "bf16x3"
    assert u1.w.shape == (512,) and np.isfinite(u1.mu).all() and abs(u1.w.sum() - 1.0) < 1e-9
    assert np.allclose(l1[:-3], l0[:-3], rtol=0, atol=0)          # the stages below 512 components are the same kernels
    assert np.abs(np.array(l1[-3:]) - np.array(l0[-3:])).max() < 1e-4 * abs(l0[-1])
