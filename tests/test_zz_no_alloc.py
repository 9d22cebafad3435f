"""skm_gmm_estep_noalloc (the entry point that allocates nothing: its scratch is a file-scope device array of
the library) against skm_gmm_estep, through the C ABI: same kernels, so the results must be bit-identical; two calls
in a row, the second with a smaller input, on every kernel family; and against the oracle at the north star's bar."""
import ctypes

import numpy as np
import pytest

from oracle import sidekit_ref as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import sidekit_amd
    return sidekit_amd.default_engine()


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("C,D,T1,T2", [(512, 60, 50000, 6001),       # single-sweep one-wave-per-SIMD kernel
                                       (2048, 60, 40000, 3000),      # team kernel (cooperative launch)
                                       (256, 20, 9000, 500),         # fused two-wave kernel
                                       (128, 60, 20000, 999),        # one-wave kernel, four component tiles
                                       (64, 13, 3000, 777),          # one-wave kernel, two tiles, scalar staging
                                       (20, 24, 5000, 31),           # one-wave kernel, one tile
                                       (64, 132, 9000, 500),         # wide rows: one component tile per wave, two waves
                                       (160, 13, 3000, 777),         # five component tiles run as eight (padding)
                                       (1000, 60, 9000, 500),        # team kernel, last wave on padding tiles
                                       (600, 13, 3000, 777)])        # two sweeps: per-frame partials in the scratch
def test_noalloc_equals_workspace_entry_point(eng, C, D, T1, T2):
    torch = eng.torch
    gmm = orc.random_gmm(C, D, seed=C + D)
    packed = eng.pack(gmm.w, gmm.mu, gmm.invcov)
    n = 2 * C * D + C + 1
    for T in (T1, T2):                                               # the second call is smaller: stale scratch contents
        cep = orc.synth_frames(T, dim=D, clusters=16, seed=T)
        X = eng.upload_frames(cep)
        want = eng.to_host(eng.estep(X, packed))
        out = torch.full((n,), float("nan"), dtype=torch.float64, device=eng.device)
        before = torch.cuda.memory_allocated(eng.device)
        eng.estep_noalloc(X, packed, out)
        assert torch.cuda.memory_allocated(eng.device) == before
        got = eng.to_host(out)
        np.testing.assert_array_equal(got, want)
        eng.estep_noalloc(X, packed, out, accumulate=True)           # accumulate adds to what is there
        np.testing.assert_allclose(eng.to_host(out), 2.0 * want, rtol=1e-13)
        rN, rF, rS, rl = orc.expectation(gmm, cep.astype(np.float64))
        ref = [rN, rF.ravel(), rS.ravel()]
        parts = [got[:C], got[C:C + C * D], got[C + C * D:C + 2 * C * D]]
        for a, b in zip(parts, ref):
            assert _rel(a, b) < 1e-4
        assert abs(got[-1] - rl) < 1e-5 * abs(rl)


def test_noalloc_rejects_what_does_not_fit(eng):
    """Two-sweep shapes keep two floats per (component block, frame) in the scratch: beyond its 144 MiB the call
    is refused with SKM_ERR_WORKSPACE before anything is launched."""
    import sidekit_amd
    lib = eng.lib
    assert lib.skm_gmm_estep_noalloc_max_bytes() == 144 << 20
    C, D, T = 600, 13, 10_000_000                                    # 19 component tiles: no single-sweep kernel
    gmm = orc.random_gmm(C, D, seed=1)
    packed = eng.pack(gmm.w, gmm.mu, gmm.invcov)
    # X really holds T frames: a library that failed to refuse the call would run it, not fault the card
    X = eng.torch.zeros((T, D), dtype=eng.torch.float32, device=eng.device)
    out = eng.torch.zeros(2 * C * D + C + 1, dtype=eng.torch.float64, device=eng.device)
    rc = lib.skm_gmm_estep_noalloc(eng.ctx, ctypes.c_void_p(X.data_ptr()), T, C, D,
                                   ctypes.c_void_p(packed.packed.data_ptr()), ctypes.c_void_p(out.data_ptr()), 0, None)
    assert rc == -3
    assert b"scratch" in lib.skm_last_error()
    assert float(out.abs().sum().item()) == 0.0
    with pytest.raises(sidekit_amd.SkmError):          # the Python binding turns the code into an exception
        eng.estep_noalloc(X, packed, out)
