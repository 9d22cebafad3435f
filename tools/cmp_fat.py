import sys, os, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch, sidekit_amd as sk
    from oracle import sidekit_ref as orc
    C, D, T = 512, 60, 20000
    gmm = orc.random_gmm(C, D, seed=5); cep = orc.synth_frames(T, D, seed=6)
    eng = sk.default_engine()
    out = eng.to_host(eng.estep(eng.upload_frames(cep), eng.pack(gmm.w, gmm.mu, gmm.invcov)))
    np.save(sys.argv[1], out)
else:
    for v in ("0", "1"):
        subprocess.check_call([sys.executable, __file__, "/tmp/o%s.npy" % v], env=dict(os.environ, SKM_FAT=v))
    a, b = np.load("/tmp/o0.npy"), np.load("/tmp/o1.npy")
    C, D = 512, 60
    N0, N1 = a[:C], b[:C]
    print("llk", a[-1], b[-1])
    print("N sum", N0.sum(), N1.sum())
    bad = np.where(np.abs(N0 - N1) > 1e-4 * np.abs(N0).max())[0]
    print("bad comps", len(bad), bad[:40])
    F0, F1 = a[C:C + C * D].reshape(C, D), b[C:C + C * D].reshape(C, D)
    S0, S1 = a[C + C * D:-1].reshape(C, D), b[C + C * D:-1].reshape(C, D)
    for nm, x, y in (("F", F0, F1), ("S", S0, S1)):
        d = np.abs(x - y) > 1e-4 * np.abs(x).max()
        print(nm, "bad entries", d.sum(), "comps", np.unique(np.where(d)[0])[:20], "cols", np.unique(np.where(d)[1])[:20])
    print(N0[:6], N1[:6])
