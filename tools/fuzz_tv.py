"""Random shapes through the total-variability entry points against a float64 torch restatement on the device:
    python tools/fuzz_tv.py [cases] [seed]
Every case draws (C, D, R, sessions, batch) and compares skm_tv_whiten / skm_tv_prepare / skm_tv_estep (training
accumulators, i-vectors with and without uncertainty) and skm_tv_mstep with the same quantities computed in float64
from the same float32 statistics (sidekit/factor_analyser.py:52-92, 604-665 restated).  Bar: 1e-4 of the largest entry,
cosine 0.9999 for i-vectors.  A diagnostic, not part of the test suite."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sidekit_amd as sk

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(SEED)
eng = sk.default_engine()
dev = eng.device
BAR = 1e-4


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def upper(A):
    R = A.shape[-1]
    iu = torch.triu_indices(R, R, device=A.device)
    return A[..., iu[0], iu[1]]


CS = [1, 2, 5, 16, 24, 33, 40, 64, 100, 128]
DS = [1, 3, 8, 12, 13, 20, 39, 60]
RS = [1, 2, 5, 6, 31, 32, 33, 50, 64, 100, 128, 150, 200, 256, 400]
SS = [1, 2, 7, 31, 33, 64, 100, 257, 300]
BS = [1, 5, 32, 100, 1024]
bad = 0
worst = 0.0
for case in range(CASES):
    while True:
        C = int(rng.choice(CS)); D = int(rng.choice(DS)); R = int(rng.choice(RS))
        if R <= C * D:
            break
    S = int(rng.choice(SS)); batch = int(rng.choice(BS))
    idx = torch.arange(C, device=dev).repeat_interleave(D)
    mu = rng.standard_normal(C * D); invcov = 1.0 / (0.5 + rng.random(C * D))
    # statistics of sessions of 100..3000 frames: N_c >= 0 sums to the frame count, F_c = N_c (mu_c + offset) + noise
    frames = rng.integers(100, 3000, S)
    post = rng.dirichlet(np.full(C, 0.3), S) * frames[:, None]
    s0 = post.astype(np.float32)
    off = rng.standard_normal((S, C * D)) * 0.3
    s1 = (np.repeat(post, D, axis=1) * (mu[None, :] + off)
          + np.sqrt(np.repeat(post, D, axis=1) / invcov[None, :]) * rng.standard_normal((S, C * D))).astype(np.float32)
    Fm = 0.1 * rng.standard_normal((C * D, R))
    s0d = torch.from_numpy(s0).to(dev); s1d = torch.from_numpy(s1).to(dev)
    mud = torch.from_numpy(mu).to(dev); sq = torch.from_numpy(np.sqrt(invcov)).to(dev)
    Fd = torch.from_numpy(Fm).to(dev)
    # float64 restatement (the two in-place whitening updates each round to float32, as the reference's do)
    f = (s1d.double() - s0d.double()[:, idx] * mud).float()
    f = (f.double() * sq).float().double()
    G = torch.einsum('cdr,cdk->crk', Fd.reshape(C, D, R), Fd.reshape(C, D, R))
    L = torch.eye(R, dtype=torch.float64, device=dev)[None] + torch.einsum('sc,crk->srk', s0d.double(), G)
    try:
        inv = torch.linalg.inv(L)
    except RuntimeError:        # hipBLAS' batched LU ran out of its own workspace (S x R x R large): a few sessions at a time
        inv = torch.cat([torch.linalg.inv(L[i:i + 8]) for i in range(0, S, 8)])
    e = torch.einsum('sr,srk->sk', f @ Fd, inv)
    ehh = inv + e[:, :, None] * e[:, None, :]
    A_ref = torch.cat([torch.einsum('sc,srk->crk', s0d.double(), ehh), ehh.sum(0)[None]])
    C_ref = e.T @ f
    sig_ref = torch.diagonal(ehh, dim1=1, dim2=2)
    try:
        ft, naug = eng.tv_whiten(s0d, s1d, mu, invcov)
        T32, Gd = eng.tv_prepare(Fd, C, D)
        A_aug = torch.zeros((C + 1, R * R), dtype=torch.float64, device=dev)
        Cacc = torch.zeros((R, C * D), dtype=torch.float64, device=dev)
        e_h, sig = eng.tv_estep(naug, ft, T32, Gd, C, D, batch=batch, A_aug=A_aug, Cacc=Cacc, uncertainty=True)
        e_s, _ = eng.tv_estep(naug, ft, T32, Gd, C, D, batch=batch, uncertainty=False)
        A_keep = A_aug.clone()
        # the planes routes: training with f~ split once per run, extraction from the planes the whitening writes
        planes = eng.tv_ftilde_planes(ft, C, D)
        A_p = torch.zeros_like(A_aug)
        C_p = torch.zeros_like(Cacc)
        e_p, sig_p = eng.tv_estep(naug, ft, T32, Gd, C, D, batch=batch, A_aug=A_p, Cacc=C_p, uncertainty=True, ft_planes=planes)
        rows, naug_r = eng.tv_whiten_planes(s0d, s1d, mu, invcov)
        e_r, _ = eng.tv_estep(naug_r, None, T32, Gd, C, D, batch=batch, uncertainty=False, ft_planes=rows)
        Fn = eng.tv_mstep(A_aug, Cacc, C, D, R)
        torch.cuda.synchronize()
    except Exception as ex:
        print("case %3d C=%3d D=%2d R=%3d S=%3d batch=%4d  rejected: %s" % (case, C, D, R, S, batch, ex), flush=True)
        continue
    # M-step from the reference accumulators
    Fr = torch.empty_like(Fd)
    for c in range(C):
        Fr[c * D:(c + 1) * D] = torch.linalg.solve(A_ref[c], C_ref[:, c * D:(c + 1) * D]).T
    cosv = torch.nn.functional.cosine_similarity(e_h, e, dim=1).min().item()
    errs = {"ft": rel(ft.double(), f), "A": rel(upper(A_keep[:C].reshape(C, R, R)), upper(A_ref[:C])),
            "R": rel(upper(A_keep[C].reshape(R, R)), upper(A_ref[C])), "C": rel(Cacc, C_ref),
            "e": rel(e_h, e), "sig": rel(sig, sig_ref), "e_solve": rel(e_s, e), "F": rel(Fn, Fr), "1-cos": 1.0 - cosv,
            "A_pl": rel(upper(A_p[:C].reshape(C, R, R)), upper(A_ref[:C])), "C_pl": rel(C_p, C_ref), "e_pl": rel(e_p, e),
            "e_rows": rel(e_r, e)}
    cond = torch.linalg.cond(L).max().item()
    # fewer sessions than factors: A_c is nearly singular and the M-step solve magnifies the 1e-6 of its inputs; shown,
    # not judged
    m = max(v for k, v in errs.items() if not (k == "F" and S < R))
    worst = max(worst, m)
    flag = "" if m <= BAR and all(math.isfinite(v) for v in errs.values()) else "  <-- over the bar"
    if flag:
        bad += 1
    print("case %3d C=%3d D=%2d R=%3d S=%3d batch=%4d cond(L)=%.0e  " % (case, C, D, R, S, batch, cond)
          + " ".join("%s=%.1e" % kv for kv in errs.items()) + flag, flush=True)
print("cases %d, over the bar %d, worst %.2e" % (CASES, bad, worst))
sys.exit(1 if bad else 0)
