"""Random shapes through the GMM entry points against a float64 torch restatement on the device:
    python tools/fuzz_gmm.py [cases] [seed] [harsh|cmvn] [big]
"harsh" (default): frames 0.3 + 1.5 N(0,1), means 1.2 N(0,1), variances in 0.4..1.9, so a logit of a 159-column row
is about -300 and its float32 accumulation alone is worth 1e-4 of a posterior; "cmvn": unit-variance frames, means
0.5 N(0,1), variances in 0.5..1.5 (what mean / variance normalised features and a trained model look like).
Every case draws (C, D), a ragged set of sessions (empty ones, single frames, lengths that are no multiple of the
32-frame tile) and compares skm_gmm_estep, skm_gmm_estep_noalloc (where the shape fits it), skm_gmm_loglik, skm_gmm_logprob and skm_bw_stats with
N / F / S / llk computed in float64 from the same float32 frames.  Bar: 1e-4 of the largest entry (tests/ use the
same figure against the oracle).  A diagnostic, not part of the test suite."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sidekit_amd as sk

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 120
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 7
REGIME = sys.argv[3] if len(sys.argv) > 3 else "harsh"
BIG = len(sys.argv) > 4 and sys.argv[4] == "big"      # 0.1 ... 1.5 M frames in up to 3000 sessions (many workgroups, long streams)
XS, XM, MS, V0, V1 = (1.5, 0.3, 1.2, 0.4, 1.5) if REGIME == "harsh" else (1.0, 0.0, 0.5, 0.5, 1.0)
rng = np.random.default_rng(SEED)
eng = sk.default_engine()
dev = eng.device
BAR = 1e-4


def reference(X, w, mu, invcov, off):
    X64 = X.double()
    w = torch.from_numpy(w).to(dev); mu = torch.from_numpy(mu).to(dev); ic = torch.from_numpy(invcov).to(dev)
    C, D = mu.shape
    cst = torch.log(w) + 0.5 * torch.log(ic).sum(1) - 0.5 * D * math.log(2 * math.pi) - 0.5 * (mu * mu * ic).sum(1)
    lp = cst[None, :] + X64 @ (mu * ic).T - 0.5 * (X64 * X64) @ ic.T
    lse = torch.logsumexp(lp, 1)
    post = torch.exp(lp - lse[:, None])
    N = post.sum(0); F = post.T @ X64; S = post.T @ (X64 * X64)
    em = torch.cat([N, F.reshape(-1), S.reshape(-1), lse.sum()[None]])
    ns = len(off) - 1
    s0 = torch.zeros((ns, C), dtype=torch.float64, device=dev); s1 = torch.zeros((ns, C * D), dtype=torch.float64, device=dev)
    for s in range(ns):
        a, b = int(off[s]), int(off[s + 1])
        if b > a:
            s0[s] = post[a:b].sum(0)
            s1[s] = (post[a:b].T @ X64[a:b]).reshape(-1)
    return em, s0, s1, lse, lp


def rel(a, b):
    d = (a - b).abs().max().item()
    return d / max(b.abs().max().item(), 1e-30)


def draw_sessions():
    kind = rng.integers(0, 5)
    ns = int(rng.integers(1, 40))
    if BIG:
        total = int(rng.integers(100_000, 1_500_000))
        mean = [40, 300, 2000, 20000][rng.integers(0, 4)]
        ns = max(1, min(3000, total // mean))
        lens = np.maximum(0, rng.normal(total / ns, 0.4 * total / ns, ns)).astype(np.int64)
        lens[rng.random(ns) < 0.05] = 0
        if lens.sum() == 0:
            lens[0] = 5
        return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    if kind == 0:
        lens = rng.integers(0, 70, ns)
    elif kind == 1:
        lens = rng.integers(1, 700, ns)
    elif kind == 2:
        lens = np.where(rng.random(ns) < 0.3, 0, rng.integers(1, 300, ns))
    elif kind == 3:
        lens = rng.integers(1, 4, ns) * 32
    else:
        lens = rng.integers(500, 6000, max(1, ns // 6))
    if lens.sum() == 0:
        lens[0] = 5
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


CS = [1, 2, 3, 7, 16, 31, 32, 33, 40, 64, 65, 96, 100, 128, 129, 160, 200, 256, 257, 300, 384, 385, 500, 512, 513, 600,
      768, 1000, 1024, 1025, 1500, 2048]
DS = [1, 2, 5, 13, 16, 20, 24, 31, 32, 33, 39, 40, 48, 59, 60, 63, 64, 65, 72, 80, 96, 100, 120, 128, 132, 140, 156, 159]
bad = 0
worst = 0.0
for case in range(CASES):
    C = int(rng.choice(CS)); D = int(rng.choice(DS))
    off = draw_sessions()
    T = int(off[-1])
    X = (torch.from_numpy(rng.standard_normal((T, D)).astype(np.float32)) * XS + XM).to(dev)
    w = rng.random(C) + 0.05; w /= w.sum()
    mu = rng.standard_normal((C, D)) * MS
    invcov = 1.0 / (V0 + rng.random((C, D)) * V1)
    packed = eng.pack(w, mu, invcov)
    em_ref, s0_ref, s1_ref, lse_ref, lp_ref = reference(X, w, mu, invcov, off)
    try:
        em = eng.estep(X, packed)
    except Exception as e:
        print("case %3d C=%4d D=%3d T=%6d  rejected: %s" % (case, C, D, T, e), flush=True)
        continue
    n = C; nf = C * D
    errs = {"N": rel(em[:n], em_ref[:n]), "F": rel(em[n:n + nf], em_ref[n:n + nf]),
            "S": rel(em[n + nf:n + 2 * nf], em_ref[n + nf:n + 2 * nf]),
            "llk": abs(em[-1].item() - em_ref[-1].item()) / max(abs(em_ref[-1].item()), 1e-30)}
    out = torch.empty_like(em)
    try:
        eng.estep_noalloc(X, packed, out)
        errs["noalloc"] = rel(out, em)       # same kernels: equal up to the order of the slab sums
    except Exception:
        pass                                    # shape beyond the built-in scratch: rejected, as documented
    errs["loglik"] = rel(eng.loglik(X, packed).double(), lse_ref)
    lpd, llk2 = eng.logprob(X, packed, with_loglik=True)
    errs["logprob"] = max(rel(lpd, lp_ref), rel(llk2.double(), lse_ref))
    s0, s1 = eng.bw_stats(X, off, packed)
    errs["stat0"] = rel(s0, s0_ref); errs["stat1"] = rel(s1, s1_ref)
    s0b, s1b = eng.bw_stats(X, off, packed)                      # again into fresh tensors: no state between calls
    errs["again"] = max(rel(s0b, s0), rel(s1b, s1))
    torch.cuda.synchronize()
    m = max(errs.values())
    worst = max(worst, m)
    flag = "" if m <= BAR and all(math.isfinite(v) for v in errs.values()) else "  <-- over the bar"
    if flag:
        bad += 1
    print("case %3d C=%4d D=%3d T=%6d S=%3d  " % (case, C, D, T, len(off) - 1)
          + " ".join("%s=%.1e" % kv for kv in errs.items()) + flag, flush=True)
print("cases %d, over the bar %d, worst %.2e" % (CASES, bad, worst))
sys.exit(1 if bad else 0)
