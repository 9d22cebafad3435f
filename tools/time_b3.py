"""Split-precision E-step (SKM_SPLIT=bf16x3, csrc/skm_gmm_b3.hip) against the float32-MFMA kernel: statistics of both
against a float64 torch restatement on the device, then the time of each over T frames.
   python tools/time_b3.py [T]      (default 10 M frames; C = 512, D = 60)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
from conftest import rel, rel_rows, rel_rownorm

C, D = int(os.environ.get("C", 512)), int(os.environ.get("D", 60))
eng = sk.default_engine()
ubm = random_ubm(sk, C, D)
packed = eng.pack_stats(ubm.w, ubm.mu, ubm.invcov) if os.environ.get("REF", "1") == "1" else eng.pack(ubm.w, ubm.mu, ubm.invcov)


def run(X, split):
    if split:
        os.environ["SKM_SPLIT"] = "bf16x3"
    else:
        os.environ.pop("SKM_SPLIT", None)
    out = eng.estep(X, packed)
    torch.cuda.synchronize()
    return eng.to_host(out)


def f64_ref(X):
    x = X.to(torch.float64)
    w = torch.as_tensor(ubm.w, device=X.device)
    mu = torch.as_tensor(ubm.mu, device=X.device)
    ic = torch.as_tensor(ubm.invcov, device=X.device)
    N = torch.zeros(C, dtype=torch.float64, device=X.device)
    F = torch.zeros(C, D, dtype=torch.float64, device=X.device)
    S = torch.zeros(C, D, dtype=torch.float64, device=X.device)
    llk = 0.0
    kc = torch.log(w) + 0.5 * torch.log(ic).sum(1) - 0.5 * D * np.log(2 * np.pi) - 0.5 * (mu * mu * ic).sum(1)
    for a in range(0, x.shape[0], 16384):
        xb = x[a:a + 16384]
        lp = kc[None] + xb @ (mu * ic).t() - 0.5 * (xb * xb) @ ic.t()
        lse = torch.logsumexp(lp, 1, keepdim=True)
        g = torch.exp(lp - lse)
        N += g.sum(0); F += g.t() @ xb; S += g.t() @ (xb * xb); llk += float(lse.sum())
    return np.concatenate([N.cpu().numpy(), F.cpu().numpy().ravel(), S.cpu().numpy().ravel(), [llk]])


for T in (32 * 7 + 5, 100_003):
    X = synth_frames_device(torch, T, D, 77, eng.device)
    ref = f64_ref(X)
    for split in (0, 1):
        o = run(X, split)
        N, F, S = o[:C], o[C:C + C * D].reshape(C, D), o[C + C * D:C + 2 * C * D].reshape(C, D)
        rN, rF, rS = ref[:C], ref[C:C + C * D].reshape(C, D), ref[C + C * D:C + 2 * C * D].reshape(C, D)
        print("T=%6d split=%d  N rows %.2e  F rownorm %.2e rows %.2e  S rows %.2e  llk %.2e" % (
            T, split, rel_rows(N, rN), rel_rownorm(F, rF), rel_rows(F, rF), rel_rows(S, rS), abs(o[-1] - ref[-1]) / abs(ref[-1])), flush=True)

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
X = synth_frames_device(torch, T, D, 1234, eng.device)
for split in (0, 1, 0, 1):
    for _ in range(2):
        run(X, split)
    eng.profile(True)
    for _ in range(3):
        run(X, split)
    k = {n: round(v[1] / v[0], 3) for n, v in eng.profile_read().items()}
    eng.profile(False)
    tot = sum(k.values())
    print("split=%d total %.3f ms  %.1f TF algorithmic (%.2f of the f32 MFMA peak)  %s" % (
        split, tot, 8.0 * D * C * T / tot / 1e9, 8.0 * D * C * T / tot / 1e9 / 157.3, k), flush=True)
