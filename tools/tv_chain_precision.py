"""float32 against float64 per-session chain of the TV E-step (SKM_TV_CHAIN): i-vectors, uncertainties and training
accumulators of the same synthetic sessions at cfg5's shape (R=400, C=2048, D=60), and the time of each.
   python tools/tv_chain_precision.py [S]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sidekit_amd as sk
from bench import random_ubm
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C, D, R = 2048, 60, 400
eng = sk.default_engine()
dev = eng.device
ubm = random_ubm(sk, C, D, seed=11)
gen = torch.Generator(device=dev); gen.manual_seed(4321)
mu_sv = torch.as_tensor(ubm.get_mean_super_vector(), device=dev)
sd_sv = torch.as_tensor(1.0 / np.sqrt(ubm.get_invcov_super_vector()), device=dev)
stat0 = torch.empty((S, C), dtype=torch.float32, device=dev)
stat1 = torch.empty((S, C * D), dtype=torch.float32, device=dev)
for a0 in range(0, S, 2048):
    b0 = min(S, a0 + 2048)
    post = torch.softmax(2.0 * torch.randn(b0 - a0, C, generator=gen, device=dev), dim=1)
    stat0[a0:b0] = (2000.0 * post).to(torch.float32)
    n_rep = stat0[a0:b0].repeat_interleave(D, dim=1).to(torch.float64)
    stat1[a0:b0] = (n_rep * mu_sv + torch.sqrt(n_rep) * sd_sv * torch.randn(b0 - a0, C * D, generator=gen, device=dev, dtype=torch.float64)).to(torch.float32)
    del n_rep, post
ft, naug = eng.tv_whiten(stat0, stat1, ubm.get_mean_super_vector(), ubm.get_invcov_super_vector())
del stat1
T = torch.as_tensor(0.1 * np.random.RandomState(7).randn(C * D, R), device=dev)
T32, G = eng.tv_prepare(T, C, D)
res = {}
for mode in ("f64", "f32"):
    os.environ["SKM_TV_CHAIN"] = mode
    out = {}
    for name in ("extract", "extract_sigma", "train"):
        def go():
            if name == "train":
                A = torch.zeros((C + 1, R * R), dtype=torch.float64, device=dev)
                Cc = torch.zeros((R, C * D), dtype=torch.float64, device=dev)
                e, _ = eng.tv_estep(naug, ft, T32, G, C, D, batch=4096, A_aug=A, Cacc=Cc)
                return [e, A, Cc]
            e, sg = eng.tv_estep(naug, ft, T32, G, C, D, batch=4096, uncertainty=(name == "extract_sigma"))
            return [e, sg]
        r = go(); torch.cuda.synchronize(); del r
        eng.profile(True)
        t0 = time.perf_counter()
        r = go(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = {k: round(v[1], 2) for k, v in eng.profile_read().items()}
        eng.profile(False)
        out[name] = (dt, [None if x is None else x.cpu().numpy() for x in r], prof)
        del r
    res[mode] = out
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
for name in ("extract", "extract_sigma", "train"):
    t64, r64, p64 = res["f64"][name]; t32, r32, p32 = res["f32"][name]
    line = "%-14s S=%d  f64 %.1f ms  f32 %.1f ms" % (name, S, t64 * 1e3, t32 * 1e3)
    for i, (a, b) in enumerate(zip(r32, r64)):
        if a is None: continue
        if name == "train" and i == 1:
            a, b = a[:C], b[:C]
        line += "  out%d rel %.1e" % (i, rel(a, b))
        if i == 0:
            cos = (a * b).sum(1) / np.linalg.norm(a, axis=1) / np.linalg.norm(b, axis=1)
            line += " (1-cos max %.1e, row-wise rel max %.1e)" % (1 - cos.min(), (np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)).max())
    print(line, flush=True)
    print("   f64 kernels", p64, flush=True)
    print("   f32 kernels", p32, flush=True)
