"""Full-size single-GPU runs of BASELINE configs[3] and configs[4] (one MI355X holds them whole):
   cfg4  Baum-Welch N/F, C=2048, D=60, 50 000 sessions x 2 000 frames (100 M frames, stat1 = 49 GB float64)
   cfg5  TV training iteration + extract_ivectors, R=400, C=2048, 50 000 sessions
Prints one JSON line.  Usage: python tools/full_size.py [--sessions 50000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sessions", type=int, default=50000)
    ap.add_argument("--frames-per-session", type=int, default=2000)
    args = ap.parse_args()
    import torch
    import sidekit_amd as sk
    from bench import synth_frames_device, random_ubm
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = sk.default_engine()
    C, D, R, S, L = 2048, 60, 400, args.sessions, args.frames_per_session
    out = {"sessions": S, "frames_per_session": L, "C": C, "D": D, "R": R}

    # ---- cfg4: Baum-Welch statistics of all sessions in one call
    ubm = random_ubm(sk, C, D, seed=11)
    packed = eng.pack(ubm.w, ubm.mu, ubm.invcov)
    X = synth_frames_device(torch, S * L, D, seed=1234, device=dev)
    seg = torch.arange(0, S + 1, dtype=torch.int64, device=dev) * L
    s0 = torch.empty((S, C), dtype=torch.float64, device=dev)
    s1 = torch.empty((S, C * D), dtype=torch.float64, device=dev)
    eng.bw_stats(X[:200 * L], seg[:201], packed, s0[:200], s1[:200])       # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.bw_stats(X, seg, packed, s0, s1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["bw_stats"] = {"seconds": dt, "frames_per_s": S * L / dt,
                       "frac_of_f32_mfma_peak": 6.0 * D * C * S * L / dt / 157.3e12,
                       "check_sum_N_over_frames": float(s0.sum().item() / (S * L))}
    print("bw done", out["bw_stats"], file=sys.stderr, flush=True)
    del X

    # ---- cfg5: the statistics go through float32 (the reference stores them as float32, statserver.py:439-446)
    stat0 = s0.to(torch.float32)
    del s0
    stat1 = torch.empty((S, C * D), dtype=torch.float32, device=dev)
    for a in range(0, S, 2048):
        stat1[a:a + 2048] = s1[a:a + 2048].to(torch.float32)
    del s1
    torch.cuda.empty_cache()
    ft, naug = eng.tv_whiten(stat0, stat1, ubm.get_mean_super_vector(), ubm.get_invcov_super_vector())
    del stat1, stat0
    torch.cuda.empty_cache()
    fa = sk.FactorAnalyser()
    T_dev = torch.as_tensor(0.1 * np.random.RandomState(7).randn(C * D, R), device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T_dev = fa._tv_iteration(ft, naug, T_dev, C, D, R, S, True, 4096)
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    T_dev = fa._tv_iteration(ft, naug, T_dev, C, D, R, S, True, 4096)
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    out["tv_train"] = {"seconds_iteration_1": dt1, "seconds_iteration_2": dt2, "sessions_per_s": S / dt2,
                       "algorithmic_tflops": 9.18e8 * S / dt2 / 1e12,
                       "T_finite": bool(torch.isfinite(T_dev).all().item())}
    print("tv train done", out["tv_train"], file=sys.stderr, flush=True)
    T32, G = eng.tv_prepare(T_dev, C, D)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e_h, _ = eng.tv_estep(naug, ft, T32, G, C, D, batch=4096, uncertainty=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["extract_ivectors"] = {"seconds": dt, "ivectors_per_s": S / dt, "algorithmic_tflops": 4.5e8 * S / dt / 1e12,
                               "finite": bool(torch.isfinite(e_h).all().item()),
                               "mean_norm": float(e_h.norm(dim=1).mean().item())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
