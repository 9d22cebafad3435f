"""Baum-Welch statistics per kernel for small models / wide rows: python tools/time_bw_small.py C D [sessions frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
C, D = int(sys.argv[1]), int(sys.argv[2])
S = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
L = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
eng = sk.default_engine()
X = synth_frames_device(torch, S * L, D, 1234, eng.device)
ubm = random_ubm(sk, C, D)
packed = eng.pack(ubm.w, ubm.mu, ubm.invcov)
off = np.arange(S + 1, dtype=np.int64) * L
for mode, pair in (("1", "1"), ("1", "0"), ("0", "1")):
    os.environ["SKM_BW_SMALL"] = mode
    os.environ["SKM_BW_PAIR"] = pair
    for _ in range(2): eng.bw_stats(X, off, packed)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(3): eng.bw_stats(X, off, packed)
    torch.cuda.synchronize()
    k = {n: round(v[1] / v[0], 3) for n, v in eng.profile_read().items()}
    eng.profile(False)
    tot = sum(k.values())
    print("C=%d D=%d SKM_BW_PAIR=%s SKM_BW_SMALL=%s total %.3f ms  %.1f TF algorithmic (%.2f of peak)  %s" % (
        C, D, pair, mode, tot, 6.0 * D * C * S * L / tot / 1e9, 6.0 * D * C * S * L / tot / 1e9 / 157.3, k), flush=True)
