"""Kernel time of one EM iteration at every component count that Mixture.EM_split walks through (1, 2, 4, ..., 512),
D = 60, frames resident in HBM: which stages sit on the HBM roof (2.4 GB of frames per iteration) and which on the MFMA's.
   python tools/time_em_stages.py [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
D = 60
eng = sk.default_engine()
X = synth_frames_device(torch, T, D, 1234, eng.device)
for C in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    m = random_ubm(sk, C, D, seed=3)
    for rep in range(2):
        if rep == 1:
            eng.profile(True)
        m._em_iterations_device(X, 2)
        torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    tot = sum(v[1] for v in prof.values()) / 2
    print("C=%4d: %.3f ms per iteration (%.2f TB/s of frames, %.1f TF)  %s" % (
        C, tot, T * D * 4 / (tot * 1e-3) / 1e12, 8.0 * D * C * T / (tot * 1e-3) / 1e12,
        {k: round(v[1] / 2, 3) for k, v in sorted(prof.items())}), flush=True)
