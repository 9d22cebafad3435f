"""Per-kernel time of one E-step over T frames for a list of model sizes (the stages of EM_split):
   python tools/time_estep.py [C ...]   (default 2 4 ... 512; T=10M, D=60)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sidekit_amd as sk
from bench import synth_frames_device, random_ubm
T = int(os.environ.get("T", 10_000_000)); D = int(os.environ.get("D", 60))
eng = sk.default_engine()
X = synth_frames_device(torch, T, D, 1234, eng.device)
for C in [int(a) for a in sys.argv[1:]] or [2, 4, 8, 16, 32, 64, 128, 256, 512]:
    ubm = random_ubm(sk, C, D)
    packed = eng.pack(ubm.w, ubm.mu, ubm.invcov)
    for _ in range(2): eng.estep(X, packed)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(3): eng.estep(X, packed)
    torch.cuda.synchronize()
    k = {n: round(v[1] / v[0], 3) for n, v in eng.profile_read().items()}
    eng.profile(False)
    tot = sum(k.values())
    print("C=%4d total %.3f ms  %.1f TF algorithmic (%.2f of peak)  %s" % (C, tot, 8.0 * D * C * T / tot / 1e9, 8.0 * D * C * T / tot / 1e9 / 157.3, k), flush=True)
